"""Host-side build of the bounding-box hierarchy (RT_OPT_FILTER 3): structural soundness on the
scene shapes the renderer meets.  The library call needs no device.  The reference has no
hierarchy (Source/main.cpp:534 tests every sphere); what the hierarchy may skip is checked on the
GPU against the brute-force oracle."""
import numpy as np
import pytest

import rtamd
import scenes


def check(spheres, leaf=4):
    st, nodes, levels, large, records = rtamd.accel_selfcheck(spheres, leaf)
    assert st == 0
    assert records == len(spheres)
    return nodes, levels, large


@pytest.mark.parametrize("leaf", [1, 2, 4, 15])
def test_random_spheres_scene(leaf):
    sc = scenes.random_spheres(500)
    nodes, levels, large = check(sc, leaf)
    assert large == 4  # the ground sphere of radius 1000 and the three of radius 1 (5 x the median) get no box
    assert nodes >= (496 + leaf - 1) // leaf - 1
    assert levels <= 28 + 9


def test_default_scene_and_tiny_scenes():
    nodes, levels, large = check(rtamd.default_scene())
    assert large == 1 and nodes >= 1  # the radius-100 ground; three spheres in the tree
    one = rtamd.default_scene()[:1]
    assert check(one) == (1, 1, 0)  # a single leaf still gets a root node
    assert check(rtamd.default_scene()[:0]) == (0, 0, 0)
    two_big = rtamd.default_scene()[[3, 3]]
    nodes, levels, large = check(two_big)
    assert large == 0 and nodes == 1  # equal radii: nothing dwarfs the median


def test_ten_thousand_spheres_and_depth_bound():
    sc = scenes.random_spheres(10000, grid_half=50, seed=7)
    nodes, levels, large = check(sc, 2)
    assert large == 4 and levels <= 40 and nodes >= 4997


def test_degenerate_inputs_keep_depth_bounded():
    n = 3000
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["radius"] = 0.5
    sc["material"] = rtamd.MAT_DIFFUSE
    # all centres identical: no bin separates them, even splits must bound the depth
    sc["center"] = (1.0, 2.0, 3.0)
    nodes, levels, large = check(sc, 1)
    assert levels <= 13 and nodes == n - 1
    # a geometric progression along one axis (the worst case for spatial bins)
    sc["center"][:, 0] = np.float32(2.0) ** np.linspace(-20, 27, n).astype(np.float32)
    nodes, levels, large = check(sc, 2)
    assert levels <= 28 + 12
    # many large spheres: only the 16 biggest go on the always-tested list
    sc["radius"][:40] = np.linspace(100, 200, 40).astype(np.float32)
    nodes, levels, large = check(sc, 4)
    assert large == 16


def test_bad_arguments():
    assert rtamd.accel_selfcheck(rtamd.default_scene(), 0)[0] != 0
    assert rtamd.accel_selfcheck(rtamd.default_scene(), 16)[0] != 0


@pytest.mark.parametrize("field,value", [("center", np.nan), ("center", np.inf), ("center", -np.inf), ("center", 1e12),
                                         ("radius", np.nan), ("radius", np.inf), ("radius", 0.0), ("radius", 1e-25)])
def test_spheres_outside_the_builders_domain_are_refused(field, value):
    """The builder sorts and bins by coordinates; NaN, Inf, huge magnitudes and zero radii are what
    rt_set_scene never hands to it (such scenes take the exact list scan), so the host-only entry
    point must refuse them instead of building on them."""
    sc = scenes.random_spheres(200, grid_half=7)
    if field == "center":
        sc["center"][17, 1] = value
    else:
        sc["radius"][17] = value
    st = rtamd.accel_selfcheck(sc, 4)[0]
    assert st == -1
