"""The box hierarchy (RT_OPT_FILTER 3) must select exactly the spheres the reference's brute-force
loop (Source/main.cpp:534-546) selects: frames and ray counts bit-identical to the oracle, which
tests every sphere."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import oracle_lib as O
import rtamd
import scenes

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
C3_CAM = scenes.RANDOM_SPHERES_CAMERA


@pytest.fixture(scope="module")
def ctx():
    c = rtamd.Context(0)
    c.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    yield c
    c.close()


def same_as_oracle(ctx, sc, cam, W, H, spp, depth, seed, flags=0):
    ctx.set_scene(sc)
    ctx.reset_stats()
    got = ctx.render(rtamd.make_camera(*cam), rtamd.params(W, H, spp, depth, seed=seed, flags=flags))
    want, _, rays = O.trace_tile(sc, O.camera(*cam), W, H, spp, depth, O.RNG_COUNTER, seed=seed, flags=flags)
    assert np.array_equal(got, want)
    assert ctx.stats().rays == rays
    return got


@pytest.mark.parametrize("leaf", [1, 4, 15])
def test_reference_scene_and_random_spheres(ctx, leaf):
    ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, leaf)
    try:
        same_as_oracle(ctx, rtamd.default_scene(), scenes.DEFAULT_CAMERA, 128, 72, 8, 10, 7)
        sc = np.fromfile(os.path.join(G, "scene_c3_500.bin"), dtype=rtamd.SPHERE_DTYPE)
        same_as_oracle(ctx, sc, C3_CAM, 160, 90, 4, 8, 1)
        same_as_oracle(ctx, sc, C3_CAM, 96, 54, 4, 12, 3, flags=rtamd.FLAG_DIELECTRIC)  # rays from inside spheres
    finally:
        ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 0)  # the shipped setting: chosen when the tree is built


def test_ten_thousand_spheres(ctx):
    sc = scenes.random_spheres(10000, grid_half=50, seed=7)
    same_as_oracle(ctx, sc, C3_CAM, 64, 36, 2, 8, 4)
    same_as_oracle(ctx, sc, ((0.0, 30.0, 0.0), (5.0, 0.0, 5.0), (0.0, 0.0, 1.0), 60.0), 64, 36, 2, 8, 5)  # from above


@pytest.mark.parametrize("n", [2040, 2047, 2048, 2049, 2052])
def test_trees_at_the_limit_of_two_byte_stack_entries(ctx, n):
    """Trees whose nodes fit the LDS copy are walked with child references in two bytes, which holds
    a leaf reference only up to record 2047: with leaves of 15 a 2 040-sphere scene has few enough
    nodes for LDS and records right up to that limit; a few spheres more and the tree must be walked
    from memory with four-byte entries.  Both sides of the limit against the oracle."""
    ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 15)
    try:
        st, nodes, levels, large, records = rtamd.accel_selfcheck(scenes.random_spheres(n, grid_half=24, seed=3), 15)
        assert st == 0 and nodes <= 512 and records == n
        same_as_oracle(ctx, scenes.random_spheres(n, grid_half=24, seed=3), C3_CAM, 96, 54, 2, 8, 11)
    finally:
        ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 0)  # the shipped setting: chosen when the tree is built


@pytest.mark.parametrize("n,grid", [(170, 7), (700, 13), (1000, 16), (1500, 19), (2000, 22), (2048, 23), (2049, 23), (3000, 27)])
def test_leaf_size_chosen_by_the_library(ctx, n, grid):
    """With RT_OPT_BVH_LEAF_SPHERES 0 (shipped) the library builds the tree with leaves of up to 4, 8, 12 or
    15 spheres, whichever keeps the LDS copy small enough (or 8 for trees walked in memory): scene sizes on
    either side of every switch, against the oracle, and against the same scene under a forced leaf size."""
    assert ctx.get_option(rtamd.OPT_BVH_LEAF_SPHERES) == 0
    sc = scenes.random_spheres(n, grid_half=grid, seed=n)
    auto = same_as_oracle(ctx, sc, C3_CAM, 96, 54, 2, 8, 21)
    ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 5)
    try:
        ctx.reset_stats()
        assert np.array_equal(ctx.render(rtamd.make_camera(*C3_CAM), rtamd.params(96, 54, 2, 8, seed=21)), auto)
    finally:
        ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 0)


def test_hundred_thousand_spheres_walked_from_memory(ctx):
    """A tree far too large for LDS (33 000 nodes, 17 levels): nodes and leaf records come through
    the caches, and bounce stack entries take four bytes (n > 65536)."""
    sc = scenes.random_spheres(100000, grid_half=160, seed=5)
    cam = ((40.0, 6.0, 10.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 30.0)
    same_as_oracle(ctx, sc, cam, 48, 27, 2, 6, 8)
    hits = ctx.primary_hits(rtamd.make_camera(*cam), 128, 72)
    assert np.array_equal(hits, O.primary_hits(sc, O.camera(*cam), 128, 72))
    assert len(np.unique(hits)) > 50


def test_coincident_and_nested_spheres_keep_the_tie_rule(ctx):
    """Equal hit distance: the later sphere of the list wins (Source/main.cpp:492-499, 539-545),
    wherever the hierarchy has put the two."""
    rs = np.random.RandomState(5)
    n = 64
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = rs.uniform(-4, 4, size=(n, 3)).astype(np.float32) + np.array([0, 0, -8], dtype=np.float32)
    sc["radius"] = rs.uniform(0.3, 0.9, size=n).astype(np.float32)
    sc["material"] = rtamd.MAT_DIFFUSE
    sc["color"] = rs.uniform(0, 255, size=(n, 3)).astype(np.float32)
    # exact duplicates at the two ends of the list, and concentric shells
    sc[60], sc[61], sc[62], sc[63] = sc[3], sc[7], sc[3], sc[20]
    sc["color"][60:] = rs.uniform(0, 255, size=(4, 3)).astype(np.float32)
    sc[30]["center"], sc[30]["radius"] = sc[31]["center"], sc[31]["radius"] * 0.5
    got = same_as_oracle(ctx, sc, scenes.DEFAULT_CAMERA, 160, 90, 3, 6, 9)
    hits = O.primary_hits(sc, O.camera(*scenes.DEFAULT_CAMERA), 160, 90)
    assert (hits == 62).any() and not (hits == 3).any() and not (hits == 60).any()
    assert (got != 255).any()


def test_rays_outside_the_filters_preconditions_take_the_exact_loop(ctx):
    """NaN rays: the reference's comparisons are all false for them, so its loop accepts every
    sphere in turn and the last one of the list wins (Source/main.cpp:481-499, 539-545) -- a walk
    over boxes would find nothing.  Such rays must be scanned by the in-order loop instead."""
    sc = scenes.random_spheres(200, grid_half=7, seed=2)
    sc[-1]["material"], sc[-1]["color"] = rtamd.MAT_DIFFUSE, (200.0, 100.0, 50.0)
    nan_cam = ((float("nan"), 2.0, 3.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 20.0)
    got = same_as_oracle(ctx, sc, nan_cam, 48, 27, 3, 6, 2)
    assert (got != 255).any()  # not sky: the rays "hit"
    # a camera far outside the 2^28 bound (its rays are too coarse to hit anything, like in the reference)
    far = ((1.0e9, 2.0e8, 3.0e8), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 1.0e-6)
    same_as_oracle(ctx, sc, far, 48, 27, 3, 6, 2)


def test_scene_of_large_spheres_only_and_single_sphere(ctx):
    sc = rtamd.default_scene()[[3]]  # one sphere: a single leaf under a root
    same_as_oracle(ctx, sc, scenes.DEFAULT_CAMERA, 64, 36, 2, 6, 1)
    two = rtamd.default_scene()[[1, 3]]
    two[0]["radius"] = 0.01  # the other is 10^4 times larger: on the always-tested list
    same_as_oracle(ctx, two, scenes.DEFAULT_CAMERA, 64, 36, 2, 6, 1)


def test_clustered_scene_with_wide_radius_spread(ctx):
    """Clusters far apart, radii over three decades: boxes of very different sizes, deep SAH splits."""
    rs = np.random.RandomState(11)
    n = 1500
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    centres = rs.uniform(-60, 60, size=(12, 3))
    which = rs.randint(0, 12, size=n)
    sc["center"] = (centres[which] + rs.normal(0, 1.5, size=(n, 3))).astype(np.float32)
    sc["radius"] = (10.0 ** rs.uniform(-2.5, 0.3, size=n)).astype(np.float32)
    sc["material"] = rs.randint(0, 4, size=n)
    sc["color"] = np.where((sc["material"] == 1)[:, None], rs.uniform(0, 255, (n, 3)), rs.uniform(0.2, 1, (n, 3)))
    sc["fuzz"] = rs.uniform(0, 1, size=n)
    cam = ((0.0, 20.0, 150.0), (0.0, 0.0, 0.0), (0.0, 1.0, 0.0), 50.0)
    got = same_as_oracle(ctx, sc, cam, 128, 72, 3, 8, 6)
    assert (got != got[0, 0]).any()


@pytest.mark.parametrize("scale,offset", [(1.0, 0.0), (1e-3, 0.0), (50.0, 3e4), (1e4, -2e6)])
def test_grazing_rays_agree_with_the_exact_scan(ctx, scale, offset):
    """Rays aimed to pass a sphere at (1 + eps) radii from its centre, eps from -1e-4 to 1e-4 around the
    point where the reference's float discriminant changes sign, from near and far origins: the
    padded boxes must never cost a hit that the reference's own loop (exact scan) finds."""
    rs = np.random.RandomState(int(scale * 7) % 1000 + 1)
    n = 400
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = (rs.uniform(-10, 10, size=(n, 3)) * scale + offset).astype(np.float32)
    sc["radius"] = (10.0 ** rs.uniform(-2, 0, size=n) * scale).astype(np.float32)
    sc["material"] = rtamd.MAT_DIFFUSE
    m = 200000
    which = rs.randint(0, n, size=m)
    c = sc["center"][which].astype(np.float64)
    r = sc["radius"][which].astype(np.float64)
    dist = r * 10.0 ** rs.uniform(0.1, 3.5, size=m)           # origin 1.3 .. 3000 radii away
    u = rs.normal(size=(m, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    o = c + u * dist[:, None]
    v = np.cross(u, rs.normal(size=(m, 3))); v /= np.linalg.norm(v, axis=1, keepdims=True)
    eps = rs.choice([0.0, 1e-7, -1e-7, 1e-6, -1e-6, 1e-5, -1e-5, 1e-4, -1e-4, 3e-8, -3e-8], size=m)
    b = r * (1.0 + eps)                                        # impact parameter
    # direction from o to the point at distance b from the centre, perpendicular to the line of sight
    foot = c + v * b[:, None]
    d = foot - o
    d *= (10.0 ** rs.uniform(-2, 2, size=m))[:, None]         # un-normalised, like the reference's rays
    rays = np.concatenate([o, d], axis=1).astype(np.float32)
    ctx.set_scene(sc)
    try:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_EXACT)
        want = ctx.intersect(rays)
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
        got = ctx.intersect(rays)
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    assert got.tobytes() == want.tobytes()
    hit_target = (want["index"] == which).mean()
    assert 0.2 < hit_target < 0.8  # the batch really straddles the edge


def test_random_scenes_every_scan_gives_the_exact_scans_frame(ctx):
    """60 random scenes (1..3000 spheres; boxes, planes, lines, clusters; scales 1e-3..1e4, far
    offsets, giants, duplicates, negative radii, extension flags): vector filter, matrix filter and
    hierarchy reproduce the frame and the ray count of the reference's own discriminant scan."""
    import sys

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import fuzz_scans

    try:
        for seed in range(5000, 5060):
            rs = np.random.RandomState(seed)
            sc, cam = fuzz_scans.random_scene(rs)
            p = rtamd.params(96, 54, 3, int(rs.choice([1, 4, 8])), seed=seed,
                             flags=int(rs.choice([0, 0, rtamd.FLAG_DIELECTRIC])))
            c = rtamd.make_camera(*cam)
            ctx.set_scene(sc)
            frames, rays = [], []
            for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
                ctx.set_option(rtamd.OPT_FILTER, filt)
                ctx.reset_stats()
                frames.append(ctx.render(c, p))
                rays.append(ctx.stats().rays)
            assert all(np.array_equal(frames[0], f) for f in frames[1:]), seed
            assert len(set(rays)) == 1, seed
    finally:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
