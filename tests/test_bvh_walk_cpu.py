"""The hierarchy scan without a GPU: oracle/libbvhcheck.so builds the tree with the library's own host
code and walks it on the CPU with the device walk's arithmetic (padding, plane offsets, order-free
closest-hit rule); what it selects must be what the reference's brute-force loop selects
(IntersectSceneSpheres, Source/main.cpp:528-549, as restated in the oracle), index and root, bit for bit."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as O
import scenes

_LIB = os.path.join(O.ORACLE_DIR, "libbvhcheck.so")


@pytest.fixture(scope="module")
def walker():
    src = os.path.join(O.ORACLE_DIR, "bvh_walk_check.cpp")
    if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", O.ORACLE_DIR, "libbvhcheck.so"], stdout=subprocess.DEVNULL)
    L = C.CDLL(_LIB)
    L.orc_bvh_intersect.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    return L


def brute_force(sc, rays):
    idx = np.empty(len(rays), dtype=np.int32)
    ts = np.zeros(len(rays), dtype=np.float32)
    t = C.c_float()
    f3 = C.c_float * 3
    lib = O.lib()
    for i, r in enumerate(rays):
        idx[i] = lib.orc_intersect(sc.ctypes.data, len(sc), f3(*r[:3]), f3(*r[3:]), 0.001, float("inf"), C.byref(t), None, None)
        ts[i] = t.value if idx[i] >= 0 else np.inf
    return idx, ts


def walk(L, sc, rays, leaf):
    sc = np.ascontiguousarray(sc)
    rays = np.ascontiguousarray(rays, dtype=np.float32)
    idx = np.empty(len(rays), dtype=np.int32)
    ts = np.empty(len(rays), dtype=np.float32)
    nodes = L.orc_bvh_intersect(sc.ctypes.data, len(sc), leaf, rays.ctypes.data, len(rays), idx.ctypes.data, ts.ctypes.data)
    return idx, ts, nodes


def mixed_rays(rs, sc, m):
    """Random rays through the scene's bounds, rays aimed at sphere rims, rays from sphere surfaces,
    axis-parallel rays."""
    c, r = sc["center"].astype(np.float64), np.abs(sc["radius"].astype(np.float64))
    lo, hi = (c - r[:, None]).min(axis=0), (c + r[:, None]).max(axis=0)
    finite = r < 50 * np.median(r)
    lo, hi = (c[finite] - r[finite, None]).min(axis=0), (c[finite] + r[finite, None]).max(axis=0)
    span = hi - lo
    o = lo - 0.5 * span + rs.uniform(size=(m, 3)) * 2.0 * span
    d = (lo + rs.uniform(size=(m, 3)) * span) - o
    k = m // 4
    which = rs.randint(0, len(sc), size=k)                    # grazing: aim at the rim
    v = rs.normal(size=(k, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
    eps = rs.choice([0.0, 1e-7, -1e-7, 1e-5, -1e-5, 1e-3], size=k)
    d[:k] = c[which] + v * (r[which] * (1 + eps))[:, None] - o[:k]
    which = rs.randint(0, len(sc), size=k)                    # bounce rays: start on a surface
    n = rs.normal(size=(k, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    o[k:2 * k] = c[which] + n * r[which][:, None]
    d[k:2 * k] = n + rs.normal(size=(k, 3)) * 0.7
    d[2 * k:2 * k + 200, rs.randint(0, 3)] = 0.0              # zero direction components
    d[2 * k + 200:2 * k + 300, :2] = 0.0
    d *= (10.0 ** rs.uniform(-2, 2, size=m))[:, None]         # un-normalised, like the reference's
    return np.concatenate([o, d], axis=1).astype(np.float32)


def clustered(rs, n, scale, offset):
    sc = np.zeros(n, dtype=O.SPHERE_DTYPE)
    cen = rs.uniform(-40, 40, size=(8, 3))
    sc["center"] = ((cen[rs.randint(0, 8, size=n)] + rs.normal(0, 2.0, size=(n, 3))) * scale + offset).astype(np.float32)
    sc["radius"] = (10.0 ** rs.uniform(-2, 0.4, size=n) * scale).astype(np.float32)
    sc["material"] = O.MAT_DIFFUSE
    return sc


@pytest.mark.parametrize("leaf", [1, 4, 15])
def test_walk_selects_what_the_brute_force_loop_selects(walker, leaf):
    rs = np.random.RandomState(100 + leaf)
    cases = [
        ("random spheres 500", scenes.random_spheres(500), 30000),
        ("reference scene", O.default_scene(), 5000),
        ("clusters", clustered(rs, 800, 1.0, 0.0), 20000),
        ("clusters, small and far from the origin", clustered(rs, 300, 1e-2, 500.0), 15000),
        ("clusters, large", clustered(rs, 300, 1e3, -4e5), 15000),
    ]
    for name, sc, m in cases:
        rays = mixed_rays(rs, sc, m)
        got_i, got_t, nodes = walk(walker, sc, rays, leaf)
        tame = got_i != -2
        assert tame.mean() > 0.99, name
        want_i, want_t = brute_force(sc, rays[tame])
        assert np.array_equal(got_i[tame], want_i), (name, int((got_i[tame] != want_i).sum()))
        hit = want_i >= 0
        assert got_t[tame][hit].tobytes() == want_t[hit].tobytes(), name
        assert 0.05 < hit.mean() < 0.98, (name, hit.mean())


def test_ten_thousand_spheres_and_duplicates(walker):
    rs = np.random.RandomState(5)
    sc = scenes.random_spheres(10000, grid_half=50, seed=7)
    sc[9999] = sc[1234]  # an exact duplicate late in the list: it must win the tie
    rays = mixed_rays(rs, sc, 6000)
    rays[:500, 3:] = sc["center"][1234] - rays[:500, :3]
    got_i, got_t, nodes = walk(walker, sc, rays, 4)
    want_i, want_t = brute_force(sc, rays)
    assert nodes > 2000
    assert np.array_equal(got_i, want_i)
    assert (want_i == 9999).sum() > 10 and not (want_i == 1234).any()
