"""Random scenes and cameras: every scan must give the exact scan's frame (run on the GPU box).
Usage: python tools/fuzz_scans.py [iterations] [first seed]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch  # noqa: E402,F401
import rtamd  # noqa: E402


def random_scene(rs):
    n = int(rs.choice([1, 2, 5, 13, 40, 64, 65, 130, 500, 1500, 3000]))
    scale = 10.0 ** rs.uniform(-3, 4)
    offset = rs.choice([0.0, 0.0, 10.0, 1e3, -1e5]) * scale
    spread = rs.choice(["box", "plane", "line", "clusters"])
    c = rs.uniform(-10, 10, size=(n, 3))
    if spread == "plane":
        c[:, 1] = rs.uniform(-0.1, 0.1, size=n)
    elif spread == "line":
        c[:, 1:] = rs.uniform(-0.2, 0.2, size=(n, 2))
    elif spread == "clusters":
        k = rs.randint(1, 6)
        c = rs.uniform(-10, 10, size=(k, 3))[rs.randint(0, k, size=n)] + rs.normal(0, 0.5, size=(n, 3))
    sc = np.zeros(n, dtype=rtamd.SPHERE_DTYPE)
    sc["center"] = (c * scale + offset).astype(np.float32)
    sc["radius"] = (10.0 ** rs.uniform(-2.5, 0.5, size=n) * scale).astype(np.float32)
    if rs.rand() < 0.3 and n > 3:
        sc["radius"][rs.randint(0, n, size=3)] *= 50.0          # a few giants
    if rs.rand() < 0.3 and n > 3:
        sc[n - 1] = sc[0]                                        # duplicates: ties
        sc[n // 2] = sc[0]
    if rs.rand() < 0.2:
        sc["radius"][rs.randint(0, n)] *= -1.0                  # the reference allows it
    sc["material"] = rs.randint(0, 4, size=n)
    sc["color"] = np.where((sc["material"] == 1)[:, None], rs.uniform(0, 255, (n, 3)), rs.uniform(0.2, 1, (n, 3)))
    sc["fuzz"] = rs.uniform(0, 1, size=n)
    eye = (rs.normal(size=3) * rs.uniform(3, 40)) * scale + offset
    cam = (tuple(eye), (offset, offset, offset), (0.0, 1.0, 0.0), float(rs.uniform(10, 100)))
    return sc, cam


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    ctx = rtamd.Context(0)
    bad = 0
    for it in range(iters):
        rs = np.random.RandomState(seed0 + it)
        sc, cam = random_scene(rs)
        flags = int(rs.choice([0, 0, rtamd.FLAG_DIELECTRIC, rtamd.FLAG_FLOAT_SHADING | rtamd.FLAG_DIELECTRIC]))
        W, H = 160, 90
        p = rtamd.params(W, H, 4, int(rs.choice([1, 4, 8, 16])), seed=it, flags=flags)
        c = rtamd.make_camera(*cam)
        ctx.set_scene(sc)
        frames, rays = [], []
        for filt in (rtamd.FILTER_EXACT, rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, int(rs.choice([1, 2, 4, 8])))
            ctx.reset_stats()
            frames.append(ctx.render(c, p))
            rays.append(ctx.stats().rays)
        ok = all(np.array_equal(frames[0], f) for f in frames[1:]) and len(set(rays)) == 1
        hit = float((frames[0] != frames[0][0, 0]).any())
        print(f"seed {seed0 + it:4d} n {len(sc):5d} flags {flags} depth {p.max_depth:2d} rays {rays[0]:8d} varied {hit:.0f} {'ok' if ok else 'MISMATCH'}",
              flush=True)
        bad += not ok
    ctx.close()
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
