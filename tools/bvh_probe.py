"""Hierarchy scan vs matrix scan: same bits, time per frame (run on the GPU box)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch  # noqa: E402,F401
import rtamd  # noqa: E402
import scenes  # noqa: E402


def timed(ctx, cam, p, d_rgb, reps=3):
    ctx.render_device(cam, p, d_rgb.data_ptr())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ctx.render_device(cam, p, d_rgb.data_ptr())
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    ctx = rtamd.Context(0)
    cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
    for n, grid, W, H, spp in ((500, 11, 1920, 1080, 64), (2000, 23, 1920, 1080, 16), (10000, 50, 1920, 1080, 16)):
        sc = scenes.random_spheres(n, grid_half=grid)
        ctx.set_scene(sc)
        p = rtamd.params(W, H, spp, 8, seed=1)
        d_rgb = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX)
        ctx.reset_stats()
        t_m = timed(ctx, cam, p, d_rgb)
        rays = ctx.stats().rays / 4
        ref = d_rgb.clone()
        print(f"n={n}: matrix {t_m:8.2f} ms  {rays / t_m / 1e3:9.1f} Mrays/s", flush=True)
        for leaf in (1, 2, 4, 8):
            ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, leaf)
            ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
            t_b = timed(ctx, cam, p, d_rgb)
            same = bool(torch.equal(ref, d_rgb))
            print(f"n={n}: bvh leaf {leaf}: {t_b:8.2f} ms  {rays / t_b / 1e3:9.1f} Mrays/s  x{t_m / t_b:5.2f}  same bits: {same}",
                  flush=True)
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_AUTO)
    ctx.close()


if __name__ == "__main__":
    main()
