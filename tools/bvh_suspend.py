"""Hierarchy scan on the bench frames: time against the suspend threshold."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch  # noqa: E402,F401
import rtamd  # noqa: E402
import scenes  # noqa: E402

ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
for n, grid, spp in ((500, 11, 64), (10000, 50, 16)):
    ctx.set_scene(scenes.random_spheres(n, grid_half=grid))
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    p = rtamd.params(W, H, spp, 8, seed=1)
    row = []
    for thr in (0, 2, 4, 8, 12, 16, 24, 32, 48):
        ctx.set_option(rtamd.OPT_BVH_SUSPEND_LANES, thr)
        best = 1e9
        for it in range(3):
            ctx.reset_stats()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        st = ctx.stats()
        row.append(f"{thr}:{best * 1e3:6.2f}ms/{st.rays / (64.0 * st.wave_passes):.3f}")
    print(f"n={n}  suspend:ms/lane-utilisation  " + "  ".join(row), flush=True)
ctx.close()
