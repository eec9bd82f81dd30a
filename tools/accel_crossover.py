"""Scene size from which the hierarchy scan beats the list scans (run on the GPU box)."""
import os
import sys
import time

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


ctx = rtamd.Context(0)
W, H, spp = 1920, 1080, 16
d_rgb = torch.empty(W * H * 3, dtype=torch.uint8, device="cuda")
cases = [("default4", rtamd.default_scene(), scenes.DEFAULT_CAMERA)]
for n, g in ((8, 2), (16, 2), (32, 3), (64, 4), (128, 6), (256, 8)):
    cases.append((f"random{n}", scenes.random_spheres(n, grid_half=g), scenes.RANDOM_SPHERES_CAMERA))
for name, sc, cam in cases:
    ctx.set_scene(sc)
    c = rtamd.make_camera(*cam)
    p = rtamd.params(W, H, spp, 8, seed=1)
    row = []
    for filt in (rtamd.FILTER_CONSERVATIVE, rtamd.FILTER_MATRIX, rtamd.FILTER_BVH):
        ctx.set_option(rtamd.OPT_FILTER, filt)
        row.append(timed(ctx, c, p, d_rgb))
    print(f"{name:10s} vector {row[0]:7.2f} ms  matrix {row[1]:7.2f} ms  hierarchy {row[2]:7.2f} ms", flush=True)
ctx.close()
