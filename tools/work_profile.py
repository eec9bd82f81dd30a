"""Where the hierarchy kernel's waves spend their passes: per section of the kernel, how often a wave
executed it per pass and with how many of its 64 lanes enabled (counting build, RT_OPT_COUNT_WORK).
Usage: python tools/work_profile.py [scan [spheres]]   (on a GPU box; bench frame: 1080p, 64 spp, depth 8,
500 spheres unless given; 10000 = the 10k-sphere configuration)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import rtamd
import scenes

scan = {"hierarchy": rtamd.FILTER_BVH, "matrix": rtamd.FILTER_MATRIX, "vector": rtamd.FILTER_CONSERVATIVE}[
    sys.argv[1] if len(sys.argv) > 1 else "hierarchy"]
ctx = rtamd.Context(0)
n_spheres = int(sys.argv[2]) if len(sys.argv) > 2 else 500
ctx.set_scene(scenes.random_spheres(n_spheres, grid_half=50 if n_spheres > 2000 else 11))
ctx.set_option(rtamd.OPT_FILTER, scan)
ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
ctx.reset_stats()
ctx.accumulate_device(cam, rtamd.params(1920, 1080, 64, 8, seed=1), d.data_ptr())
torch.cuda.synchronize()
st = ctx.stats()
prof = ctx.work_profile()
prof.pop("cycles", None)
prof.pop("timeline", None)
passes = prof["pass"][0]
print(f"rays {st.rays}  wave passes {passes}  rays per pass {st.rays / passes:.1f}")
print(f"{'section':14s} {'per pass':>9s} {'lanes on':>9s} {'lane-executions per ray':>24s}")
for name, (waves, lanes) in prof.items():
    if waves:
        print(f"{name:14s} {waves / passes:9.3f} {lanes / waves:9.1f} {lanes / st.rays:24.3f}")
