"""Times the reference's shipped configuration (1024x576, 100 spp, depth 10, its four spheres)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["RTAMD_LIBRARY"] = os.path.join(ROOT, "tools", sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd
ctx = rtamd.Context(0)
ctx.set_scene(rtamd.default_scene())
cam = rtamd.default_camera()
W, H = 1024, 576
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
p = rtamd.params(W, H, 100, 10, seed=1)
best = 1e9
for it in range(8):
    ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
    ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
print(f"{os.environ.get('RTAMD_LIBRARY', 'shipped'):40s} C1: {best * 1e3:7.2f} ms  rays {ctx.stats().rays}  checksum {int(d.sum())}")
