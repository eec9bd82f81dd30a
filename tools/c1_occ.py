"""The reference's shipped frame against the cap on resident workgroups per CU."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd
ctx = rtamd.Context(0)
ctx.set_scene(rtamd.default_scene())
cam = rtamd.default_camera()
W, H = 1024, 576
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
p = rtamd.params(W, H, 100, 10, seed=1)
for cap in (0, 8, 7, 6, 5, 4, 3):
    ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, cap)
    best = 1e9
    for it in range(5):
        torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    print(f"workgroups per CU <= {cap}: {best * 1e3:6.2f} ms", flush=True)
