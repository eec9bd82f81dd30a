"""The reference's scene at several frame sizes and sample counts: ps per ray."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd
ctx = rtamd.Context(0)
ctx.set_scene(rtamd.default_scene())
cam = rtamd.default_camera()
for W, H, spp in ((1024, 576, 100), (1024, 576, 64), (1920, 1080, 64), (1920, 1080, 100), (512, 288, 100), (3840, 2160, 16), (1024, 576, 400)):
    d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
    p = rtamd.params(W, H, spp, 10, seed=1)
    best = 1e9
    for it in range(6):
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    st = ctx.stats()
    print(f"{W}x{H} {spp:4d} spp: {best * 1e3:7.2f} ms  {best * 1e12 / st.rays:6.1f} ps/ray  chunks {((W + 7) // 8) * ((H + 7) // 8)}  util {st.rays / 64 / st.wave_passes:.3f}", flush=True)
