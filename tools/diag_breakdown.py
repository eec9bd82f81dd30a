"""Timing breakdown of the trace kernel with the diagnostic library (phase 2 skipped = all rays miss)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["RTAMD_LIBRARY"] = os.path.join(ROOT, "tools", "librt_diag.so")
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for filt, depth, name in ((1, 8, "full"), (2, 8, "phase-1 only (all miss)"), (1, 1, "full, depth 1"), (2, 1, "phase-1 only, depth 1")):
    ctx._check(ctx._L.rt_set_option(ctx._h, 0, filt))
    p = rtamd.params(W, H, 64, depth, seed=1)
    for it in range(3):
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr(), stream=s); torch.cuda.synchronize(); dt = time.time() - t0
    st = ctx.stats()
    print(f"{name:28s} {dt*1e3:7.2f} ms  rays {st.rays:>10d}  ns/wave-pass {dt*1e9/st.wave_passes*1024:8.1f} (per SIMD)  util {st.rays/64/st.wave_passes:.3f}")
