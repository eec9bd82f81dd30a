"""Cost of one accumulate call that has next to nothing to trace, and of small sample counts at 1080p:
device time per call, cost per ray, wave passes and lane utilisation."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
s = torch.cuda.current_stream().cuda_stream
DEPTHS = tuple(int(v) for v in sys.argv[1:]) or (8,)
for depth in DEPTHS:
    for W, H, spp, reps in ((8, 8, 1, 200), (256, 256, 1, 200), (1920, 1080, 1, 50), (1920, 1080, 2, 50), (1920, 1080, 4, 50),
                            (1920, 1080, 8, 50), (1920, 1080, 16, 20), (1920, 1080, 64, 5)):
        d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
        p = rtamd.params(W, H, spp, depth, seed=1)
        for _ in range(3):
            ctx.accumulate_device(cam, p, d.data_ptr(), stream=s)
        torch.cuda.synchronize()
        ctx.reset_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ctx.accumulate_device(cam, p, d.data_ptr(), stream=s)
        e1.record()
        torch.cuda.synchronize()
        st = ctx.stats()
        rays, passes = st.rays / reps, st.wave_passes / reps
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"depth {depth} {W}x{H} {spp:3d} spp: {us:8.1f} us/call, {us * 1e6 / max(rays, 1):7.1f} ps/ray, "
              f"{passes:9.0f} wave passes, {us * 1e3 / max(passes, 1):6.1f} ns/pass, utilisation {rays / 64 / max(passes, 1):.3f}", flush=True)
