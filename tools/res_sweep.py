"""Frame sizes and sample counts against work queue granularity (RT_OPT_CHUNKS_PER_WAVE): ps per ray.

The reference's own scene (short list, vector scan) and the 500-sphere bench scene (hierarchy)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes

ctx = rtamd.Context(0)
SETS = (
    ("reference scene", rtamd.default_scene(), rtamd.default_camera(), 10,
     ((512, 288, 100), (1024, 576, 100), (1024, 576, 400), (1920, 1080, 64), (3840, 2160, 16))),
    ("500 random spheres", scenes.random_spheres(500), rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA), 8,
     ((480, 270, 64), (960, 540, 64), (1920, 1080, 64), (1920, 1080, 8), (3840, 2160, 16))),
)
GRAN = tuple(int(v) for v in sys.argv[1:]) or (0, 16, -16, -32)  # negative: tiles are not cut into strips
for title, scene, cam, depth, frames in SETS:
    ctx.set_scene(scene)
    print(title, flush=True)
    for W, H, spp in frames:
        d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
        p = rtamd.params(W, H, spp, depth, seed=1)
        line = f"  {W}x{H} {spp:4d} spp, {((W + 7) // 8) * ((H + 7) // 8):6d} tiles:"
        for g in GRAN:
            ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, abs(g))
            ctx.set_option(rtamd.OPT_SUBTILES, 1 if g < 0 else 4)
            best = 1e9
            for it in range(6):
                ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
                ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
            st = ctx.stats()
            line += f"  [{g}] {best * 1e3:6.2f} ms {best * 1e12 / st.rays:5.1f} ps/ray"
        print(line, flush=True)
