"""The bench frame with the opt-in flags (extended kernel): ms per frame.  Usage: ext_time.py [library under tools/]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["RTAMD_LIBRARY"] = os.path.join(ROOT, "tools", sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
for flags in (0, 1, 4, 5):
    p = rtamd.params(W, H, 64, 8, seed=1, flags=flags)
    best = 1e9
    for it in range(5):
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    st = ctx.stats()
    print(f"{os.environ.get('RTAMD_LIBRARY', 'shipped'):36s} flags {flags}: {best * 1e3:7.2f} ms  rays {st.rays}  {best * 1e12 / st.rays:5.1f} ps/ray", flush=True)
