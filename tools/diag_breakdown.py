"""Times the bench frame with the shipped library and with a diagnostic build (tools/librt_diag.so)."""
import os, sys, time, subprocess
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
s = torch.cuda.current_stream().cuda_stream
for depth in (8, 1):
    p = rtamd.params(W, H, 64, depth, seed=1)
    best = 1e9
    for it in range(6):
        ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr(), stream=s); torch.cuda.synchronize(); best = min(best, time.time() - t0)
    st = ctx.stats()
    print(f"{os.environ.get('RTAMD_LIBRARY','shipped'):40s} depth {depth}: {best*1e3:7.2f} ms  rays {st.rays}  ps/ray {best*1e12/st.rays:6.1f}")
