import sys
sys.path.insert(0, "ray-tracer_amd/python")
import numpy as np, torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920*1080*3, dtype=torch.int32, device="cuda")
p = rtamd.params(1920, 1080, 64, 8, seed=1)
for wg in (1, 2, 3, 4, 5):
    ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, wg)
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"wg/cu {wg}: {np.median(ts[1:]):.3f} ms", flush=True)
