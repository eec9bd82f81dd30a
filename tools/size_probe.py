import os, sys
sys.path.insert(0, "ray-tracer_amd/python")
import torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(3840*2160*3, dtype=torch.int32, device="cuda")
for depth in (1, 8):
    for (W,H) in ((64,64),(240,135),(480,270),(960,540),(1920,1080),(3840,2160)):
        p = rtamd.params(W, H, 1, depth, seed=1)
        ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
        ctx.reset_stats()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        st = ctx.stats()
        tiles = ((W+7)//8)*((H+7)//8)
        print(f"depth {depth} {W}x{H}: {min(ts)*1e3:8.1f} us  tiles {tiles:6d}  rays {st.rays//10:9d}  {st.rays/10/min(ts)/1e6:6.2f} Grays/s  passes {st.wave_passes//10}")
