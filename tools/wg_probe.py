import os, sys
sys.path.insert(0, "ray-tracer_amd/python")
import torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920*1080*3, dtype=torch.int32, device="cuda")
for spp in (1, 2, 4):
    for wg in (0, 1, 2, 3, 4):
        ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, wg)
        p = rtamd.params(1920, 1080, spp, 8, seed=1)
        ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"spp {spp} wg/cu {wg}: {min(ts)*1e3:8.1f} us")
# empty-ish launch: depth 1, tiny frame
ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, 0)
for (W,H,spp,depth) in ((64,64,1,1),(1920,1080,1,1),(1920,1080,1,2),(1920,1080,1,4)):
    p = rtamd.params(W, H, spp, depth, seed=1)
    ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
    ts = []
    for _ in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{W}x{H} spp {spp} depth {depth}: {min(ts)*1e3:8.1f} us")
