import sys, zlib
sys.path.insert(0, "ray-tracer_amd/python")
import numpy as np, torch, rtamd, scenes
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920*1080*3, dtype=torch.int32, device="cuda")
for name, flags in (("dielectric", rtamd.FLAG_DIELECTRIC), ("float shading", rtamd.FLAG_FLOAT_SHADING), ("both", rtamd.FLAG_DIELECTRIC | rtamd.FLAG_FLOAT_SHADING)):
    p = rtamd.params(1920, 1080, 64, 8, seed=1, flags=flags)
    d.zero_(); ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
    crc = zlib.crc32(d.cpu().numpy().tobytes())
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{name:14s} {np.median(ts[1:]):.3f} ms crc {crc:08x}", flush=True)
