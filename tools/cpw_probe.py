import sys
sys.path.insert(0, "ray-tracer_amd/python")
import numpy as np, torch, rtamd, scenes, dist_render
ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(3840*2160*3, dtype=torch.int32, device="cuda")
def t(p, reps=8):
    ts=[]
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ctx.accumulate_device(cam, p, d.data_ptr()); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))
ctx.set_scene(scenes.random_spheres(500))
cases = {"1080p/64": rtamd.params(1920,1080,64,8,seed=1), "1080p/16": rtamd.params(1920,1080,16,8,seed=1),
         "N=8 share": dist_render.rank_params(1920,1080,64,8,1,"tiles",0,8), "N=8 samples": dist_render.rank_params(1920,1080,64,8,1,"samples",0,8),
         "4K/64": rtamd.params(3840,2160,64,8,seed=1), "640x360/64": rtamd.params(640,360,64,8,seed=1)}
for c in (16, 24, 32, 36, 40, 48, 64):
    ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, c)
    print(c, "  ".join(f"{k}: {t(p):.3f}" for k,p in cases.items()), flush=True)
