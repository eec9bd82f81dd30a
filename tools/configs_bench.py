"""Times the BASELINE configurations that fit one GPU (C1, C2, C3, C5 at reduced spp)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes

ctx = rtamd.Context(0)
s = torch.cuda.current_stream().cuda_stream

def run(name, scene, cam, W, H, spp, depth, reps=3):
    ctx.set_scene(scene)
    d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
    p = rtamd.params(W, H, spp, depth, seed=1)
    best = 1e9
    for _ in range(reps):
        ctx.reset_stats(); d.zero_(); torch.cuda.synchronize(); t0 = time.time()
        ctx.accumulate_device(cam, p, d.data_ptr(), stream=s); torch.cuda.synchronize()
        best = min(best, time.time() - t0)
    st = ctx.stats()
    n = len(scene)
    print(f"{name:34s} {best*1e3:9.2f} ms  rays {st.rays:>12d}  {st.rays/best/1e6:9.1f} Mrays/s  "
          f"{st.rays*n/best/1e12:6.2f} Ttests/s  frac {st.rays*n*25/best/157.3e12:.3f}  util {st.rays/64/max(st.wave_passes,1):.3f}", flush=True)

filt = int(sys.argv[1]) if len(sys.argv) > 1 else 2
ctx.set_option(rtamd.OPT_FILTER, filt)
print("filter", filt)
run("C1 default 1024x576 100spp d10", rtamd.default_scene(), rtamd.default_camera(), 1024, 576, 100, 10)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
run("C3 500 spheres 1080p 64spp d8", scenes.random_spheres(500), cam, 1920, 1080, 64, 8)
run("C3' 500 spheres 1080p 64spp d1", scenes.random_spheres(500), cam, 1920, 1080, 64, 1)
run("C5' 10k spheres 1080p 8spp d8", scenes.random_spheres(10000, grid_half=50), cam, 1920, 1080, 8, 8, reps=2)
run("2000 spheres 1080p 16spp d8", scenes.random_spheres(2000, grid_half=22), cam, 1920, 1080, 16, 8, reps=2)
t0 = time.time(); ctx.set_scene(scenes.random_spheres(500)); h = torch.zeros(1920 * 1080, dtype=torch.int32, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.time(); ctx.primary_hits_device(cam, 1920, 1080, h.data_ptr(), stream=s); torch.cuda.synchronize(); dt = time.time() - t0
print(f"C2 primary hit map 1080p 500 spheres: {dt*1e3:.3f} ms")
if len(sys.argv) > 2:
    for cap in (256, 512, 1024, 2048):
        ctx.set_option(rtamd.OPT_TILE_SPHERES, cap)
        run(f"C5' 10k spheres tile {cap}", scenes.random_spheres(10000, grid_half=50), cam, 1920, 1080, 8, 8, reps=2)
    ctx.set_option(rtamd.OPT_TILE_SPHERES, 1024)
    for wg in (2, 3, 4):
        ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, wg)
        run(f"C3 wg/CU {wg}", scenes.random_spheres(500), cam, 1920, 1080, 64, 8)
