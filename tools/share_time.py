"""One rank's share of the weak-scaling bench frame on one GPU: a 64-spp slice of every pixel against
1/N of the 8x8 tiles at 64*N spp (same number of paths)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch, rtamd, scenes, dist_render

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
for world in (1, 2, 4, 8):
    for partition in ("samples", "tiles"):
        worst = 0.0
        for rank in range(world):
            p = dist_render.rank_params(W, H, 64 * world, 8, 1, partition, rank, world)
            best = 1e9
            for _ in range(3):
                ctx.reset_stats(); torch.cuda.synchronize(); t0 = time.time()
                ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize(); best = min(best, time.time() - t0)
            worst = max(worst, best)
        print(f"world {world} {partition:8s}: slowest rank {worst * 1e3:6.2f} ms", flush=True)
