"""Hierarchy scan on the bench frame: time against resident workgroups per CU and leaf size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1:
    os.environ["RTAMD_LIBRARY"] = os.path.join(ROOT, "tools", sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))
import torch  # noqa: E402,F401
import rtamd  # noqa: E402
import scenes  # noqa: E402

ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
for n, grid, spp in ((500, 11, 64), (10000, 50, 16)):
    ctx.set_scene(scenes.random_spheres(n, grid_half=grid))
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    p = rtamd.params(W, H, spp, 8, seed=1)
    for leaf, lds in ((4, 0), (4, 512), (8, 512)):
        ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, leaf)
        ctx.set_option(rtamd.OPT_BVH_LDS_NODES, lds)
        row = []
        for wg in (3, 4, 5, 6, 8):
            ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, wg)
            best = 1e9
            for it in range(6):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ctx.accumulate_device(cam, p, d.data_ptr()); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            row.append(f"{wg}:{best * 1e3:6.2f}")
        print(f"n={n} leaf={leaf} lds_nodes<={lds}  ms by workgroups/CU  " + "  ".join(row), flush=True)
ctx.close()
