"""Bench frame time against the hierarchy's tuning options.  Usage: python tools/tune_sweep.py   (on a GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import rtamd
import scenes

ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")


def frame_ms(p, reps=8):
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.accumulate_device(cam, p, d.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return min(ts)


for n, grid, spp in ((500, 11, 64), (10000, 50, 64)):
    ctx.set_scene(scenes.random_spheres(n, grid_half=grid))
    p = rtamd.params(1920, 1080, spp, 8, seed=1)
    print(f"{n} spheres:", flush=True)
    for leaf in (2, 3, 4, 6, 8):
        ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, leaf)
        print(f"  leaf {leaf}: {frame_ms(p):.3f} ms", flush=True)
    ctx.set_option(rtamd.OPT_BVH_LEAF_SPHERES, 4)
    for lanes in (0, 8, 12, 16, 20, 24, 32, 40):
        ctx.set_option(rtamd.OPT_BVH_SUSPEND_LANES, lanes)
        print(f"  suspend at {lanes}: {frame_ms(p):.3f} ms", flush=True)
    ctx.set_option(rtamd.OPT_BVH_SUSPEND_LANES, 16)
    for chunks in (4, 8, 16, 32, 64):
        ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, chunks)
        print(f"  chunks per wave {chunks}: {frame_ms(p):.3f} ms", flush=True)
    ctx.set_option(rtamd.OPT_CHUNKS_PER_WAVE, 32)
    for nodes in (0, 512, 1024, 2048):
        ctx.set_option(rtamd.OPT_BVH_LDS_NODES, nodes)
        print(f"  nodes in LDS up to {nodes}: {frame_ms(p):.3f} ms", flush=True)
    ctx.set_option(rtamd.OPT_BVH_LDS_NODES, 512)
