"""Kernel time of ONE rank's share of the bench frame under an N-way split (tiles or samples), on one
GPU: what strong scaling can reach at best if the exchange is hidden.  Usage: python tools/share_time.py"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np
import torch

import dist_render
import rtamd
import scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
full = None
for world in (1, 2, 4, 8):
    for partition in ("tiles", "samples"):
        worst = 0.0
        for rank in range(world):
            p = dist_render.rank_params(1920, 1080, 64, 8, 1, partition, rank, world)
            times = []
            for _ in range(6):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.accumulate_device(cam, p, d.data_ptr())
                e1.record()
                torch.cuda.synchronize()
                times.append(e0.elapsed_time(e1))
            worst = max(worst, float(np.median(times[1:])))
        if full is None:
            full = worst
        print(f"N={world} {partition:8s} slowest rank's kernel {worst:7.3f} ms  -> speed-up at best {full / worst:5.2f} of {world}", flush=True)
ctx.close()
