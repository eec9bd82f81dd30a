"""What a launch that carries several jobs (rt_accumulate_batch_device) saves over one launch per job:
kernel time per job for one rank's share of the bench frame under an N-way split, for whole low-sample
frames, and for the reference's shipped frame, in launches of 1, 2, 4 and 8 jobs (every job a frame of its
own, in its own sums).  Usage: python tools/batch_time.py   (on a GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np
import torch

import dist_render
import rtamd
import scenes

ctx = rtamd.Context(0)
cam3 = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H = 1920, 1080
bufs = [torch.zeros(W * H * 3, dtype=torch.int32, device="cuda") for _ in range(rtamd.MAX_JOBS)]


def timed(jobs, reps=8):
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.accumulate_batch_device(jobs)
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    return float(np.median(times[1:]))


def sweep(name, make_params, cam):
    base = None
    for n in (1, 2, 4, 8):
        t = timed([(cam, make_params(), bufs[k].data_ptr()) for k in range(n)]) / n
        base = base or t
        print(f"{name:44s} {n} jobs per launch: {t:7.3f} ms per job ({t / base:5.3f} of one per launch)", flush=True)


ctx.set_scene(scenes.random_spheres(500))
full = timed([(cam3, rtamd.params(W, H, 64, 8, seed=1), bufs[0].data_ptr())])
print(f"bench frame, one job: {full:.3f} ms", flush=True)
for world in (2, 4, 8):
    for partition in ("tiles", "samples"):
        sweep(f"share of rank 1 of {world}, by {partition}", lambda: dist_render.rank_params(W, H, 64, 8, 1, partition, 1, world), cam3)
for spp in (1, 4, 16, 64):
    sweep(f"1080p frame of {spp} spp", lambda: rtamd.params(W, H, spp, 8, seed=1), cam3)
ctx.set_scene(rtamd.default_scene())
sweep("shipped 1024x576 / 100 spp / depth 10", lambda: rtamd.params(1024, 576, 100, 10, seed=1), rtamd.default_camera())
ctx.close()
