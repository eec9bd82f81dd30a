"""When the waves of one launch start, find the work queue empty and end (counting build; cycles of the
chip-wide 100 MHz clock after the first wave's start): how much of a launch is ramp and tail.
Usage: python tools/wave_timeline.py [hierarchy|matrix|vector]   (on a GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import dist_render
import rtamd
import scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
ctx.set_option(rtamd.OPT_FILTER, {"hierarchy": rtamd.FILTER_BVH, "matrix": rtamd.FILTER_MATRIX, "vector": rtamd.FILTER_CONSERVATIVE}[
    sys.argv[1] if len(sys.argv) > 1 else "hierarchy"])
ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
shapes = [("1080p 64 spp", rtamd.params(1920, 1080, 64, 8, seed=1)),
          ("1080p 16 spp", rtamd.params(1920, 1080, 16, 8, seed=1)),
          ("1080p 8 spp", rtamd.params(1920, 1080, 8, 8, seed=1)),
          ("1/8 of the tiles", dist_render.rank_params(1920, 1080, 64, 8, 1, "tiles", 3, 8)),
          ("1080p 1 spp", rtamd.params(1920, 1080, 1, 8, seed=1))]
for name, p in shapes:
    for _ in range(2):
        ctx.accumulate_device(cam, p, d.data_ptr())
        torch.cuda.synchronize()
        ctx.reset_stats()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.accumulate_device(cam, p, d.data_ptr())
        e1.record()
        torch.cuda.synchronize()
    prof = ctx.work_profile()
    tl, waves = prof["timeline"], prof["cycles"]["waves"]
    span = tl["end"]["last"]
    ms = e0.elapsed_time(e1)
    print(f"{name:18s} {ms:7.3f} ms, {waves} waves, span {span / 100:.1f} us")
    for k in ("start", "queue_empty", "end"):
        v = tl[k]
        print(f"    {k:12s} first {v['first'] / span:6.3f}  mean {v['mean'] / span:6.3f}  last {v['last'] / span:6.3f}   of the span")
    print(f"    mean lifetime {(tl['end']['mean'] - tl['start']['mean']) / span:.3f} of the span; "
          f"after the queue ran dry a wave lives {(tl['end']['mean'] - tl['queue_empty']['mean']) / span:.3f} of the span on average", flush=True)
ctx.close()
