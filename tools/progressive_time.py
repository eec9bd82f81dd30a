"""Wall time of the host-facing entry points: rt_render of a frame at once and as the reference's four column
strips (Source/main.cpp:647-659), and rt_render_progressive (frame delivered to host memory after every
pass, callback called) for passes of 1, 4 and 16 samples per pixel.
Usage: python tools/progressive_time.py   (on a GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np

import rtamd
import scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
W, H, SPP = 1920, 1080, 64
p = rtamd.params(W, H, SPP, 8, seed=1)
rgb = np.zeros((H, W, 3), dtype=np.uint8)


def wall(fn, reps=5):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts[1:])) * 1e3


whole = ctx.render(cam, p)
print(f"rt_render, {SPP} spp at once: {wall(lambda: ctx.render(cam, p, rgb=rgb)):7.2f} ms", flush=True)


def strips():
    for k in range(4):
        ctx.render(cam, p, tile=(k * (W // 4), 0, W // 4, H), rgb=rgb)


rgb[:] = 0
ms = wall(strips)
print(f"rt_render, four column strips one after the other: {ms:7.2f} ms, frame {'==' if np.array_equal(rgb, whole) else '!='} the whole call's",
      flush=True)
import torch

pinned = torch.empty((H, W, 3), dtype=torch.uint8, pin_memory=True).numpy()  # the caller's frame in pinned host memory
for per_launch, frame in ((1, rgb), (0, rgb), (1, pinned), (0, pinned)):  # every pass a launch of its own / small passes share launches
    ctx.set_option(rtamd.OPT_PROGRESSIVE_PASSES_PER_LAUNCH, per_launch)
    for per_pass in (16, 4, 1):
        ms = wall(lambda: ctx.render_progressive(cam, p, per_pass, lambda d, t, img: False, rgb=frame))
        same = np.array_equal(frame, whole)
        print(f"rt_render_progressive, {per_pass:2d} spp per pass ({SPP // per_pass} passes), "
              f"{'one pass' if per_launch else 'up to 16 spp'} per launch, {'pinned' if frame is pinned else 'pageable'} frame: {ms:7.2f} ms, "
              f"{ms / (SPP // per_pass):6.3f} ms per pass, final frame {'==' if same else '!='} rt_render's", flush=True)
ctx.close()
