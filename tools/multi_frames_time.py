"""rt_multi on one device: wall time per frame of a sequence of small frames, enqueued one frame per call
(rt_multi_render_device) and eight per call (rt_multi_render_frames_device: one trace launch for the eight).
Usage: python tools/multi_frames_time.py   (on a GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np
import torch

import rtamd
import scenes

N_FRAMES = 64
m = rtamd.Multi([0], rtamd.EXCHANGE_PEER_COPY)
m.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
for spp in (1, 4, 16):
    p = rtamd.params(1920, 1080, spp, 8, seed=1)
    outs = [torch.zeros(1920 * 1080 * 3, dtype=torch.uint8, device="cuda") for _ in range(8)]
    for per_call in (1, 8):
        for partition, name in ((rtamd.PARTITION_TILES, "tiles"), (rtamd.PARTITION_SAMPLES, "samples")):
            ts = []
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(0, N_FRAMES, per_call):
                    if per_call == 1:
                        m.render_device(cam, p, partition, outs[k % 8].data_ptr())
                    else:
                        m.render_frames_device([(cam, p, outs[j].data_ptr()) for j in range(per_call)], partition)
                m.synchronize()
                ts.append((time.perf_counter() - t0) / N_FRAMES * 1e3)
            print(f"1080p / {spp:2d} spp, {per_call} frame(s) per call, {name:7s}: {float(np.median(ts[1:])):6.3f} ms per frame", flush=True)
m.close()
