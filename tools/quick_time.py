"""Kernel time of the main configurations, with a checksum of the sums so that a build that changes
results shows at once.  Usage: python tools/quick_time.py [reps]   (on a GPU box)"""
import os
import sys
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np
import torch

import rtamd
import scenes

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
cam3 = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
CASES = [
    ("c3 hierarchy 1080p/64", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 64, 8)),
    ("c3 matrix    1080p/64", scenes.random_spheres(500), cam3, rtamd.FILTER_MATRIX, (1920, 1080, 64, 8)),
    ("c1 shipped 1024x576/100", rtamd.default_scene(), rtamd.default_camera(), rtamd.FILTER_AUTO, (1024, 576, 100, 10)),
    ("c5 hierarchy 1080p/64", scenes.random_spheres(10000, grid_half=50), cam3, rtamd.FILTER_AUTO, (1920, 1080, 64, 8)),
    ("c3 hierarchy 1080p/4", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 4, 8)),
    ("c3 hierarchy 1080p/1", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 1, 8)),
    ("c3 hierarchy 1080p/8", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 8, 8)),
    ("c3 hierarchy 1080p/16", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 16, 8)),
    # one rank's share of the 64 spp frame split over 8 devices by tiles
    ("c3 hierarchy 1080p/64 tiles 3 mod 8", scenes.random_spheres(500), cam3, rtamd.FILTER_AUTO, (1920, 1080, 64, 8),
     dict(tile_first=3, tile_stride=8)),
]
only = os.environ.get("QT_ONLY")
ctx = rtamd.Context(0)
for name, scene, cam, filt, (W, H, spp, depth), *share in CASES:
    if only and only not in name:
        continue
    ctx.set_option(rtamd.OPT_FILTER, filt)
    ctx.set_scene(scene)
    p = rtamd.params(W, H, spp, depth, seed=1, **(share[0] if share else {}))
    d = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
    ctx.reset_stats()
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    rays = ctx.stats().rays
    crc = zlib.crc32(d.cpu().numpy().tobytes())
    times = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.accumulate_device(cam, p, d.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    t = float(np.median(times))
    print(f"{name:36s} {t:8.3f} ms (min {min(times):.3f})  {rays / t / 1e6:8.2f} Grays/s  rays {rays}  crc {crc:08x}", flush=True)
ctx.close()
