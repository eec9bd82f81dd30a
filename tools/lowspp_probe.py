"""Calls of few samples per pixel (passes of a progressive render): time, rate and lane utilisation against
RT_OPT_DIRECT_ADD_SAMPLES.  Usage: python tools/lowspp_probe.py   (on a GPU box)"""
import os
import sys
import zlib

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import rtamd
import scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
for spp in (1, 2, 4, 8, 16):
    for direct in (0, 1, 2, 4):
        ctx.set_option(rtamd.OPT_DIRECT_ADD_SAMPLES, direct)
        p = rtamd.params(1920, 1080, spp, 8, seed=1)
        d.zero_()
        ctx.accumulate_device(cam, p, d.data_ptr())
        torch.cuda.synchronize()
        crc = zlib.crc32(d.cpu().numpy().tobytes())
        ctx.reset_stats()
        ts = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.accumulate_device(cam, p, d.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        st = ctx.stats()
        print(f"spp {spp:3d} direct<= {direct:2d}: {min(ts) * 1e3:8.1f} us  {st.rays / 10 / min(ts) / 1e6:6.2f} Grays/s  "
              f"lane util {st.rays / 64 / st.wave_passes:.3f}  crc {crc:08x}", flush=True)
