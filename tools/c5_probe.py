"""The 10 000-sphere frame (1080p, hierarchy walked in memory): time of the plain build, then the counting
build's sections per pass and phase shares of a wave's lifetime, next to the 500-sphere frame's.
Usage: python tools/c5_probe.py [spp]   (on a GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import rtamd
import scenes

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
p = rtamd.params(1920, 1080, spp, 8, seed=1)


def timed(ctx, reps=3):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ctx.accumulate_device(cam, p, d.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


for n in (500, 10000):
    ctx = rtamd.Context(0)
    ctx.set_scene(scenes.random_spheres(n, grid_half=50 if n > 2000 else 11))
    ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_BVH)
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    ctx.reset_stats()
    ms = timed(ctx)
    ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    ctx.reset_stats()
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    st = ctx.stats()
    prof = ctx.work_profile()
    cyc = prof.pop("cycles")
    prof.pop("timeline", None)
    waves = max(cyc.pop("waves"), 1)
    passes = prof["pass"][0]
    life = cyc["life"] / waves
    print(f"--- {n} spheres, {spp} spp: {ms:.2f} ms, {st.rays} rays, {st.rays / ms / 1e6:.1f} Grays/s; "
          f"{passes / waves:.1f} passes per wave, {st.rays / passes:.1f} rays per pass, {life / (passes / waves):.0f} cycles per pass")
    print("    share of lifetime: " + ", ".join(f"{k} {v / waves / life:.3f}" for k, v in cyc.items() if k != "life"))
    for name, (w, lanes) in prof.items():
        if w:
            print(f"    {name:14s} {w / passes:9.3f} per pass {lanes / w:6.1f} lanes {lanes / st.rays:8.3f} per ray")
    del ctx
