"""Where a wave's lifetime goes, phase by phase (counting build, s_memtime stamps), for a few call shapes.
Usage: python tools/phase_time.py [hierarchy|matrix|vector]   (on a GPU box)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch

import rtamd
import scenes

ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
ctx.set_option(rtamd.OPT_FILTER, {"hierarchy": rtamd.FILTER_BVH, "matrix": rtamd.FILTER_MATRIX, "vector": rtamd.FILTER_CONSERVATIVE}[
    sys.argv[1] if len(sys.argv) > 1 else "hierarchy"])
ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
if os.environ.get("PT_WG_PER_CU"):  # e.g. 1: one wave per SIMD, nothing hides a phase's latencies
    ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, int(os.environ["PT_WG_PER_CU"]))
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
for W, H, spp, depth in ((1920, 1080, 1, 1), (1920, 1080, 1, 8), (1920, 1080, 4, 8), (1920, 1080, 64, 8)):
    p = rtamd.params(W, H, spp, depth, seed=1)
    ctx.accumulate_device(cam, p, d.data_ptr())
    torch.cuda.synchronize()
    ctx.reset_stats()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ctx.accumulate_device(cam, p, d.data_ptr())
    e1.record()
    torch.cuda.synchronize()
    prof = ctx.work_profile()
    cyc = prof.pop("cycles")
    waves = max(cyc.pop("waves"), 1)
    passes = prof["pass"][0]
    life = cyc["life"] / waves
    print(f"{W}x{H} spp {spp} depth {depth}: {e0.elapsed_time(e1) * 1e3:.0f} us, {waves} waves, {passes / waves:.1f} passes per wave, "
          f"lifetime {life:.0f} cycles per wave; share of lifetime: " +
          ", ".join(f"{k} {v / waves / life:.3f}" for k, v in cyc.items() if k != "life"))
