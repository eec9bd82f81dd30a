"""Frame time of the bench frame against resident workgroups per CU (one workgroup = one wave per
SIMD), with the hierarchy's nodes in LDS and read through the caches.
Usage: python tools/occ_probe.py   (on a GPU box; RTAMD_LIBRARY selects another build)"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import numpy as np
import torch

import rtamd
import scenes

ctx = rtamd.Context(0)
cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
d = torch.zeros(1920 * 1080 * 3, dtype=torch.int32, device="cuda")
p = rtamd.params(1920, 1080, 64, 8, seed=1)
for lds_nodes in (512, 0):
    ctx.set_option(rtamd.OPT_BVH_LDS_NODES, lds_nodes)
    ctx.set_scene(scenes.random_spheres(500))
    for wg in (1, 2, 3, 4, 5, 6, 7, 8):
        ctx.set_option(rtamd.OPT_WORKGROUPS_PER_CU, wg)
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ctx.accumulate_device(cam, p, d.data_ptr())
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"nodes in LDS up to {lds_nodes:3d}: at most {wg} workgroups per CU: {np.median(ts[1:]):7.3f} ms", flush=True)
ctx.close()
