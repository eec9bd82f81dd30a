"""Host cost of one pipelined step of the multi-GPU path: frames small enough that the kernel takes about
as long as one rank's share of the bench frame on 8 GPUs (1080p at 8 spp = 1/8 of the samples), run
through FramePipeline with the exchange over a one-rank RCCL communicator, against the same frames
traced back to back without it.  If the step time stays at the kernel time, the host side (Python,
launches, the collective call) is hidden behind the device.
Usage: python tools/step_overhead.py   (on a GPU box)"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))
import torch
import torch.distributed as dist

import dist_render
import rtamd
import scenes

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
ctx = rtamd.Context(0)
ctx.set_scene(scenes.random_spheres(500))
tracer = dist_render.DeviceTracer(ctx, rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA))
W, H, STEPS = 1920, 1080, 200
for spp in (8, 16, 64):
    p = dist_render.rank_params(W, H, spp, 8, 1, "tiles", 0, 1)
    for partition, exchange in (("tiles", False), ("tiles", True), ("samples", True)):
        pipe = dist_render.FramePipeline(tracer, p, (W, H, spp), partition, 0, 1, dist, torch, "cuda",
                                         always_exchange=exchange)
        for _ in range(10):
            pipe.submit()
        pipe.finish()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(STEPS):
            pipe.submit()
        t_host = time.perf_counter() - t0
        pipe.finish()
        torch.cuda.synchronize()
        t = time.perf_counter() - t0
        print(f"{spp:3d} spp  {'exchange: ' + partition if exchange else 'no exchange':18s}  {t / STEPS * 1e3:7.3f} ms per step "
              f"(host returned from the submits after {t_host / STEPS * 1e3:6.3f} ms per step)", flush=True)
ctx.close()
dist.destroy_process_group()
