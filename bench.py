#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (incl. bounces) of the per-pixel ray-tracing path.

Workload (BASELINE.json configs[2]): the 500-sphere random-spheres scene at
1920x1080, 64 samples per pixel, max depth 8.  One "step" = one frame.

  N = 1   one MI355X traces all 64 samples and resolves the RGB8 frame.
  N > 1   weak scaling, sample-sliced: every rank traces 64 samples of every
          pixel (the frame has 64*N spp), the per-pixel integer sums are
          reduced to rank 0 over RCCL and rank 0 divides and packs the frame.
          (--partition tiles: the 64*N spp are split by interleaved 8x8 tiles
          instead; same work per rank, same reduce.)

Inputs (scene, camera) and outputs (sums, frame) stay in HBM during the timed
region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "ray-tracer_amd", "python"))

import numpy as np
import torch
import torch.distributed as dist

import dist_render
import rtamd
import scenes

W, H, SPP, DEPTH, N_SPHERES = 1920, 1080, 64, 8, 500

# Other BASELINE.json configurations, selectable with --workload (the default and the headline
# number is "c3"): name -> (width, height, frame spp or None for 64 per GPU, spheres, grid, partition)
WORKLOADS = {
    # configs[2]; weak scaling: 64 spp per GPU.  Sample slices: a rank's 1/N of the tiles at 64*N spp
    # measured 0.26 ms slower at N = 8 (tools/share_time.py), about what its smaller exchange saves
    "c3": (1920, 1080, None, 500, 11, "samples"),
    "c4": (3840, 2160, 1024, 500, 11, "samples"),   # configs[3]: 1024 spp split over the GPUs
    "c5": (1920, 1080, 256, 10000, 50, "tiles"),    # configs[4]: 10k spheres, tile-partitioned
}
SCANS = {"auto": rtamd.FILTER_AUTO, "exact": rtamd.FILTER_EXACT, "vector": rtamd.FILTER_CONSERVATIVE,
         "matrix": rtamd.FILTER_MATRIX, "hierarchy": rtamd.FILTER_BVH}
FLOP_PER_TEST = 25.0          # miss path of RaySphereIntersectionTest, Source/main.cpp:475-479
PEAK_FP32_TFLOPS = 157.3      # MI355X FP32, matrix (v_mfma_f32_*_f32) = vector: 256 CU x 128 FMA lanes x 2 x 2.4 GHz


def cpu_baseline(scene):
    """The reference's own code (oracle/_ref/ref_harness, kind "reference") or, if that binary is
    absent, the C restatement (kind "port"), timed on this host on a bounded crop of the workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    x0, xs = 928, 64  # 64 centre columns x 1080 rows x 64 spp = 4.4 M samples: about 12 s on one core
    sample = f"columns {x0}..{x0 + xs - 1} of the 1920x1080/64spp/depth-8 frame, all rows, 1 thread, mt19937 stream"
    ocam = O.camera(*scenes.RANDOM_SPHERES_CAMERA)
    t0 = time.time()
    _, _, rays = O.trace_tile(scene, ocam, W, H, SPP, DEPTH, O.RNG_MT19937, tile=(x0, 0, xs, H))
    port_s = time.time() - t0
    out = {"value": rays / port_s / 1e6, "unit": "Mrays/s", "cores": 1, "kind": "port", "sample": sample,
           "rays": rays, "seconds": port_s}
    if O.have_ref():
        scene_file = os.path.join(ROOT, "tests", "golden", "scene_c3_500.bin")
        cam = scenes.RANDOM_SPHERES_CAMERA
        try:
            txt = O.run_ref("render", scene_file, *cam[0], *cam[1], *cam[2], cam[3], W, H, SPP, DEPTH, x0, xs, "-")
            ref_s = float(re.search(r"seconds ([0-9.]+)", txt).group(1))
            # the restatement reproduces the reference pixel for pixel, hence ray for ray
            out.update({"value": rays / ref_s / 1e6, "kind": "reference", "seconds": ref_s,
                        "port_value": rays / port_s / 1e6})
        except Exception as e:  # keep the port number
            out["reference_error"] = str(e)[:200]
    # all host threads, column strips like TraceScene, one generator per strip
    threads = max(1, min(16, os.cpu_count() or 1))
    import ctypes as C
    rgb = np.zeros((H, W, 3), dtype=np.uint8)
    r = C.c_uint64(0)
    t0 = time.time()
    O.lib().orc_trace_scene_threads(scene.ctypes.data, len(scene), C.byref(ocam), W, H, SPP, DEPTH, W // 2 - 4 * threads,
                                    8 * threads, threads, rgb.ctypes.data, C.byref(r))
    mt_s = time.time() - t0
    out["port_threads"] = {"value": r.value / mt_s / 1e6, "cores": threads,
                           "sample": f"{8 * threads} centre columns, one strip and one generator per thread"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--partition", choices=["samples", "tiles"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scan", choices=sorted(SCANS), default="auto",
                    help="how rays find their candidate spheres (all give the same frame); auto = the library's choice")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global W, H, N_SPHERES
    W, H, frame_spp, N_SPHERES, grid, default_partition = WORKLOADS[args.workload]
    if args.partition is None:
        args.partition = default_partition
    ctx = rtamd.Context(local_rank)
    scene = scenes.random_spheres(N_SPHERES, grid_half=grid)
    ctx.set_scene(scene)
    ctx.set_option(rtamd.OPT_FILTER, SCANS[args.scan])
    scan = args.scan
    if scan == "auto":  # the library's rule (rt_mi355x.h, RT_OPT_FILTER)
        scan = "hierarchy" if N_SPHERES >= 64 else "vector"
    cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
    total_spp = SPP * world if frame_spp is None else frame_spp
    p = dist_render.rank_params(W, H, total_spp, DEPTH, 1, args.partition, rank, world)
    tracer = dist_render.DeviceTracer(ctx, cam)

    d_sums = torch.zeros(W * H * 3, dtype=torch.int32, device="cuda")
    d_rgb = torch.zeros(W * H * 3, dtype=torch.uint8, device="cuda")
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]

    class TimedTracer:
        """Brackets the trace kernel with events on the stream it is launched on."""
        k = None

        def accumulate(self, params, sums):
            if self.k is not None:
                ev0[self.k].record()
            tracer.accumulate(params, sums)
            if self.k is not None:
                ev1[self.k].record()

        resolve = staticmethod(tracer.resolve)

    timed = TimedTracer()

    def step(k=None):
        timed.k = k
        dist_render.trace_frame(timed, p, d_sums, d_rgb, rank, world, dist)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.reset_stats()
    fence()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    fence()
    elapsed = time.perf_counter() - t0

    st = ctx.stats()
    kernel_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))

    # The hierarchy skips most of the reference's ray-sphere tests, so its time says nothing about
    # how well the FP32 pipe is used.  For that, the kernel that does perform every test (the
    # matrix-core list scan) is timed beside it on rank 0, outside the timed region.
    list_ms = None
    if rank == 0 and scan == "hierarchy" and N_SPHERES <= 2000:
        ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tracer.accumulate(p, d_sums)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            tracer.accumulate(p, d_sums)
        e1.record()
        torch.cuda.synchronize()
        list_ms = e0.elapsed_time(e1) / 3.0
        ctx.set_option(rtamd.OPT_FILTER, SCANS[args.scan])
    if world > 1:
        dist.barrier()
    t = torch.tensor([elapsed, float(st.rays), float(st.wave_passes), kernel_ms], dtype=torch.float64, device="cuda")
    if world > 1:
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        kernel_ms_max = float(tmax[3])
    else:
        kernel_ms_max = kernel_ms
    total_rays = float(t[1])
    total_passes = float(t[2])

    if rank == 0:
        rays_per_launch = st.rays / args.steps  # this rank's launches
        flop_per_launch = rays_per_launch * N_SPHERES * FLOP_PER_TEST
        achieved = flop_per_launch / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "r01_hbm_traffic.json")
        if os.path.exists(tpath) and args.workload == "c3":
            try:
                traffic = json.load(open(tpath)).get("hbm_bytes_per_launch", {}).get(scan)
            except Exception:
                traffic = None
        line = {
            "metric": "Mrays/s (incl. bounces) at 1080p 64spp" if args.workload == "c3" else "Mrays/s (incl. bounces)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if frame_spp is None else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": {"c3": "random-spheres scene, 500 spheres, 1920x1080, 64 spp per GPU, max depth 8 (BASELINE configs[2])",
                             "c4": "random-spheres scene, 500 spheres, 3840x2160, 1024 spp per frame, max depth 8 (BASELINE configs[3])",
                             "c5": "random-spheres scene, 10000 spheres, 1920x1080, 256 spp per frame, max depth 8 (BASELINE configs[4])"}[args.workload],
                "spheres": N_SPHERES, "width": W, "height": H, "frame_spp": total_spp,
                "max_depth": DEPTH, "partition": args.partition if world > 1 else "none (one GPU)", "scan": scan,
                "collective": ("none" if world == 1 else
                               "reduce(sum,%s,%d) to rank 0" % ("uint8" if args.partition == "tiles" else "int32", W * H * 3)),
            },
            "rays_per_step": total_rays / args.steps,
            "lane_utilisation": total_rays / (64.0 * total_passes) if total_passes else None,
            "kernel_ms": kernel_ms,
            "kernel_ms_max_over_ranks": kernel_ms_max,
            "roofline": {
                "bound": "mfma", "pipe": "FP32 matrix/vector ALU (one shared pipe on gfx950)",
                "achieved": achieved, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_TFLOPS, "traffic": traffic,
                "flop_per_unit": "25 flop per ray-sphere test x %d spheres per ray" % N_SPHERES,
                "kernel": "trace_kernel, %s scan" % scan,
            },
        }
        # what the kernel executes, from the committed counter passes of the same command (profiles/)
        ppath = os.path.join(ROOT, "profiles", "r01_last", "pmc_trace_kernel.json")
        if os.path.exists(ppath) and args.workload == "c3":
            try:
                for name, rec in json.load(open(ppath)).items():
                    if name.startswith(scan + " scan") and "derived" in rec:
                        d = rec["derived"]
                        line["roofline"]["executed"] = {
                            "source": "profiles/r01_last/pmc_trace_kernel.json (rocprofv3 --pmc, separate passes)",
                            "valu_instructions_per_launch": rec.get("SQ_INSTS_VALU"),
                            "valu_instructions_per_cycle_per_simd": d.get("valu_instructions_per_cycle_per_simd"),
                            "resident_waves_per_simd": d.get("resident_waves_per_simd"),
                            "note": "wave64 VALU instructions issue every 2.3 or 4.7 cycles on this chip (measured), so "
                                    "about 0.3 per cycle per SIMD is back-to-back issue for this mix",
                        }
            except Exception:
                pass
        if scan == "hierarchy":
            line["roofline"]["note"] = ("algorithmic work = the reference's brute-force tests; the hierarchy skips most of "
                                        "them, so frac can exceed 1 -- see list_scan for the kernel that performs them all")
            if list_ms:
                a_list = flop_per_launch / (list_ms * 1e-3) / 1e12
                line["roofline"]["list_scan"] = {"kernel": "trace_kernel, matrix scan", "kernel_ms": list_ms,
                                                 "achieved": a_list, "frac": a_list / PEAK_FP32_TFLOPS}
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            line["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
