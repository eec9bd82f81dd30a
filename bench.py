#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (incl. bounces) of the per-pixel ray-tracing path.

Workload (BASELINE.json configs[2]): the 500-sphere random-spheres scene at
1920x1080, 64 samples per pixel, max depth 8.  One "step" = one frame.

  N = 1   one MI355X traces all 64 samples and resolves the RGB8 frame.
  N > 1   the SAME frame on N GPUs (strong scaling, the configuration the metric
          names), split by interleaved 8x8 tiles; every rank divides and packs its
          own tiles and the tile blocks are gathered to rank 0 over RCCL.  The
          exchange of frame k overlaps the tracing of frame k + 1.
          --scaling weak: every rank traces 64 samples of every pixel instead (the
          frame has 64*N spp) and the integer sums are sum-reduced to rank 0.
          --partition samples|tiles overrides the split.

Inputs (scene, camera) and outputs (sums, frame) stay in HBM during the timed
region.  Prints ONE JSON line on rank 0.

What the line's `roofline` means.  The timed kernel walks a bounding-box hierarchy: it
executes a few per cent of the reference's brute-force ray-sphere tests, so pricing it
at "25 flop x spheres x rays" would exceed the machine's peak and say nothing.  Its
`roofline` is therefore the vector ALU it actually runs on: lane-operations issued per
launch (hardware counters of the same command, committed under profiles/) over the
kernel's duration measured here, against the chip's VALU issue peak; the executed box
and sphere tests (counted by the kernel itself in an untimed launch) are priced in flop
beside it, and the brute-force-equivalent figure is kept under its own key.  The kernel
that does perform every one of the reference's tests (the matrix-core list scan) is
timed under the same steps/warmup protocol as a second leg, `roofline.list_scan`, with
the fraction of the FP32 peak the survey's definition gives.
"""
import argparse
import json
import os
import re
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

# BASELINE.json configurations: name -> (width, height, frame spp, spheres, grid, partition for N > 1)
WORKLOADS = {
    "c3": (1920, 1080, 64, 500, 11, "tiles"),       # configs[2], the one the metric is quoted on
    "c4": (3840, 2160, 1024, 500, 11, "samples"),   # configs[3]: 1024 spp split over the GPUs
    "c5": (1920, 1080, 256, 10000, 50, "tiles"),    # configs[4]: 10k spheres, tile-partitioned
}
WORKLOAD_TEXT = {
    "c3": "random-spheres scene, 500 spheres, 1920x1080, 64 spp, max depth 8 (BASELINE configs[2])",
    "c4": "random-spheres scene, 500 spheres, 3840x2160, 1024 spp, max depth 8 (BASELINE configs[3])",
    "c5": "random-spheres scene, 10000 spheres, 1920x1080, 256 spp, max depth 8 (BASELINE configs[4])",
}
SCANS = {"auto": rtamd.FILTER_AUTO, "exact": rtamd.FILTER_EXACT, "vector": rtamd.FILTER_CONSERVATIVE,
         "matrix": rtamd.FILTER_MATRIX, "hierarchy": rtamd.FILTER_BVH}
FLOP_PER_TEST = 25.0          # miss path of RaySphereIntersectionTest, Source/main.cpp:475-479
# One visit of an inner node tests both children's boxes: 12 plane distances of two fused
# multiply-adds each (48 flop) and 12 min/max (rt_bvh_walk.h).
FLOP_PER_NODE_VISIT = 60.0
FLOP_PER_ROOT = 6.0           # negate, subtract, sqrt, divide, two range compares (Source/main.cpp:490-499)
PEAK_FP32_TFLOPS = 157.3      # 256 CU x 128 FMA lanes x 2 flop x 2.4 GHz (matrix f32 = vector f32 on gfx950)
PEAK_VALU_TLANEOPS = 78.6     # 256 CU x 4 SIMD x 32 lanes per clock x 2.4 GHz: one wave64 instruction per 2 cycles
# rocprofv3 evidence of this build's `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline`; the files
# carry the fingerprint of the sources they were taken from, and are used only if it is this build's
PROFILE_DIR = os.path.join("profiles", "r03_final")


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
    # All host threads, column strips like TraceScene (Source/main.cpp:7, 647-659), one generator per strip
    # (the reference's shared generator is a data race).  One thread per CPU the host reports; the CPUs this
    # job may actually use can be fewer (a GPU box gives one GPU's job a share of its host), so both are
    # stated, and the sample is sized from the single-thread rate above for the share, not for the count.
    host_cores = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else host_cores
    try:
        share = max(1, int(os.environ.get("OMP_NUM_THREADS", "") or usable))
    except ValueError:
        share = usable
    threads = max(1, min(host_cores, W))
    per_column_s = port_s / xs
    columns = int(min(W, max(threads, 12.0 * min(share, usable, threads) / per_column_s)))
    columns = max(threads, columns // threads * threads)  # whole strips
    import ctypes as C
    rgb = np.zeros((H, W, 3), dtype=np.uint8)
    r = C.c_uint64(0)
    t0 = time.time()
    O.lib().orc_trace_scene_threads(scene.ctypes.data, len(scene), C.byref(ocam), W, H, SPP, DEPTH, (W - columns) // 2,
                                    columns, threads, rgb.ctypes.data, C.byref(r))
    mt_s = time.time() - t0
    out["port_threads"] = {"value": r.value / mt_s / 1e6, "unit": "Mrays/s", "threads": threads, "host_cores": host_cores,
                           "cpus_in_affinity_mask": usable, "cpu_share_of_this_job": share, "seconds": mt_s,
                           "sample": f"{columns} centre columns of the frame, all rows: one strip of {columns // threads} "
                                     f"column(s) and one generator per thread, {threads} threads"}
    return out


def profile_matches_build():
    """True if the committed counter profile was taken from the sources this run's library was built from
    (the profile carries their fingerprint), False if from others, None if there is no profile."""
    try:
        prof = json.load(open(os.path.join(ROOT, PROFILE_DIR, "pmc_trace_kernel.json")))
    except Exception:
        return None
    return prof.get("_build", {}).get("source_fingerprint") == rtamd.source_fingerprint()


def load_profile(scan):
    """Per-launch counters of `scan`'s trace kernel from the committed rocprofv3 passes -- of THIS build --
    or None."""
    if not profile_matches_build():
        return None
    path = os.path.join(ROOT, PROFILE_DIR, "pmc_trace_kernel.json")
    try:
        for name, rec in json.load(open(path)).items():
            if name.startswith(scan + " scan") and name.endswith(", 0>"):  # the plain build, not the counting one
                return rec
    except Exception:
        pass
    return None


def load_traffic(scan):
    if not profile_matches_build():
        return None
    try:
        return json.load(open(os.path.join(ROOT, PROFILE_DIR, "hbm_traffic.json")))["hbm_bytes_per_launch"].get(scan)
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="c3")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the workload's frame split over the GPUs; weak = its spp per GPU")
    ap.add_argument("--partition", choices=["samples", "tiles"], default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the list-scan leg and the counting launches (N = 1 only has them)")
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="N > 1: nccl = RCCL, one GPU per rank (the measured configuration); gloo = rehearsal of the N > 1 "
                         "path on fewer GPUs than ranks (ranks share devices, exchanged buffers go through the host)")
    ap.add_argument("--exchange-alone", action="store_true",
                    help="N = 1: run the N > 1 path's pack / collective / unpack steps over a one-rank RCCL communicator "
                         "(exercises that path on a one-GPU machine; not the N = 1 measurement)")
    ap.add_argument("--frames-per-launch", type=int, default=0,
                    help="frames (steps) whose shares one trace launch carries (rt_accumulate_batch_device); 0 = 1 on one "
                         "GPU, 4 on several, where a rank's share of a frame is small against a launch's drain")
    ap.add_argument("--batched-leg", action="store_true",
                    help="N = 1: also time the same frames four to a launch (informational `batched_frames` entry; off by "
                         "default so that every launch of the timed kernel in a profile of this command is one frame)")
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
    rehearsal = world > 1 and args.backend == "gloo"
    device_index = local_rank % torch.cuda.device_count() if rehearsal else local_rank
    torch.cuda.set_device(device_index)
    alone = world == 1 and args.exchange_alone
    if alone:
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
    if world > 1 or alone:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    global W, H, SPP, N_SPHERES
    W, H, SPP, N_SPHERES, grid, default_partition = WORKLOADS[args.workload]
    if args.partition is None:
        args.partition = default_partition if args.scaling == "strong" else "samples"
    ctx = rtamd.Context(device_index)
    scene = scenes.random_spheres(N_SPHERES, grid_half=grid)
    ctx.set_scene(scene)
    ctx.set_option(rtamd.OPT_FILTER, SCANS[args.scan])
    scan = args.scan
    if scan == "auto":  # the library's rule (rt_mi355x.h, RT_OPT_FILTER)
        scan = "hierarchy" if N_SPHERES >= 13 else "vector"
    cam = rtamd.make_camera(*scenes.RANDOM_SPHERES_CAMERA)
    total_spp = SPP * world if args.scaling == "weak" else SPP
    p = dist_render.rank_params(W, H, total_spp, DEPTH, 1, args.partition, rank, world)
    tracer = dist_render.DeviceTracer(ctx, cam)

    class TimedTracer:
        """Brackets the trace kernel with events on the stream it is launched on."""

        def __init__(self):
            self.events = None

        def _timed(self, launch):
            if self.events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            launch()
            if self.events is not None:
                e1.record()
                self.events.append((e0, e1))

        def accumulate(self, params, sums):
            self._timed(lambda: tracer.accumulate(params, sums))

        def accumulate_batch(self, jobs):
            self._timed(lambda: tracer.accumulate_batch(jobs))

        resolve = staticmethod(tracer.resolve)
        pack_tiles = staticmethod(tracer.pack_tiles)
        unpack_tiles = staticmethod(tracer.unpack_tiles)

    timed = TimedTracer()
    per_launch = args.frames_per_launch or (4 if world > 1 else 1)
    per_launch = max(1, min(per_launch, rtamd.MAX_JOBS))

    def make_pipe(frames_per_launch):
        return dist_render.FramePipeline(timed, p, (W, H, total_spp), args.partition, rank, world, dist, torch, "cuda",
                                         host_exchange=rehearsal, always_exchange=alone, frames_per_launch=frames_per_launch)

    pipe = make_pipe(per_launch)

    def fence():
        torch.cuda.synchronize()
        if world > 1 or alone:
            dist.barrier()
        torch.cuda.synchronize()

    def run_leg(steps, warmup):
        """`warmup` untimed frames, then `steps` timed ones -> (seconds, mean kernel ms, stats)."""
        timed.events = None
        for _ in range(warmup):
            pipe.submit()
        pipe.finish()
        fence()
        ctx.reset_stats()
        fence()
        timed.events = []
        t0 = time.perf_counter()
        for _ in range(steps):
            pipe.submit()
        pipe.finish()
        fence()
        seconds = time.perf_counter() - t0
        # kernel time per STEP: a launch may carry several steps' frames
        kernel_ms = float(np.sum([a.elapsed_time(b) for a, b in timed.events])) / steps if timed.events else 0.0
        timed.events = None
        return seconds, kernel_ms, ctx.stats()

    elapsed, kernel_ms, st = run_leg(args.steps, args.warmup)

    # ---- second leg and counting launches (one GPU only: they are about the kernels, not the exchange)
    extra = world == 1 and not alone and not args.no_extra_legs and N_SPHERES <= 2000
    list_leg = counted = batched_leg = None
    if extra and per_launch == 1 and args.batched_leg:
        # the same frames, four to a launch: what the timed figure would be without the drain at every launch's end
        pipe = make_pipe(4)
        b_warmup = max(args.warmup, 8)  # every one of the pipeline's eight slots once before the clock starts
        b_elapsed, b_kernel_ms, b_st = run_leg(args.steps, b_warmup)
        batched_leg = {"frames_per_launch": 4, "steps": args.steps, "warmup": b_warmup,
                       "ms_per_step": b_elapsed / args.steps * 1e3, "kernel_ms_per_step": b_kernel_ms,
                       "value": b_st.rays / b_elapsed / 1e6, "unit": "Mrays/s",
                       "note": "informational: the same frames traced four to a launch (rt_accumulate_batch_device), "
                               "every frame zeroed, traced and resolved as in the timed leg"}
        pipe = make_pipe(1)
    if extra:
        def count_work(filt):
            ctx.set_option(rtamd.OPT_FILTER, filt)
            ctx.set_option(rtamd.OPT_COUNT_WORK, 1)
            ctx.reset_stats()
            pipe.submit()
            pipe.finish()
            s = ctx.stats()
            ctx.set_option(rtamd.OPT_COUNT_WORK, 0)
            return {"rays": s.rays, "node_visits": s.node_visits, "exact_tests": s.exact_tests,
                    "root_evaluations": s.root_evaluations}

        counted = {scan: count_work(SCANS[args.scan])}
        if scan == "hierarchy":
            ctx.set_option(rtamd.OPT_FILTER, rtamd.FILTER_MATRIX)
            l_elapsed, l_kernel_ms, l_st = run_leg(args.steps, args.warmup)
            list_leg = (l_elapsed, l_kernel_ms, l_st)
            counted["matrix"] = count_work(rtamd.FILTER_MATRIX)
        ctx.set_option(rtamd.OPT_FILTER, SCANS[args.scan])

    if world > 1 or alone:
        dist.barrier()
    t = torch.tensor([elapsed, float(st.rays), float(st.wave_passes), kernel_ms], dtype=torch.float64,
                     device="cpu" if rehearsal else "cuda")
    if world > 1 or alone:
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
        kernel_s = kernel_ms * 1e-3
        ref_equiv = rays_per_launch * N_SPHERES * FLOP_PER_TEST / kernel_s / 1e12
        collective = "none"
        if world > 1 or alone:
            collective = ("gather(uint8, %d bytes per rank) to rank 0" % pipe.block_bytes if args.partition == "tiles"
                          else "reduce(sum, int32, %d) to rank 0" % (W * H * 3))
        line = {
            "metric": "Mrays/s (incl. bounces) at 1080p 64spp" if args.workload == "c3" else "Mrays/s (incl. bounces)",
            "value": total_rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "none",
            "vs_baseline": None,
            "dtype": "f32",
            "data": ("synthetic (REHEARSAL over gloo: ranks share GPUs, not a measurement)" if rehearsal else
                     "synthetic (one rank running the N > 1 exchange path, not the N = 1 measurement)" if alone else "synthetic"),
            "config": {
                "workload": WORKLOAD_TEXT[args.workload] + (" per GPU (weak scaling)" if world > 1 and args.scaling == "weak" else ""),
                "spheres": N_SPHERES, "width": W, "height": H, "frame_spp": total_spp,
                "max_depth": DEPTH, "partition": args.partition if world > 1 or alone else "none (one GPU)", "scan": scan,
                "collective": collective,
                "overlap": "exchange of frame k runs beside the tracing of frame k + 1" if world > 1 else "none",
                "frames_per_launch": per_launch,
            },
            "rays_per_step": total_rays / args.steps,
            "lane_utilisation": total_rays / (64.0 * total_passes) if total_passes else None,
            "kernel_ms": kernel_ms,
            "kernel_ms_max_over_ranks": kernel_ms_max,
        }

        # ---- roofline of the timed kernel
        prof = load_profile(scan) if args.workload == "c3" and world == 1 and per_launch == 1 else None
        roof = {"kernel": "trace_kernel, %s scan" % scan, "kernel_ms": kernel_ms,
                # the counters below are used only when they were taken from this build's sources
                "profile": PROFILE_DIR, "profile_matches_build": profile_matches_build(),
                "build_fingerprint": rtamd.source_fingerprint()}
        if scan in ("matrix", "vector", "exact"):
            # a list scan performs every one of the reference's tests: the survey's definition applies
            roof.update({"bound": "mfma" if scan == "matrix" else "valu",
                         "pipe": "FP32 matrix/vector ALU (one shared pipe on gfx950)",
                         "achieved": ref_equiv, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": ref_equiv / PEAK_FP32_TFLOPS,
                         "flop_per_unit": "25 flop per ray-sphere test x %d spheres per ray" % N_SPHERES})
        else:
            roof.update({"bound": "valu", "pipe": "vector ALU issue (the kernel executes no MFMA)",
                         "achieved": None, "peak": PEAK_VALU_TLANEOPS, "unit": "Tlaneop/s", "frac": None,
                         "basis": ("no counter profile of this build (profile stale or absent): achieved and frac left null"
                                   if args.workload == "c3" and world == 1 and per_launch == 1 else
                                   "the committed counter profile is of the default one-GPU command (one 1080p / 64 spp frame "
                                   "per launch): achieved and frac left null for this one")})
            if prof and "SQ_INSTS_VALU" in prof and "derived" in prof and prof["derived"].get("valu_active_lane_share"):
                d = prof["derived"]
                lane_ops = prof["SQ_INSTS_VALU"] * 64.0 * d["valu_active_lane_share"]
                roof.update({
                    "achieved": lane_ops / kernel_s / 1e12,
                    "frac": lane_ops / kernel_s / 1e12 / PEAK_VALU_TLANEOPS,
                    "lane_ops_per_launch": lane_ops,
                    "valu_instructions_per_launch": prof["SQ_INSTS_VALU"],
                    "valu_active_lane_share": d["valu_active_lane_share"],
                    "valu_issue_busy_min": d.get("pipe_busy_min"),
                    "resident_waves_per_simd": d.get("resident_waves_per_simd"),
                    "basis": "lane-operations per launch = SQ_INSTS_VALU x 64 x SQ_THREAD_CYCLES_VALU / (64 x "
                             "SQ_ACTIVE_INST_VALU), rocprofv3 --pmc passes of this command in %s; duration = this "
                             "run's events; peak = one wave64 instruction per 2 cycles per SIMD; valu_issue_busy_min = "
                             "SQ_INSTS_VALU x 2 cycles / SIMD-cycles (a lower bound: about half of the mix issues in 4)" % PROFILE_DIR,
                })
            # the brute-force-equivalent rate: a speed-up figure, NOT a utilisation (the walk skips most tests)
            roof["reference_equivalent_tflops"] = ref_equiv
        roof["traffic"] = load_traffic(scan) if args.workload == "c3" and world == 1 and per_launch == 1 else None
        if counted and scan in counted:
            c = counted[scan]
            flop = (c["node_visits"] * FLOP_PER_NODE_VISIT + c["exact_tests"] * FLOP_PER_TEST +
                    c["root_evaluations"] * FLOP_PER_ROOT)
            roof["executed_tests"] = dict(c, flop_per_launch=flop, tflops=flop / kernel_s / 1e12,
                                          frac_of_fp32_peak=flop / kernel_s / 1e12 / PEAK_FP32_TFLOPS,
                                          per_ray={"node_visits": c["node_visits"] / max(c["rays"], 1),
                                                   "exact_tests": c["exact_tests"] / max(c["rays"], 1),
                                                   "root_evaluations": c["root_evaluations"] / max(c["rays"], 1)},
                                          pricing="60 flop per node visit (two boxes), 25 per discriminant, 6 per root; "
                                                  "counted by the kernel's counting build in an untimed launch")
        if list_leg:
            l_elapsed, l_kernel_ms, l_st = list_leg
            l_rays = l_st.rays / args.steps
            a_list = l_rays * N_SPHERES * FLOP_PER_TEST / (l_kernel_ms * 1e-3) / 1e12
            leg = {
                "kernel": "trace_kernel, matrix scan (performs every one of the reference's ray-sphere tests)",
                "steps": args.steps, "warmup": args.warmup,
                "value": l_st.rays / l_elapsed / 1e6, "unit_value": "Mrays/s",
                "ms_per_step": l_elapsed / args.steps * 1e3, "kernel_ms": l_kernel_ms,
                "bound": "mfma", "pipe": "FP32 matrix/vector ALU (one shared pipe on gfx950)",
                "achieved": a_list, "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": a_list / PEAK_FP32_TFLOPS,
                "flop_per_unit": "25 flop per ray-sphere test x %d spheres per ray" % N_SPHERES,
                "traffic": load_traffic("matrix") if args.workload == "c3" else None,
            }
            lp = load_profile("matrix") if args.workload == "c3" else None
            if lp and "derived" in lp:
                leg["executed"] = {k: lp["derived"].get(k) for k in
                                   ("mfma_pipe_share", "pipe_busy_min", "valu_active_lane_share", "resident_waves_per_simd")}
                leg["executed"]["mfma_instructions_per_launch"] = lp.get("SQ_INSTS_MFMA")
                leg["executed"]["valu_instructions_per_launch"] = lp.get("SQ_INSTS_VALU")
            if counted and "matrix" in counted:
                leg["executed_tests"] = counted["matrix"]
            roof["list_scan"] = leg
            # the same leg's headline figures as scalars of their own (a reader that keeps only scalar keys keeps them)
            roof["list_scan_frac"] = leg["frac"]
            roof["list_scan_ms"] = l_kernel_ms
            roof["list_scan_mrays"] = leg["value"]
            roof["list_scan_tflops"] = a_list
        if batched_leg:
            line["batched_frames"] = batched_leg
        line["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and args.workload == "c3":
            line["cpu_baseline"] = cpu_baseline(scene)
        print(json.dumps(line), flush=True)

    ctx.close()
    if world > 1 or alone:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
