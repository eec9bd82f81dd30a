"""Condenses the output of tools/profile.sh into the files kept under profiles/.

Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<dir>
Writes <dir>/kernel_stats.csv (copy of the --stats summary), <dir>/pmc_trace_kernel.json (per-launch
averages of every counter for each trace kernel, with derived occupancy, issue and lane figures),
<dir>/hbm_traffic.json (bytes per launch from the FETCH_SIZE / WRITE_SIZE passes) and, if present,
<dir>/bench_under_profiler.json (the bench line printed under the --stats pass).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "ray-tracer_amd", "python"))

NAMES = {", 3, ": "hierarchy scan", ", 2, ": "matrix scan", ", 1, ": "vector scan", ", 0, ": "exact scan"}
N_SIMD = 1024.0  # 256 CUs x 4
NOTE = ("per launch. kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; SQ_WAVE_CYCLES counts quad-cycles summed over the chip: "
        "resident_waves_per_simd = SQ_WAVE_CYCLES x 4 / kernel_cycles / 1024 SIMDs. On this chip SQ_ACTIVE_INST_VALU comes "
        "out equal to the number of VALU instructions (it is no measure of issue time), and SQ_THREAD_CYCLES_VALU to the "
        "instructions times their enabled lanes, so valu_active_lane_share = SQ_THREAD_CYCLES_VALU / (64 x "
        "SQ_ACTIVE_INST_VALU) is the share of lanes EXEC enables per issued VALU instruction; valu_lane_ops = SQ_INSTS_VALU "
        "x 64 x that share; valu_lane_ops_frac_of_issue_peak = valu_lane_ops / (kernel_cycles x 1024 SIMDs x 32 lanes per "
        "clock). A wave64 VALU instruction holds its SIMD's issue for at least 2 cycles (4 for conversions, compares, "
        "selects and wide multiplies, 8 for transcendental ones: tools/inst_rate.hip), an f32 MFMA for the "
        "SQ_VALU_MFMA_BUSY_CYCLES it reports: pipe_busy_min = ((SQ_INSTS_VALU - SQ_INSTS_MFMA) x 2 + "
        "SQ_VALU_MFMA_BUSY_CYCLES) / (kernel_cycles x 1024) is a lower bound of the share of cycles the shared "
        "vector/matrix pipe is issuing; mfma_pipe_share its MFMA part")


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    bench_line = os.path.join(src, "bench_under_profiler.json")
    if os.path.exists(bench_line) and os.path.getsize(bench_line):
        shutil.copy(bench_line, os.path.join(dst, "bench_under_profiler.json"))
    # counter -> list of per-pass averages (a counter may be collected in several passes)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sorted(glob.glob(os.path.join(src, "*", "*_counter_collection.csv"))):
        per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if "trace_kernel" in row["Kernel_Name"]:
                    # a counter may come as one row per dimension: sum them per dispatch
                    per[row["Kernel_Name"]][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
        for kernel, counters in per.items():
            for name, by_dispatch in counters.items():
                acc[kernel][name].append(sum(by_dispatch.values()) / len(by_dispatch))
    out = {}
    traffic = {"hbm_bytes_per_launch": {}, "fetch_kib_reported": {}, "write_kib_reported": {}}
    for kernel, counters in acc.items():
        short = kernel[kernel.find("trace_kernel"):].split("(")[0]
        scan = next((v for k, v in NAMES.items() if k in short), "scan")
        label = scan + ": " + short
        rec = {name: sum(vals) / len(vals) for name, vals in sorted(counters.items())}
        if "GRBM_GUI_ACTIVE" in rec and "SQ_WAVE_CYCLES" in rec:
            cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
            d = {
                "kernel_cycles": cyc,
                "resident_waves_per_simd": rec["SQ_WAVE_CYCLES"] * 4.0 / cyc / N_SIMD,
                "occupancy_of_8_waves_per_simd": rec["SQ_WAVE_CYCLES"] * 4.0 / cyc / N_SIMD / 8.0,
                "valu_instructions_per_cycle_per_simd": rec["SQ_INSTS_VALU"] / cyc / N_SIMD,
                "salu_instructions_per_cycle_per_simd": rec["SQ_INSTS_SALU"] / cyc / N_SIMD,
            }
            mfma_n = rec.get("SQ_INSTS_MFMA", 0.0)
            mfma_cyc = rec.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            d["pipe_busy_min"] = ((rec["SQ_INSTS_VALU"] - mfma_n) * 2.0 + mfma_cyc) / cyc / N_SIMD
            if "SQ_THREAD_CYCLES_VALU" in rec and rec.get("SQ_ACTIVE_INST_VALU"):
                share = rec["SQ_THREAD_CYCLES_VALU"] / (64.0 * rec["SQ_ACTIVE_INST_VALU"])
                d["valu_active_lane_share"] = share
                d["valu_lane_ops"] = rec["SQ_INSTS_VALU"] * 64.0 * share
                d["valu_lane_ops_frac_of_issue_peak"] = d["valu_lane_ops"] / (cyc * N_SIMD * 32.0)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in rec:
                d["mfma_pipe_share"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / cyc / N_SIMD
            d["note"] = NOTE
            rec["derived"] = d
        out[label] = rec
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec and short.endswith(", 0>"):
            key = scan.split()[0]
            traffic["hbm_bytes_per_launch"][key] = 1024.0 * (rec["FETCH_SIZE"] + rec["WRITE_SIZE"])
            traffic["fetch_kib_reported"][key] = rec["FETCH_SIZE"]
            traffic["write_kib_reported"][key] = rec["WRITE_SIZE"]
            print(f"{label}: HBM bytes per launch {1024.0 * (rec['FETCH_SIZE'] + rec['WRITE_SIZE']):.0f} "
                  f"(FETCH_SIZE {rec['FETCH_SIZE']:.1f} KiB, WRITE_SIZE {rec['WRITE_SIZE']:.1f} KiB)")
    out["_note"] = ("per-launch averages of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline` (tools/profile.sh): "
                    "kernels named <.., 3, 0> are the timed hierarchy scan, <.., 2, 0> bench.py's list-scan leg, <.., N, 2> "
                    "the counting builds of its untimed launches; SQ_* summed over the chip, GRBM_GUI_ACTIVE summed over "
                    "the 8 XCDs; FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them (separate passes)")
    # what bench.py compares with the sources it runs from before it trusts these counters: the
    # fingerprint tools/profile.sh took on the GPU box next to the counters, else this tree's
    build = None
    try:
        build = json.load(open(os.path.join(src, "build.json")))
    except Exception:
        try:
            import rtamd

            build = {"source_fingerprint": rtamd.source_fingerprint(), "taken": "when the profile was summarised"}
        except Exception as e:
            build = {"error": str(e)[:200]}
    out["_build"] = build
    traffic["_build"] = build
    with open(os.path.join(dst, "pmc_trace_kernel.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    traffic["note"] = ("rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes, per trace_kernel launch of the bench frame. "
                       "The kernel's only frame traffic is the flush of the per-wave chunk sums into the uint32 sums "
                       "(1920*1080*3*4 = 24.9 MB).  Chunks that own their pixels flush with a plain read-modify-write (24.9 MB "
                       "each way); when the work queue also cuts a strip's sample range, the segments share pixels and flush "
                       "with atomicAdd (no return value), which is carried out behind the L2s: almost nothing is fetched and "
                       "every addition counts as a write.  4-byte accesses read 1:1 in the counters, so the 2x correction the "
                       "gfx950 guide prescribes for 16-byte streaming reads is not applied.")
    traffic["algorithmic_bytes_per_launch"] = 1920 * 1080 * 3 * 4 * 2 + 36 * 500 + 16 * 3 * 512
    with open(os.path.join(dst, "hbm_traffic.json"), "w") as fh:
        json.dump(traffic, fh, indent=1)
    for label, rec in out.items():
        if isinstance(rec, dict) and "derived" in rec:
            print(label, {k: (round(v, 4) if isinstance(v, float) else v) for k, v in rec["derived"].items() if k != "note"})


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
