"""Condenses the output of tools/profile.sh into the files kept under profiles/.

Usage: python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/<dir>
Writes <dir>/kernel_stats.csv (copy of the --stats summary), <dir>/pmc_trace_kernel.json (per-launch
averages of every counter for each trace kernel, with derived occupancy and issue figures) and
prints the HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

NAMES = {", 3, ": "hierarchy scan", ", 2, ": "matrix scan", ", 1, ": "vector scan", ", 0, ": "exact scan"}
NOTE = ("kernel_cycles = GRBM_GUI_ACTIVE / 8 XCDs; resident waves = SQ_WAVE_CYCLES x 4 / kernel_cycles over 1024 SIMDs; "
        "the utilisation estimate prices a wave64 VALU instruction at 3.5 cycles (half of the mix at 2.3, half at 4.7: "
        "profiles/r01_microbench/inst_rate.txt); for the matrix scan the MFMA instructions (32 cycles each on the same "
        "pipe) are not in SQ_INSTS_VALU's price")


def main(src, dst):
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "kernel_stats.csv"))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(src, "*", "*_counter_collection.csv")):
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if "trace_kernel" in row["Kernel_Name"]:
                    acc[row["Kernel_Name"]][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
    out = {}
    for kernel, counters in acc.items():
        short = kernel[kernel.find("trace_kernel"):].split("(")[0]
        label = next((v for k, v in NAMES.items() if k in short), "scan") + ": " + short
        rec = {}
        for name, vals in sorted(counters.items()):
            per_dispatch = collections.defaultdict(float)  # a counter may come as one row per dimension
            for d, v in vals:
                per_dispatch[d] += v
            rec[name] = sum(per_dispatch.values()) / len(per_dispatch)
        if "GRBM_GUI_ACTIVE" in rec and "SQ_WAVE_CYCLES" in rec:
            cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
            waves = rec["SQ_WAVE_CYCLES"] * 4.0 / cyc / 1024.0
            valu = rec["SQ_INSTS_VALU"] / cyc / 1024.0
            rec["derived"] = {
                "kernel_cycles": cyc,
                "resident_waves_per_simd": waves,
                "occupancy_of_8_waves_per_simd": waves / 8.0,
                "valu_instructions_per_cycle_per_simd": valu,
                "salu_instructions_per_cycle_per_simd": rec["SQ_INSTS_SALU"] / cyc / 1024.0,
                "valu_issue_utilisation_estimate": valu * 3.5,
                "note": NOTE,
            }
        out[label] = rec
        if "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
            print(f"{label}: HBM bytes per launch {1024.0 * (rec['FETCH_SIZE'] + rec['WRITE_SIZE']):.0f} "
                  f"(FETCH_SIZE {rec['FETCH_SIZE']:.1f} KiB, WRITE_SIZE {rec['WRITE_SIZE']:.1f} KiB)")
    out["_note"] = ("per-launch averages of `python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline` (tools/profile.sh): "
                    "the hierarchy scan is the timed path, the matrix scan is bench.py's list-scan leg; SQ_* summed over the "
                    "chip, SQ_ACTIVE_*/SQ_WAVE_CYCLES in units of 4 cycles, GRBM_GUI_ACTIVE summed over the 8 XCDs; "
                    "FETCH_SIZE/WRITE_SIZE in KiB as rocprofv3 reports them (separate passes)")
    with open(os.path.join(dst, "pmc_trace_kernel.json"), "w") as fh:
        json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
