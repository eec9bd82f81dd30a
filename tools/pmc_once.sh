#!/bin/bash
# One rocprofv3 --pmc pass over a short bench.py run (timed leg only).  Usage: tools/pmc_once.sh <name> <counters...>
set -u
NAME=$1; shift
OUT=gpurun_out/pmc_$NAME
mkdir -p "$OUT"; export TMPDIR=/tmp
timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT" -o "$NAME" -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs > "$OUT/log.txt" 2>&1 || { echo "pass $NAME failed"; tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT/${NAME}_counter_collection.csv" <<'PY'
import csv, sys, collections
acc=collections.defaultdict(float); disp=set()
for r in csv.DictReader(open(sys.argv[1])):
    if "trace_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]]+=float(r["Counter_Value"]); disp.add(r["Dispatch_Id"])
for c,v in sorted(acc.items()): print(f"{c:32s} {v/len(disp):.4g}")
PY
