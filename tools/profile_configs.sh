#!/bin/bash
# rocprofv3 --kernel-trace --stats summaries of the BASELINE configurations other than the bench frame:
# bench.py --workload c4 (4K / 1024 spp), --workload c5 (10 000 spheres, 1080p / 256 spp) and
# tools/c1_time.py (the reference's shipped frame).  Usage: tools/profile_configs.sh <tag>
# (outputs under gpurun_out/prof_configs_<tag>/; copy the *_kernel_stats.csv and *.json into profiles/)
set -u
TAG=${1:-r03}
OUT=gpurun_out/prof_configs_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
run() { # name command...
    local name=$1; shift
    timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o "$name" -- "$@" > "$OUT/$name.log" 2>&1 \
        || { echo "$name failed"; tail -5 "$OUT/$name.log"; exit 1; }
    cp "$OUT/$name/${name}_kernel_stats.csv" "$OUT/${name}_kernel_stats.csv"
    grep -h '"metric"\|C1:' "$OUT/$name.log" > "$OUT/${name}_result.txt" || true
}
run c4 python3 bench.py --workload c4 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs &&
run c5 python3 bench.py --workload c5 --steps 5 --warmup 1 --no-cpu-baseline --no-extra-legs &&
run c1 python3 tools/c1_time.py
ls "$OUT"
