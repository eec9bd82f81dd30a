#!/bin/bash
# Collects the rocprofv3 evidence for bench.py on a GPU box: one kernel-trace
# pass with --stats, then counter passes (each --pmc set in a run of its own).
# Usage: tools/profile.sh <tag>     (outputs under gpurun_out/prof_<tag>/)
set -u
TAG=${1:-r02}
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline"
# which sources these counters belong to (bench.py refuses a profile of another build)
python3 -c "import sys, json; sys.path.insert(0, 'ray-tracer_amd/python'); import rtamd; print(json.dumps({'source_fingerprint': rtamd.source_fingerprint(), 'taken': 'on the GPU box, by tools/profile.sh'}))" > "$OUT/build.json" || exit 1

# (the timing pass with bench.py's own default steps, so that the first launch under the profiler, which
# takes a millisecond longer, weighs little in the --stats average)
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline > "$OUT/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$OUT/trace.log"; exit 1; }
grep '"metric"' "$OUT/trace.log" > "$OUT/bench_under_profiler.json" || true

pass() { # name counters...
    local name=$1; shift
    timeout -s KILL 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o "$name" -- $BENCH > "$OUT/$name.log" 2>&1 || { echo "pmc pass $name failed"; tail -5 "$OUT/$name.log"; exit 1; }
}
# at most 8 SQ_, 2 GRBM_ and 4 TCC_ slots per pass (FETCH_SIZE takes 3 TCC_, WRITE_SIZE 2)
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE &&
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR &&
pass sq3 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE &&
pass sq4 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_IOPS &&
pass fetch FETCH_SIZE &&
pass write WRITE_SIZE
find "$OUT" -name "*.csv" | head -30
