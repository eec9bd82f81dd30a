#!/bin/bash
# Builds the library with other scheduling thresholds of the path kernel (tools/librt_thr_<name>.so) for an A/B
# in one GPU job with tools/quick_time.py (RTAMD_LIBRARY).  Usage: tools/threshold_variants.sh
cd "$(dirname "$0")/../ray-tracer_amd/csrc" || exit 1
build() { # name defines...
    local name=$1; shift
    make -B OUT="$(pwd)/../../tools/librt_thr_$name.so" FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-fast-math -fno-slp-vectorize -I$(pwd)/../../include -I. -Wall -Wno-unused-function $*" > /dev/null 2>&1 || echo "build $name failed"
}
build r16 -DRT_REFILL_MIN_WALK=16u &
build r20 -DRT_REFILL_MIN_WALK=20u &
build r28 -DRT_REFILL_MIN_WALK=28u &
build r32 -DRT_REFILL_MIN_WALK=32u &
wait
build s24 -DRT_SHADE_MIN_WALK=24u &
build s28 -DRT_SHADE_MIN_WALK=28u &
build s40 -DRT_SHADE_MIN_WALK=40u &
build r20s28 -DRT_REFILL_MIN_WALK=20u -DRT_SHADE_MIN_WALK=28u &
wait
ls -la ../../tools/librt_thr_*.so
