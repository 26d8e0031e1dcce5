#!/bin/bash
# A/B of prebuilt library variants on the GPU box: tools/ab_conv.sh <variant.so>...   (run from the repo root)
# Each variant is copied over lib/libpcc_hip.so and tools/bench_conv.py times the dominant layer with it.
set -u
LIB=demo-learned-point-cloud-compression_amd/lib/libpcc_hip.so
cp "$LIB" /tmp/libpcc_orig.so
for v in "$@"; do
  echo "=== $v"
  cp "$v" "$LIB"
  python tools/bench_conv.py --cases ${CASES:-cand_pruned} --reps ${REPS:-20} 2>&1 | grep -v "^\[" 
done
cp /tmp/libpcc_orig.so "$LIB"
