#!/bin/bash
# A/B of prebuilt library variants for the input layer: tools/ab_first.sh <variant.so>...   (run from the repo root)
set -u
LIB=demo-learned-point-cloud-compression_amd/lib/libpcc_hip.so
cp "$LIB" /tmp/libpcc_orig.so
for v in "$@"; do
  echo "=== $v"
  cp "$v" "$LIB"
  python tools/bench_first.py 2>&1 | grep "first layer"
done
cp /tmp/libpcc_orig.so "$LIB"
