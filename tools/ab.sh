#!/bin/bash
# A/B of prebuilt library variants on the GPU box: tools/ab.sh "<command>" <variant.so>...   (run from the repo root)
# Each variant is copied over lib/libpcc_hip.so and the command is run with it; the shipped library is restored.
set -u
LIB=demo-learned-point-cloud-compression_amd/lib/libpcc_hip.so
CMD="$1"; shift
cp "$LIB" /tmp/libpcc_orig.so
for v in "$@"; do
  echo "=== $v"
  cp "$v" "$LIB"
  bash -c "$CMD" 2>&1 | grep -v "^\[" | grep -v amdgpu.ids
done
cp /tmp/libpcc_orig.so "$LIB"
