#!/bin/bash
# A/B of prebuilt library variants on the whole bench step: tools/ab_bench.sh <variant.so>...   (repo root, GPU box)
set -u
LIB=demo-learned-point-cloud-compression_amd/lib/libpcc_hip.so
cp "$LIB" /tmp/libpcc_orig.so
for v in "$@"; do
  echo "=== $v"
  cp "$v" "$LIB"
  python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-psnr --inflight 0 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('value %.1f  ms/step %.3f  hbm-resident %.1f  gpu_rans %s  enc %s dec %s' % (d['value'], d['ms_per_step'], d['value_hbm_resident'], d.get('value_gpu_rans'), d.get('encode_ms'), d.get('decode_ms')))
        print('roofline %s avg_ms %.4f frac %.3f' % (r.get('kernel'), r.get('avg_ms'), r.get('frac')))
"
done
cp /tmp/libpcc_orig.so "$LIB"
