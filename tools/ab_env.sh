#!/bin/bash
# A/B of environment switches on the whole bench step, same box, interleaved:  tools/ab_env.sh VAR v0 v1 ...   (repo root, GPU box)
# prints per value: frames/s, and the in-step HIP-event times of the three g_s conv launches and the up stage
set -u
VAR=$1; shift
for rep in 1 2; do
for v in "$@"; do
  env $VAR=$v python bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --no-psnr --inflight 0 --no-configs 2> /tmp/ab_env.log | python3 -c "
import sys, json, re
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']
        print('$VAR=$v value %.1f hbm %.1f v1 %.1f | dominant in timed region: avg_ms %.4f frac %.3f' % (d['value'], d['value_hbm_resident'], d.get('value_gpu_rans') or 0, r.get('avg_ms'), r.get('frac')))
"
  grep -E "sparse_conv +\((3262640|845992|211088|1000000, 4|407830, 32, 32, 27|407830, 32, 32, 8)|convT_gen +\(407830" /tmp/ab_env.log | awk '{printf "    %s %s %s avg %s\n", $1, $2, $3, $NF-1 ? $(NF-1) : $(NF-1)}'
done
done
