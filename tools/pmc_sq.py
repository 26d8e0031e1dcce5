#!/usr/bin/env python3
"""Mean SQ counters of the largest launch of kernels whose name contains a pattern.

    gpurun: rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY ... --output-format csv \
                -d gpurun_out/pmc_sq -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --inflight 0
    here:   python tools/pmc_sq.py gpurun_out/pmc_sq k_gconv_mfma
"""
import collections
import csv
import glob
import json
import sys


def main(src, pat):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"{src}/**/*_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if pat in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    if not agg:
        print("no kernel matches", pat)
        return
    key = max(agg, key=lambda k: k[1])
    out = {"kernel": key[0], "grid_threads": key[1]}
    out.update({c: sum(v) / len(v) for c, v in sorted(agg[key].items())})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
