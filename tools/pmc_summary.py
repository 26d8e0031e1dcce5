#!/usr/bin/env python3
"""Summarise rocprofv3 PMC passes (one counter per pass, MI355X guide §HBM) into profiles/.

    gpurun: for c in FETCH_SIZE WRITE_SIZE; do rocprofv3 --kernel-trace --pmc $c --output-format csv \
                -d gpurun_out/pmc_$c -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline; done
    here:   python tools/pmc_summary.py gpurun_out profiles/r01_pmc_summary.json

Units / corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE reports half of the bytes of a wide (16 B/lane) read, so hbm_bytes = (2*FETCH + WRITE)*1024.
"""
import collections
import csv
import glob
import json
import sys


def main(src, dst):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"{src}/pmc_{c}/*/*_counter_collection.csv"):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == c:
                    agg[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))][c].append(float(r["Counter_Value"]))
    out = []
    for (name, grid), d in agg.items():
        if "FETCH_SIZE" not in d or "WRITE_SIZE" not in d:
            continue
        fk = sum(d["FETCH_SIZE"]) / len(d["FETCH_SIZE"])
        wk = sum(d["WRITE_SIZE"]) / len(d["WRITE_SIZE"])
        out.append({"kernel": name, "grid_threads": grid, "launches_sampled": len(d["FETCH_SIZE"]),
                    "FETCH_SIZE_KiB": fk, "WRITE_SIZE_KiB": wk, "hbm_bytes_corrected": (2 * fk + wk) * 1024})
    out.sort(key=lambda r: -r["hbm_bytes_corrected"])
    json.dump({"note": "per-launch means; hbm_bytes_corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 correction)",
               "kernels": out[:40]}, open(dst, "w"), indent=1)
    for r in out[:12]:
        print(f"{r['kernel'][:44]:46s} grid {r['grid_threads']:>9d}  {r['hbm_bytes_corrected'] / 1e6:9.1f} MB")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
