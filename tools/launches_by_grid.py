#!/usr/bin/env python3
"""Kernel trace of a rocprofv3 run grouped by kernel and grid size: tools/launches_by_grid.py <trace dir> <out.json> ["note"]"""
import collections
import csv
import glob
import json
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
g = collections.defaultdict(list)
for fn in glob.glob(f"{src}/**/*_kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        grid = int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)
        g[(r["Kernel_Name"].split("(")[0], grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
rows = [{"kernel": k, "grid_threads": gr, "launches": len(v), "avg_us": round(sum(v) / len(v), 1), "min_us": round(min(v), 1),
         "max_us": round(max(v), 1)} for (k, gr), v in g.items()]
rows.sort(key=lambda r: -r["avg_us"] * r["launches"])
json.dump({"source": note, "kernels": rows[:60]}, open(dst, "w"), indent=1)
for r in rows[:14]:
    print(r)
