#!/usr/bin/env python3
"""The figures of a bench.py log that are compared run to run: tools/line.py <log>"""
import json
import sys

for ln in open(sys.argv[1]):
    if ln.startswith("{"):
        d = json.loads(ln)
        keys = ("value", "ms_per_step", "value_hbm_resident", "value_gpu_rans")
        print({k: round(d[k], 2) for k in keys if d.get(k) is not None})
        g = d.get("gpu_rans") or {}
        if g.get("kernels"):
            print({k: round(v["avg_ms"], 4) for k, v in g["kernels"].items()}, {"enc_ms": round(g["encode_ms"], 3), "dec_ms": round(g["decode_ms"], 3)})
        r = d["roofline"]
        print({k: (round(r[k], 4) if isinstance(r[k], float) else r[k]) for k in ("achieved", "frac", "avg_ms", "traffic") if k in r})
    elif "timed region" in ln or "stages ms" in ln:
        print(ln.rstrip()[:300])
