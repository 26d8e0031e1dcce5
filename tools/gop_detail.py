#!/usr/bin/env python3
"""Per-op HIP-event table of one encode + decode of the reference's operating point (GOP of the first 5 recorded ZED
frames, tests/golden/zed_seq25.npz), like tools/step_detail.py for the 1M-point frame.
    python tools/gop_detail.py [container_version]"""
import importlib
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
pkg = importlib.import_module(PKG)
S = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
with np.load(os.path.join(ROOT, "tests", "golden", "zed_seq25.npz")) as f:
    zed = [{"points": f[f"points_{i}"], "colors": f[f"colors_u8_{i}"].astype(np.float64) / 255.0} for i in range(5)]
cv = int(sys.argv[1]) if len(sys.argv) > 1 else 0
enc = pkg.CompressionPipeline(S, slots=1, container_version=cv)
dec = pkg.DecompressionPipeline(slots=1)
gop = lambda: {"frames": [dict(f) for f in zed], "timestamps": {}}   # noqa: E731
for _ in range(5):
    out, side = enc.compress(gop())
    rec, ds = dec.decompress(out[3])
te, td = [], []
for _ in range(20):
    t0 = time.perf_counter(); out, side = enc.compress(gop()); t1 = time.perf_counter(); rec, ds = dec.decompress(out[3]); t2 = time.perf_counter()
    te.append(1e3 * (t1 - t0)); td.append(1e3 * (t2 - t1))
print("container version", cv, "points", sum(len(f["points"]) for f in zed), "encode ms median", round(float(np.median(te)), 3),
      "decode ms median", round(float(np.median(td)), 3))
rts = enc.runtimes + dec.runtimes
for r in rts:
    r.prof_enable(True, reserve=600)
out, side = enc.compress(gop())
rec, ds = dec.decompress(out[3])
torch.cuda.synchronize()
print({k: (round(1e3 * v, 3) if not isinstance(v, list) else None) for k, v in side["enc_time_measurements"].items()})
print({k: round(1e3 * v, 3) for k, v in ds["time_measurements"].items()})
for name, r in (("enc", enc.runtimes[0]), ("dec", dec.runtimes[0])):
    recs = r.prof_records()
    agg = {}
    for op, ms, dims in recs:
        a = agg.setdefault((op, dims), [0, 0.0]); a[0] += 1; a[1] += ms
    print(name, "ops", len(recs), "sum of op events ms", round(sum(ms for _, ms, _ in recs), 3))
    for (op, dims), (c, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"   {op:18s} {str(dims):34s} x{c:2d} {1e3 * ms:8.1f} us")
