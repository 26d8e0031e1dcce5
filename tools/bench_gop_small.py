#!/usr/bin/env python3
"""Latency at the reference's own operating point (evaluation/logs final_test_5fps: GOP of 5 ZED frames,
~13k voxels each, 2-3 qualities; Jetson AGX Orin: encode 841 ms, decode 715 ms per GOP).
Uses the two recorded ZED frames of tests/golden (shifted copies to make 5 distinct frames).
    python tools/bench_gop_small.py            (on an MI355X)
"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("demo-learned-point-cloud-compression_amd")
wl = importlib.import_module("demo-learned-point-cloud-compression_amd.workloads")

with np.load(os.path.join(ROOT, "tests", "golden", "zed_gop2.npz")) as f:
    base = [{"points": f[f"points_{i}"], "colors": f[f"colors_u8_{i}"].astype(np.float64) / 255.0} for i in range(2)]
frames = []
for i in range(5):
    b = base[i % 2]
    frames.append({"points": (b["points"].astype(np.int32) + np.array([3 * i, -2 * i, i])).astype(np.int16),
                   "colors": b["colors"]})
enc = pkg.CompressionPipeline([[1.0, 0.0], [0.0, 1.0], [1, 1]], slots=1)
dec = pkg.DecompressionPipeline(slots=1)
te, td = [], []
for it in range(25):
    gop = wl.gop([dict(f) for f in frames])
    t0 = time.perf_counter()
    out, side = enc.compress(gop)
    t1 = time.perf_counter()
    rec, _ = dec.decompress(out[3])
    t2 = time.perf_counter()
    if it >= 5:
        te.append(1e3 * (t1 - t0))
        td.append(1e3 * (t2 - t1))
n = sum(f["points"].shape[0] for f in frames)
print(f"GOP of 5 frames, {n} voxels, numpy in / numpy out: encode {np.median(te):.2f} ms, decode {np.median(td):.2f} ms "
      f"(Jetson AGX Orin reference: 841 / 715 ms); bpp {[round(b, 2) for b in side['gop_info']['bpp']]}")
