#!/usr/bin/env python3
"""per-op device times (HIP events around every C-ABI call) of one encode + decode of BASELINE.json configs[4] on one GPU
(4M points as 8 tiles in one GOP)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
pkg = importlib.import_module(PKG); wl = importlib.import_module(PKG + ".workloads"); tiled = importlib.import_module(PKG + ".tiled")
import torch
S = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
tiles, _ = tiled.cut_tiles(wl.fused_scan(4_000_000), (512, 512, 256))
enc = pkg.CompressionPipeline(S, slots=1); dec = pkg.DecompressionPipeline(slots=1, output="numpy")
for it in range(3):
    out, side = enc.compress(wl.gop([dict(f) for f in tiles])); rec, ds = dec.decompress(out[3])
for name, rts, fn in (("encode", enc.runtimes, lambda: enc.compress(wl.gop([dict(f) for f in tiles]))), ("decode", dec.runtimes, lambda: dec.decompress(out[3]))):
    for r in rts: r.prof_enable(True, reserve=600)
    fn(); torch.cuda.synchronize()
    recs = []
    for r in rts:
        recs += r.prof_records(); r.prof_enable(False)
    g = {}
    for op, ms, dims in recs:
        k = (op, dims); g.setdefault(k, [0, 0.0]); g[k][0] += 1; g[k][1] += ms
    tot = sum(v[1] for v in g.values())
    print(f"== {name}: {len(recs)} ops, {tot:.3f} ms of recorded device time")
    for (op, dims), (c, ms) in sorted(g.items(), key=lambda kv: -kv[1][1])[:22]:
        print(f"  {op:18s} {str(dims):40s} x{c:3d} {ms:8.3f} ms")
