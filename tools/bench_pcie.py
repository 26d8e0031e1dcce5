#!/usr/bin/env python3
"""PCIe-inclusive rate of the 1M-point frame: numpy frames in (as the capturer leaves them), numpy frames out (as
pack_batches returns them).  bench.py's `value` keeps inputs and outputs resident in HBM; this is the figure a caller
of the unmodified reference services would see.  Run from the repository root: python tools/bench_pcie.py"""
import sys, importlib, time, numpy as np, torch
sys.path.insert(0, '.')
pkg = importlib.import_module("demo-learned-point-cloud-compression_amd")
wl = importlib.import_module("demo-learned-point-cloud-compression_amd.workloads")
fr = wl.room(1_000_000, seed=0)
S = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
enc = pkg.CompressionPipeline(S, device=0, slots=1)
dec = pkg.DecompressionPipeline(device=0, slots=1)   # numpy out (reference schema)
for variant, pts, col in (("int16+float64 (reference schema)", fr["points"].astype(np.int16), fr["colors"].astype(np.float64)),
                          ("int32+float32", fr["points"].astype(np.int32), fr["colors"].astype(np.float32))):
    def step():
        out, side = enc.compress({"frames": [{"points": pts, "colors": col}], "timestamps": {}})
        rec, d = dec.decompress(out[3])
        return side, d, rec
    for _ in range(3): side, d, rec = step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    e = []; dd = []
    for _ in range(10):
        side, d, rec = step()
        e.append(side["timestamps"]["codec_end"] - side["timestamps"]["codec_start"]); dd.append(d["timestamps"]["codec_end"] - d["timestamps"]["codec_start"])
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{variant}: {ms:.2f} ms per frame (encode {1e3*np.mean(e):.2f} + decode {1e3*np.mean(dd):.2f}) = {1e3/ms:.1f} frames/s; out types {type(rec[0]['points']).__name__} {rec[0]['points'].dtype} {rec[0]['colors'].dtype}", flush=True)
