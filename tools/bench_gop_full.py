#!/usr/bin/env python3
"""Multi-frame GOPs at full size: encode (Q=3) + decode of a GOP of F room frames of N points each, one MI355X.

    python tools/bench_gop_full.py [--frames 1,2,5] [--points 1000000] [--steps 5]

bench.py's headline is F=1 (BASELINE.json configs[1]); the reference's services code GOPs of several frames
(shared/config.yaml gop_size), which amortises the per-GOP host work and makes every kernel launch F times
larger.  Prints one line per F: ms per GOP, frames/s, bits per point of the decoded quality.
"""
import argparse
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
SETTINGS = [[1.0, 0.0], [0.0, 1.0], [1, 1]]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", default="1,2,5")
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=5)
    args = ap.parse_args()
    import torch
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    fmax = max(int(f) for f in args.frames.split(","))
    frames = []
    for s in range(fmax):
        fr = wl.room(args.points, seed=s)
        frames.append({"points": torch.from_numpy(fr["points"].astype(np.int32)).cuda(),
                       "colors": torch.from_numpy(fr["colors"].astype(np.float32)).cuda()})
    enc = pkg.CompressionPipeline(SETTINGS, device=0, slots=1)
    dec = pkg.DecompressionPipeline(device=0, slots=1, output="device")
    for f in (int(v) for v in args.frames.split(",")):
        def step():
            out, side = enc.compress({"frames": [dict(x) for x in frames[:f]], "timestamps": {}})
            rec, dside = dec.decompress(out[len(SETTINGS)])
            return out, side, rec, dside
        for _ in range(2):
            out, side, rec, dside = step()
        assert len(rec) == f and all(r["points"].shape[0] == fr["points"].shape[0] for r, fr in zip(rec, frames))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        em, dm = [], []
        for _ in range(args.steps):
            out, side, rec, dside = step()
            em.append(side["timestamps"]["codec_end"] - side["timestamps"]["codec_start"])
            dm.append(dside["timestamps"]["codec_end"] - dside["timestamps"]["codec_start"])
        torch.cuda.synchronize()
        ms = 1e3 * (time.perf_counter() - t0) / args.steps
        n = sum(int(x["points"].shape[0]) for x in frames[:f])
        print(f"F={f}: {ms:8.2f} ms per GOP (encode {1e3 * np.mean(em):.2f} + decode {1e3 * np.mean(dm):.2f}) = "
              f"{1e3 * f / ms:6.1f} frames/s, {ms / f:6.2f} ms per frame, "
              f"{8 * len(out[len(SETTINGS)]) / n:.3f} bpp, peak HBM {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB "
              f"(torch side only)", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
