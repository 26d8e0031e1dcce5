#!/usr/bin/env python3
"""cProfile of bench.py's step (compress Q=3 + decompress of the 1M-point room frame): where the host time goes."""
import cProfile
import importlib
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    import torch
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    frame = wl.room(1_000_000, seed=0)
    pts = torch.from_numpy(frame["points"].astype(np.int32)).cuda()
    col = torch.from_numpy(frame["colors"].astype(np.float32)).cuda()
    settings = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
    enc = pkg.CompressionPipeline(settings, device=0, slots=1)
    dec = pkg.DecompressionPipeline(device=0, slots=1, output="device")

    def step():
        out, side = enc.compress({"frames": [{"points": pts, "colors": col}], "timestamps": {}})
        rec, _ = dec.decompress(out[3])
        return side

    for _ in range(5):
        step()
    torch.cuda.synchronize()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    pr.disable()
    st = pstats.Stats(pr)
    st.sort_stats("tottime").print_stats(28)
    st.sort_stats("cumulative").print_stats(40)


if __name__ == "__main__":
    main()
