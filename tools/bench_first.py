#!/usr/bin/env python3
"""Time the 4 -> 32 input layer alone on the 1M-point room frame (tools/bench_first.py [--reps 20])."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
import torch  # noqa: E402

rtm = importlib.import_module(PKG + ".runtime")
sp = importlib.import_module(PKG + ".sparse")
wl = importlib.import_module(PKG + ".workloads")
rt = rtm.Runtime(0)
with rt:
    f = wl.room(1_000_000, seed=0)
    pts = torch.from_numpy(f["points"].astype(np.int32)).cuda()
    coords = torch.cat([torch.zeros((pts.shape[0], 1), dtype=torch.int32, device="cuda"), pts], 1).contiguous()
    keys = rt.morton_keys(coords)
    rt.sort_pairs(keys)
    cs = sp.CoordSet(rt, keys, 1, 1)
    nbr = cs.nbr27()
    pairs = rt.count_nonneg(nbr)
    x = torch.rand((cs.n, 4), device="cuda")
    w = torch.randn((27, 4, 32), device="cuda") * 0.1
    b = torch.randn(32, device="cuda")
    for _ in range(3):
        rt.sparse_conv(x, nbr, w, b, True)
    rt.sync()
    rt.timer_start()
    for _ in range(20):
        rt.sparse_conv(x, nbr, w, b, True)
    ms = rt.timer_stop_ms() / 20
    nbytes = 4 * (cs.n * 4 + cs.n * 32) + 4 * 27 * 4 * 32 + 4 * 27 * cs.n
    print(f"first layer: {cs.n} rows, {pairs / cs.n:.2f} pairs/row, {ms * 1e3:.1f} us, "
          f"{nbytes / ms / 1e6:.0f} GB/s of {nbytes / 1e6:.0f} MB (features in/out + full rule book)")
rt.close()
