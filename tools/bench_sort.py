#!/usr/bin/env python3
"""Time the Morton-key sort of the 1M-point room frame (tools/bench_sort.py)."""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
import torch  # noqa: E402

rtm = importlib.import_module(PKG + ".runtime")
wl = importlib.import_module(PKG + ".workloads")
rt = rtm.Runtime(0)
with rt:
    f = wl.room(1_000_000, seed=0)
    pts = torch.from_numpy(f["points"].astype(np.int32)).cuda()
    coords = torch.cat([torch.zeros((pts.shape[0], 1), dtype=torch.int32, device="cuda"), pts], 1).contiguous()
    keys0 = rt.morton_keys(coords)
    ref = torch.sort(keys0.view(torch.int64) ^ 0)  # keys are below 2^63: signed order == unsigned order
    ts = []
    for it in range(8):
        keys = keys0.clone()
        rt.sync()
        rt.timer_start()
        perm = rt.sort_pairs(keys)
        ts.append(rt.timer_stop_ms())
    assert torch.equal(keys.view(torch.int64), ref.values)
    assert torch.equal(keys0[perm.long()], keys)
    n = keys.shape[0]
    print(f"sort of {n} (key, row) pairs: {np.mean(ts[2:]) * 1e3:.1f} us (min {min(ts) * 1e3:.1f})")
rt.close()
