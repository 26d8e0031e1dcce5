#!/usr/bin/env python3
"""Time the latent-sized gather-convolutions of a step one by one (the shapes of g_a's lower levels, h_a and h_s on the
1M-point room frame): levels of stride 4 .. 32, 3^3 32 -> 32 and 32 -> 64 on the level's rule book, 2^3 stride-2 32 -> 32
from the level below.

    python tools/bench_small_conv.py [--reps 50]          # PCC_CONV_ROWS16_MAX=0 for k_gconv16 on every launch
"""
import argparse
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=50)
    ap.add_argument("--points", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch
    runtime = importlib.import_module(PKG + ".runtime")
    sparse = importlib.import_module(PKG + ".sparse")
    wl = importlib.import_module(PKG + ".workloads")
    rt = runtime.Runtime(0)
    with rt:
        frame = wl.room(args.points, seed=0)
        pts = torch.from_numpy(frame["points"].astype(np.int32)).cuda()
        coords = torch.cat([torch.zeros((pts.shape[0], 1), dtype=torch.int32, device="cuda"), pts], 1).contiguous()
        keys = rt.morton_keys(coords)
        rt.sort_pairs(keys)
        cs = sparse.CoordSet(rt, keys, 1, 1).down()[0]          # stride 2
        gw = torch.Generator(device="cuda").manual_seed(1)
        w27 = (torch.randn((27, 32, 32), generator=gw, device="cuda") * 0.05).contiguous()
        w27x64 = (torch.randn((27, 32, 64), generator=gw, device="cuda") * 0.05).contiguous()
        w8 = (torch.randn((8, 32, 32), generator=gw, device="cuda") * 0.1).contiguous()
        b32 = torch.randn((32,), generator=gw, device="cuda").contiguous()
        b64 = torch.randn((64,), generator=gw, device="cuda").contiguous()
        print(f"PCC_CONV_ROWS16_MAX={os.environ.get('PCC_CONV_ROWS16_MAX', '(default)')}")
        while cs.stride < 32:
            fine = cs
            cs, nbr8, _ = fine.down()
            xf = torch.randn((fine.n, 32), generator=gw, device="cuda").contiguous()
            x = torch.randn((cs.n, 32), generator=gw, device="cuda").contiguous()
            nbr = cs.nbr27()
            fns = [("2^3 down 32->32", lambda: rt.sparse_conv(xf, nbr8, w8, b32, True)),
                   ("3^3 32->32", lambda: rt.sparse_conv(x, nbr, w27, b32, True)),
                   ("3^3 32->64", lambda: rt.sparse_conv(x, nbr, w27x64, b64, True))]
            for label, fn in fns:
                for _ in range(3):
                    fn()
                rt.sync()
                rt.timer_start()
                for _ in range(args.reps):
                    fn()
                us = 1e3 * rt.timer_stop_ms() / args.reps
                print(f"stride {cs.stride:3d} rows {cs.n:7d}  {label:16s} {us:7.1f} us", flush=True)
    rt.close()


if __name__ == "__main__":
    main()
