#!/usr/bin/env python3
"""Time the 32->32 gather-convolution alone on the rule books of the 1M-point room frame.

    python tools/bench_conv.py [--reps 20] [--points 1000000]

Cases (rows, how the coordinate set is made):
  cand_pruned : generative children of a RANDOM 407k-row subset of the stride-2 candidates: the
                geometry bench.py's decoder sees with random-init weights (bench.py's dominant launch)
  cand_true   : generative children of the true stride-2 voxels (what trained weights would keep)
  stride2     : the true stride-2 voxels themselves (analysis-side layer)
  stride1     : the input voxels (analysis-side layer, surface statistics)
  stride4 / stride8 : the true stride-4 / stride-8 voxels (the small analysis-side layers; not in the default list)
Prints ms per launch, active pairs and useful TFLOP/s for the plain and the fused-head entry point.
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
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--points", type=int, default=1_000_000)
    ap.add_argument("--cases", default="cand_pruned,cand_true,stride2,stride1")
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
        cs1 = sparse.CoordSet(rt, keys, 1, 1)
        cs2 = cs1.down()[0]
        cs4 = cs2.down()[0]
        g = torch.Generator(device="cpu").manual_seed(0)
        cand2 = cs4.up()
        keep = torch.sort(torch.randperm(cand2.n, generator=g)[:cs2.n]).values.to(torch.int32).cuda()
        pruned2 = cand2.subset(keep)
        sets = {"cand_pruned": lambda: pruned2.up(), "cand_true": lambda: cs2.up(),
                "stride2": lambda: cs2, "stride1": lambda: cs1,
                "stride4": lambda: cs4, "stride8": lambda: cs4.down()[0]}
        gw = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn((27, 32, 32), generator=gw, device="cuda") * 0.05).contiguous()
        b = torch.randn((32,), generator=gw, device="cuda").contiguous()
        hw = torch.randn((32, 1), generator=gw, device="cuda").contiguous()
        hb = torch.zeros((1,), device="cuda")
        for name in args.cases.split(","):
            cs = sets[name]()
            nbr = cs.nbr27()
            pairs = rt.count_nonneg(nbr)
            x = torch.randn((cs.n, 32), generator=gw, device="cuda").contiguous()
            fns = [("conv", lambda: rt.sparse_conv(x, nbr, w, b, True)),
                   ("conv+head", lambda: rt.sparse_conv_head(x, nbr, w, b, True, hw, hb))]
            if name in ("stride4", "stride8"):
                # the latent-sized layers of g_a / h_a / h_s: launches of one partial round of waves
                w64 = (torch.randn((27, 32, 64), generator=gw, device="cuda") * 0.05).contiguous()
                b64 = torch.randn((64,), generator=gw, device="cuda").contiguous()
                fns = [("conv", fns[0][1]), ("conv32x64", lambda: rt.sparse_conv(x, nbr, w64, b64, True))]
            if name == "cand_pruned":
                # the form the native decoder runs: the candidates' rule book formed in-kernel from the parents' book
                pn = pruned2.nbr27()
                fns.append(("head_up", lambda: rt.sparse_conv_head_up(x, pn, w, b, True, hw, hb)))
                # the same with the input rows cold: four input tensors (4 x 417 MB > the 256 MB Infinity Cache) in turn,
                # as in the decoder, where the rows were written once by the up stage in front
                xs = [x] + [torch.randn((cs.n, 32), generator=gw, device="cuda").contiguous() for _ in range(3)]
                turn = [0]

                def cold():
                    turn[0] = (turn[0] + 1) % 4
                    return rt.sparse_conv_head_up(xs[turn[0]], pn, w, b, True, hw, hb)
                fns.append(("head_up_cold", cold))
                # ... and behind the launch that precedes it in the decoder: the up stage writing these rows (its time is
                # taken off with a run of the up stage alone)
                par = torch.randn((pruned2.n, 32), generator=gw, device="cuda").contiguous()
                w8 = (torch.randn((8, 32, 32), generator=gw, device="cuda") * 0.1).contiguous()
                fns.append(("up alone", lambda: rt.convT_gen(par, w8, b, True)))

                def chained():
                    xin = rt.convT_gen(par, w8, b, True)
                    return rt.sparse_conv_head_up(xin, pn, w, b, True, hw, hb)
                fns.append(("up+head_up", chained))
            for label, fn in fns:
                for _ in range(3):
                    fn()
                rt.sync()
                rt.timer_start()
                for _ in range(args.reps):
                    fn()
                ms = rt.timer_stop_ms() / args.reps
                print(f"{name:12s} {label:9s} rows {cs.n:8d} pairs/row {pairs / cs.n:5.2f}  {ms:7.4f} ms  "
                      f"{2 * pairs * 1024 / ms * 1e-9:6.1f} TFLOP/s useful", flush=True)
            del x, nbr, cs
    rt.close()


if __name__ == "__main__":
    main()
