#!/usr/bin/env python3
"""Active (row, offset) pairs per eighth of the Morton-sorted rows of bench.py's dominant conv launch: k_gconv16 gives
every XCD one contiguous eighth of the windows (conv16.h), so the launch lasts as long as the heaviest eighth.
    python tools/xcd_balance.py"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    import torch
    runtime = importlib.import_module(PKG + ".runtime")
    sparse = importlib.import_module(PKG + ".sparse")
    wl = importlib.import_module(PKG + ".workloads")
    rt = runtime.Runtime(0)
    with rt:
        frame = wl.room(1_000_000, seed=0)
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
        cs = cand2.subset(keep).up()
        nbr = cs.nbr27()[:, :cs.n]
        present = (nbr >= 0)
        per_row = present.sum(0).to(torch.float64)
        n = cs.n
        # items of 16 slots per 64-row window and offset
        nw = (n + 63) // 64
        pad = torch.zeros((27, nw * 64), dtype=torch.bool, device="cuda")
        pad[:, :n] = present
        cnt = pad.view(27, nw, 64).sum(2)
        items = ((cnt + 15) // 16).clamp(min=1).sum(0).to(torch.float64)   # item 0 of an offset always runs
        empty = (cnt == 0).sum().item()
        print(f"(window, offset) pairs without a present row: {empty} of {27 * nw} ({100.0 * empty / (27 * nw):.2f} %): their item 0 "
              f"runs on pad slots; slots issued / pairs = {16 * items.sum().item() / present.sum().item():.3f}")
        wpx = (nw + 7) // 8
        print(f"rows {n}, windows {nw}, pairs/row {per_row.mean().item():.2f}, items/window {items.mean().item():.2f}")
        for e in range(8):
            sl = slice(e * wpx, min((e + 1) * wpx, nw))
            print(f"XCD {e}: windows {sl.stop - sl.start:6d}  items {items[sl].sum().item():10.0f}  "
                  f"({items[sl].sum().item() / items.sum().item() * 8:.3f} of the mean)")
    rt.close()


if __name__ == "__main__":
    main()
