#!/usr/bin/env python3
"""Phase timeline of the dominant conv launch from a -DPCC_CONV_STAMP=1 build of the library (diagnostic only).

    python tools/stamp_conv.py     (on the GPU box, with the stamped library in place)

Runs pcc_sparse_conv_head_up on bench.py's candidate geometry and prints, per wave and offset step, the mean
number of shader cycles between the stamps of k_gconv_mfma_compact_w4 (conv_compact.h)."""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"
NAMES = ["top", "compact(k+1)", "load_nb(k+2)", "w/slots/gather issue", "acc read", "gather wait+shape",
         "mfma+write", "pre-barrier", "barrier", "tail"]
if os.environ.get("PCC_CONVUP"):
    NAMES = ["X loads issued", "(s_memtime ticks per 100-MHz tick x 1000)", "wait X + sibling product", "loop back (+ tiles to LDS, prologue)", "tile + G[0] wait, MFMA pair 0",
             "compaction, MFMA pair 1", "requests, MFMA pairs 2-3", "sync + records issue, MFMA pairs 4-7", "write-back, records wait, gather 0",
             "items 1-3 + overflow"]
elif os.environ.get("PCC_CONV16"):
    NAMES = ["loop back", "acc0 read + compact(k+1)", "requests + records", "item 0", "items 1-3 + gathers", "-", "-", "-",
             "-", "-"]


def main():
    import torch
    runtime = importlib.import_module(PKG + ".runtime")
    sparse = importlib.import_module(PKG + ".sparse")
    wl = importlib.import_module(PKG + ".workloads")
    abi = importlib.import_module(PKG + "._abi")
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
        pruned2 = cand2.subset(keep)
        pn = pruned2.nbr27()
        gw = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn((27, 32, 32), generator=gw, device="cuda") * 0.05).contiguous()
        b = torch.randn((32,), generator=gw, device="cuda").contiguous()
        hw = torch.randn((32, 1), generator=gw, device="cuda").contiguous()
        hb = torch.zeros((1,), device="cuda")
        x = torch.randn((8 * pruned2.n, 32), generator=gw, device="cuda").contiguous()
        for _ in range(3):
            rt.sparse_conv_head_up(x, pn, w, b, True, hw, hb)
        rt.sync()
        rt.timer_start()
        rt.sparse_conv_head_up(x, pn, w, b, True, hw, hb)
        ms = rt.timer_stop_ms()
        n = 4096 * 10
        buf = (C.c_ulonglong * n)()
        fn = abi.lib().pcc_debug_stamps
        fn.restype = C.c_int
        assert fn(buf, n) == 0
        a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 10).astype(np.float64)
        a = a[a.sum(1) > 0]
        print(f"launch {ms:.3f} ms (stamped build); {a.shape[0]} waves sampled; cycles per wave, whole kernel and per offset step:")
        tot = a.sum(1).mean()
        for i, nm in enumerate(NAMES):
            print(f"  {nm:24s} {a[:, i].mean():10.0f}  {a[:, i].mean() / 27:8.0f} / step   {100 * a[:, i].mean() / tot:5.1f} %")
        print(f"  {'sum':24s} {tot:10.0f}  {tot / 27:8.0f} / step")
        if os.environ.get("PCC_CONVUP"):
            span = (a[:, 7].max() - a[:, 6].min()) * 0.01
            life = (a[:, 7] - a[:, 6]) * 0.01
            print("  sampled waves: first start to last end %.1f us; wave life mean %.1f us (p99 %.1f); resident waves (6 x sum of lives / span) %.0f"
                  % (span, life.mean(), np.percentile(life, 99), 6 * life.sum() / span))
            t2 = a[:, [0, 2, 3, 4, 5, 8]].sum(1)
            print("  window totals (stamped phases, cycles): mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f; items per window mean %.1f p99 %.0f max %.0f"
                  % (t2.mean(), *np.percentile(t2, [50, 90, 99]), t2.max(), a[:, 9].mean(), np.percentile(a[:, 9], 99), a[:, 9].max()))
    rt.close()


if __name__ == "__main__":
    main()
