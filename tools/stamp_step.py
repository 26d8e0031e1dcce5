#!/usr/bin/env python3
"""Shader clock of the dominant conv launch INSIDE a codec step, from a -DPCC_CONV_STAMP=1 build of the library
(diagnostic only): a few compress + decompress steps of the bench frame, then the stamps of the last big k_gconv_up
launch — s_memtime ticks per 100-MHz s_memrealtime tick over each sampled wave's life.
    python tools/stamp_step.py [container_version]"""
import ctypes as C
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    version = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    pkg = importlib.import_module(PKG)
    wl = importlib.import_module(PKG + ".workloads")
    abi = importlib.import_module(PKG + "._abi")
    s = [[1.0, 0.0], [0.0, 1.0], [1, 1]]
    f = wl.room(1_000_000, seed=0)
    enc = pkg.CompressionPipeline(s, slots=1, container_version=version)
    dec = pkg.DecompressionPipeline(slots=1)
    for _ in range(6):
        out, _ = enc.compress({"frames": [dict(f)], "timestamps": {}})
        dec.decompress(out[3])
    n = 4096 * 10
    buf = (C.c_ulonglong * n)()
    fn = abi.lib().pcc_debug_stamps
    fn.restype = C.c_int
    assert fn(buf, n) == 0
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 10).astype(np.float64)
    a = a[a[:, 1] > 0]
    print(f"container version {version}: {a.shape[0]} waves sampled, shader clock {a[:, 1].mean() / 10:.1f} MHz "
          f"(min {a[:, 1].min() / 10:.0f}, max {a[:, 1].max() / 10:.0f})")
    tot = a[:, [0, 2, 3, 4, 5, 8]].sum(1)
    print("  window totals (stamped phases, cycles): mean %.0f  p50 %.0f  p90 %.0f  p99 %.0f  max %.0f; items per window mean %.1f p99 %.0f max %.0f"
          % (tot.mean(), *np.percentile(tot, [50, 90, 99]), tot.max(), a[:, 9].mean(), np.percentile(a[:, 9], 99), a[:, 9].max()))
    span = (a[:, 7].max() - a[:, 6].min()) * 0.01
    life = (a[:, 7] - a[:, 6]) * 0.01
    print("  sampled waves: first start to last end %.1f us; wave life mean %.1f us (p99 %.1f); resident waves (6 x sum of lives / span) %.0f"
          % (span, life.mean(), np.percentile(life, 99), 6 * life.sum() / span))
    names = ["X loads issued", "-", "wait X + sibling product", "loop back", "item 0 + bookkeeping", "items 1-3 + gathers", "-", "-",
             "overflow items", "items (count)"]
    for i, nm in enumerate(names):
        if nm != "-":
            print(f"  {nm:28s} {a[:, i].mean():10.0f} cycles per window")


if __name__ == "__main__":
    main()
