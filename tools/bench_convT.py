#!/usr/bin/env python3
"""Time the generative up stage (k_convT_mfma) alone at the bench's sizes: python tools/bench_convT.py [parents ...]"""
import importlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "demo-learned-point-cloud-compression_amd"


def main():
    import torch
    runtime = importlib.import_module(PKG + ".runtime")
    rt = runtime.Runtime(0)
    sizes = [int(a) for a in sys.argv[1:]] or [407830, 105749, 26386]
    with rt:
        g = torch.Generator(device="cuda").manual_seed(1)
        w = (torch.randn((8, 32, 32), generator=g, device="cuda") * 0.1).contiguous()
        b = torch.randn((32,), generator=g, device="cuda").contiguous()
        for n in sizes:
            xs = [torch.randn((n, 32), generator=g, device="cuda").contiguous() for _ in range(4)]
            for i in range(3):
                rt.convT_gen(xs[i], w, b, True)
            rt.sync()
            rt.timer_start()
            reps = 20
            for i in range(reps):
                rt.convT_gen(xs[i % 4], w, b, True)
            ms = rt.timer_stop_ms() / reps
            nbytes = 4 * (n * 32 + 8 * n * 32)
            print(f"convT {n:8d} parents  {ms * 1e3:8.1f} us  {nbytes / ms / 1e9:6.2f} TB/s  {2 * 8 * n * 32 * 32 / ms / 1e9:6.1f} TFLOP/s", flush=True)
    rt.close()


if __name__ == "__main__":
    main()
