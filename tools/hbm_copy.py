#!/usr/bin/env python3
"""What a plain device copy reaches on this box (context for the HBM-bound kernels' fractions of the 8 TB/s peak)."""
import torch
for mb in (64, 256, 1024):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20):
        y.copy_(x)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"copy {mb} MB -> {mb} MB: {ms * 1e3:.1f} us, {2 * mb * 1.048576 / ms:.0f} GB/s read+write")
    z = torch.empty(n // 8, dtype=torch.float32, device="cuda")
    a.record()
    for _ in range(20):
        torch.sum(x.view(8, -1), 0, out=z)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 20
    print(f"read {mb} MB as 8 streams (sum): {ms * 1e3:.1f} us, {1.125 * mb * 1.048576 / ms:.0f} GB/s")
