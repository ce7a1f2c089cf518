#!/usr/bin/env python3
"""Does a working set that fits the 256 MB infinity cache move faster than HBM?  Device copies (read + write) of
planes from 8 MB to 1 GiB, repeated back to back on the same buffers (developer probe).

    python tools/experiments/mall_probe.py
"""
import torch

for mb in (8, 16, 32, 64, 96, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    a = torch.rand(n, device="cuda")
    b = torch.empty_like(a)
    for _ in range(3):
        b.copy_(a)
    reps = max(5, 4096 // mb)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print("plane %5d MB (footprint %5d MB): %.4f ms per copy, %.0f GB/s read+write" % (mb, 2 * mb, t, 2 * n * 4 / t / 1e6), flush=True)
    # producer -> consumer: write plane b (from a), then read b into c: is b still on chip when c reads it?
    c = torch.empty_like(a)
    e0.record()
    for _ in range(reps):
        b.copy_(a)
        c.copy_(b)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    print("      produce + consume: %.4f ms per pair, %.0f GB/s" % (t, 4 * n * 4 / t / 1e6), flush=True)
    del a, b, c
