#!/usr/bin/env python3
"""End-to-end smrf() on NumPy inputs (the drop-in boundary: points cross PCIe in, rasters and flags out)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import neilpy_amd  # noqa: E402

npts, extent = 20_000_000, 8192.0
x, y, z = neilpy_amd.synth_points(npts, extent, seed=20241)
for rep in range(3):
    t0 = time.perf_counter()
    out = neilpy_amd.smrf(x, y, z, cellsize=1, windows=18)
    torch.cuda.synchronize()
    print("run %d: smrf on %d NumPy points -> %s grid: %.1f ms (%.1f Mpoints/s), %d object points"
          % (rep, npts, out[0].shape, (time.perf_counter() - t0) * 1e3, npts / (time.perf_counter() - t0) / 1e6,
             int(np.sum(out[3]))), flush=True)

# the headline workload through the NumPy boundary (PCIe-inclusive; never the bench's `value`)
n = 16384
Z = neilpy_amd.synth_dem(n, seed=20240)
win = np.arange(1, 51)
for rep in range(3):
    t0 = time.perf_counter()
    m = neilpy_amd.progressive_filter(Z, win, 1, .15)
    dt = time.perf_counter() - t0
    print("run %d: progressive_filter %dx%d fp32 NumPy in -> bool NumPy out: %.1f ms = %.0f Mcells/s (PCIe inclusive), %d objects"
          % (rep, n, n, dt * 1e3, n * n / dt / 1e6, int(m.sum())), flush=True)
