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
