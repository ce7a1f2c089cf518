#!/usr/bin/env python3
"""Warm wall time of smrf() on 500 000 synthetic points (1025^2 raster), for a kernel trace beside it (timing experiment:
is a small call bound by launches or by kernels?  Round 5: 5.86 ms wall, 5.44 ms of kernels in 564 launches - the LSQR's
streaming kernels at ~4.5 TB/s out of the caches and 1.1 ms of one-block reduce + scalar kernels).

    rocprofv3 --kernel-trace --stats -- python3 tools/experiments/smrf_small.py
"""
import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, neilpy_amd
from neilpy_amd import api
x, y, z = neilpy_amd.synth_points(500000, 1024.0, seed=20241)
xd, yd, zd = api._points_to_device(x, y, z)
for _ in range(3):
    out = neilpy_amd.smrf(xd, yd, zd, cellsize=1, windows=18)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    out = neilpy_amd.smrf(xd, yd, zd, cellsize=1, windows=18)
torch.cuda.synchronize()
print("smrf warm: %.2f ms" % ((time.perf_counter() - t) * 100))
