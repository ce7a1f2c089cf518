#!/usr/bin/env python3
"""A/B the whole progressive_filter between library builds in one process (developer tool).

    python tools/pf_ab.py --size 16384 --windows 50 --libs neilpy_amd/_lib/variants/v7.so
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--windows", type=int, default=50)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--libs", default="")
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
n = a.size
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240)).cuda()
win = np.arange(1, a.windows + 1).astype(np.int32)
thr = (.15 * (win * 1)).astype(np.float64)
nbytes = lib.smrf_progressive_filter_workspace_bytes(n, n, 4)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
mask = torch.empty((n, n), dtype=torch.uint8, device="cuda")
fns = {"cur": lib.smrf_progressive_filter_f32}
for path in [v for v in a.libs.split(",") if v]:
    o = C.CDLL(os.path.abspath(path))
    f = o.smrf_progressive_filter_f32
    f.restype, f.argtypes = fns["cur"].restype, fns["cur"].argtypes
    fns[os.path.basename(path)] = f
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ts = {k: [] for k in fns}
sums = {}
for i in range(a.reps + 1):
    for name, fn in fns.items():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = fn(C.c_void_p(Z.data_ptr()), n, n, win.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p), len(win),
                C.c_void_p(mask.data_ptr()), None, C.c_void_p(ws.data_ptr()), nbytes, 0, 0, st)
        assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts[name].append(e0.elapsed_time(e1))
        sums[name] = int(mask.sum().item())
for name in fns:
    t = float(np.median(ts[name]))
    print("%-10s %.2f ms/call  %.0f Mcells/s  objects %d" % (name, t, n * n / t / 1e3, sums[name]), flush=True)
