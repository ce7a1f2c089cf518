#!/usr/bin/env python3
"""Fused opening launches with and without the flag step (developer probe; run under rocprofv3 --pmc FETCH_SIZE etc.).

    python tools/fused_probe.py --radii 1,2,3,4,8 [--size 16384]
Each radius: smrf_pf_open_flag_f32 with a mask (flag step: re-reads `last`), then with d_mask = NULL (opening only).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--radii", default="1,2,3,4,8")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib, api  # noqa: E402

lib = _lib.load()
n = a.size
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240)).cuda()
assert not api._has_nan(Z)                       # calibration read of 4 n^2 bytes (count_nan kernel)
out = torch.empty_like(Z)
mask = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for r in [int(v) for v in a.radii.split(",")]:
    for with_mask in (1, 0):
        ts = []
        for i in range(a.reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = lib.smrf_pf_open_flag_f32(C.c_void_p(Z.data_ptr()), C.c_void_p(out.data_ptr()),
                                           C.c_void_p(mask.data_ptr()) if with_mask else None, None, .15 * r, 0, n, n, n, 0, n,
                                           0, n, r, st)
            assert rc == 0
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print("R=%d %s: %.4f ms" % (r, "open+flag" if with_mask else "open only", float(np.median(ts))), flush=True)
