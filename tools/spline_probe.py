#!/usr/bin/env python3
"""Developer probe: time of the bicubic spline solve per raster size, chunked (smrf_spline_solve_ws_f64) against the
line-by-line form (smrf_spline_solve_f64), and their largest difference.

    python tools/spline_probe.py --sizes 2049,8193,16385
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="1025,2049,4100,8193,16385")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib, spline  # noqa: E402

lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
for n in [int(v) for v in a.sizes.split(",")]:
    Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=3).astype(np.float64)).cuda()
    lu = torch.from_numpy(spline.axis_factors(n)[1]).cuda()
    scratch = torch.empty_like(Z)
    ts = {"sequential": [], "chunked": []}
    res = {}
    for i in range(a.reps + 1):
        for name in ts:
            c = Z.clone()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            if name == "sequential":
                _lib.check(lib.smrf_spline_solve_f64(p(c), n, n, p(lu), p(lu), st))
            else:
                _lib.check(lib.smrf_spline_solve_ws_f64(p(c), p(scratch), n, n, p(lu), p(lu), st))
            e1.record()
            torch.cuda.synchronize()
            if i:
                ts[name].append(e0.elapsed_time(e1))
            res[name] = c
    d = float((res["sequential"] - res["chunked"]).abs().max())
    print("n=%5d  sequential %.3f ms  chunked %.3f ms  (2 x %.2f GB planes)  max |difference| %.2e of %.1f"
          % (n, np.median(ts["sequential"]), np.median(ts["chunked"]), n * n * 8 / 1e9, d, float(res["sequential"].abs().max())),
          flush=True)
