#!/usr/bin/env python3
"""One spring-inpaint solve plus calibration passes, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (tools/pmc_lsqr.sh).

The calibration kernel is the library's own smrf_negate_f64 on a plane of the same size: it reads 8 n and writes 8 n bytes
with the solver's access shape (one float64 per lane, 512 B per wave instruction), so its FETCH_SIZE / WRITE_SIZE scale the
counters for this shape (MI355X_MICROARCH, HBM: other widths than 16 B per lane are uncalibrated).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8193)
ap.add_argument("--holes", type=float, default=0.74, help="fraction of cells that are NaN (the 20 M-point benchmark's first inpaint: 0.74)")
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
n = a.size
g = torch.Generator(device="cuda").manual_seed(7)
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240).astype(np.float64)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = torch.empty_like(Z)
_lib.check(lib.smrf_negate_f64(C.c_void_p(Z.data_ptr()), C.c_void_p(out.data_ptr()), n * n, st))   # calibration: 8 n^2 in, 8 n^2 out
torch.cuda.synchronize()
Z[torch.rand((n, n), device="cuda", generator=g) < a.holes] = float("nan")
nbytes = lib.smrf_springs_workspace_bytes(n, n)
ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
istop, itn, nunk = C.c_int(0), C.c_int64(0), C.c_int64(0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
e0.record()
_lib.check(lib.smrf_springs_lsqr_f64(C.c_void_p(Z.data_ptr()), n, n, 1e-6, 1e-6, 1e8, -1, C.byref(istop), C.byref(itn), C.byref(nunk),
                                     C.c_void_p(ws.data_ptr()), nbytes, st))
e1.record()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) * 1e3
print("LSQR n=%d cells=%d istop=%d itn=%d unknowns=%d event_ms=%.2f wall_ms=%.2f (the entry returns after its last synchronisation: "
      "wall = the whole call as a caller sees it)" % (n, n * n, istop.value, itn.value, nunk.value, e0.elapsed_time(e1), wall))
