#!/usr/bin/env python3
"""Developer probe for the two-role ring experiment (tools/experiments/morph_ring2.h, built by
tools/experiments/build_ring2.sh): per radius, bit-equality of the eroded plane, the opened plane and the flag mask with the
product ring kernels, and the time of each launch (events, median, interleaved in one process).

    python tools/ring2_probe.py --n 16384 --radii 20,32,40,50 [--reps 5] [--exp tools/experiments/libring2_exp_np3.so]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--rows", type=int, default=0)
ap.add_argument("--radii", default="20,32,40,50")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--exp", default=os.path.join(ROOT, "tools", "experiments", "libring2_exp.so"))
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
exp = C.CDLL(os.path.abspath(a.exp))
exp.smrf_exp_ring2_f32.restype = C.c_int
exp.smrf_exp_ring2_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_void_p]
n = a.n
m = a.rows or n
Z = torch.from_numpy(neilpy_amd.synth_dem(max(n, m), seed=20240)[:m, :n].copy()).cuda()
last = Z + 0.4
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
bad = 0
for r in [int(v) for v in a.radii.split(",")]:
    res = {}
    ts = {(k, w): [] for k in ("ring", "ring2") for w in ("erode", "dilate+flag")}
    for i in range(a.reps + 1):
        for which in ("ring", "ring2"):
            er = torch.empty_like(Z)
            op = torch.empty_like(Z)
            mask = torch.zeros((m, n), dtype=torch.uint8, device="cuda")
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e0.record()
            if which == "ring":
                rc = lib.smrf_disk_filter_f32(C.c_void_p(Z.data_ptr()), C.c_void_p(er.data_ptr()), m, n, n, 0, m, 0, m, r, 0, 0, 0, st)
            else:
                rc = exp.smrf_exp_ring2_f32(Z.data_ptr(), None, er.data_ptr(), None, 0.0, m, n, r, 0, st)
            assert rc == 0, rc
            e1.record()
            if which == "ring":
                rc = lib.smrf_pf_dilate_flag_f32(C.c_void_p(er.data_ptr()), C.c_void_p(last.data_ptr()), C.c_void_p(op.data_ptr()),
                                                 C.c_void_p(mask.data_ptr()), None, 0.15 * r, 3, m, n, n, 0, m, 0, m, r, 0, 0, st)
            else:
                rc = exp.smrf_exp_ring2_f32(er.data_ptr(), last.data_ptr(), op.data_ptr(), mask.data_ptr(), 0.15 * r, m, n, r, 1, st)
            assert rc == 0, rc
            e2.record()
            torch.cuda.synchronize()
            if i:
                ts[(which, "erode")].append(e0.elapsed_time(e1))
                ts[(which, "dilate+flag")].append(e1.elapsed_time(e2))
            else:
                res[which] = (er, op, mask)
    same = [bool(torch.equal(x, y)) for x, y in zip(res["ring"], res["ring2"])]
    bad += not all(same)
    t = {k: float(np.median(v)) for k, v in ts.items()}
    print("r=%2d  erode %.3f -> %.3f ms (%+.1f %%)   dilate+flag %.3f -> %.3f ms (%+.1f %%)   equal: eroded %s opened %s mask %s  "
          "(mask sum %d)" % (r, t[("ring", "erode")], t[("ring2", "erode")], 100 * (t[("ring2", "erode")] / t[("ring", "erode")] - 1),
                             t[("ring", "dilate+flag")], t[("ring2", "dilate+flag")],
                             100 * (t[("ring2", "dilate+flag")] / t[("ring", "dilate+flag")] - 1), *same, int(res["ring"][2].sum())),
          flush=True)
print("MISMATCHES" if bad else "all equal")
sys.exit(1 if bad else 0)
