#!/usr/bin/env python3
"""Soak test (developer tool): the same progressive_filter call many times, every result compared bit for bit with the
first - a race or a miscounted wait in the kernels would show up as a run that differs.

    python tools/soak.py --size 16384 --windows 50 --reps 300 [--dtype f64]
    python tools/soak.py --kind lsqr --size 4097 --holes 0.7 --reps 100     # the same spring solve again and again
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--windows", type=int, default=50)
ap.add_argument("--reps", type=int, default=300)
ap.add_argument("--dtype", default="f32")
ap.add_argument("--kind", default="pf", choices=["pf", "lsqr"])
ap.add_argument("--holes", type=float, default=0.7)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402

if a.kind == "lsqr":
    g = torch.Generator(device="cuda").manual_seed(11)
    A = torch.from_numpy(neilpy_amd.synth_dem(a.size, seed=20240).astype(np.float64)).cuda()
    A[torch.rand((a.size, a.size), device="cuda", generator=g) < a.holes] = float("nan")
    ref = neilpy_amd.inpaint_nans_by_springs(A)
    st0 = dict(neilpy_amd.last_stats["inpaint"])
    bad, t0 = 0, time.time()
    for i in range(a.reps):
        out = neilpy_amd.inpaint_nans_by_springs(A)
        if not torch.equal(out, ref) or dict(neilpy_amd.last_stats["inpaint"]) != st0:
            bad += 1
            print("rep %d differs: %d cells, stats %s" % (i, int((out != ref).sum()), neilpy_amd.last_stats["inpaint"]), flush=True)
        if (i + 1) % 25 == 0:
            print("%d reps, %d differing, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
    print("DONE lsqr %dx%d %.0f %% holes %s: %d reps, %d differing" % (a.size, a.size, 100 * a.holes, st0, a.reps, bad))
    sys.exit(1 if bad else 0)

Z = torch.from_numpy(neilpy_amd.synth_dem(a.size, seed=20240).astype(np.float32 if a.dtype == "f32" else np.float64)).cuda()
win = np.arange(1, a.windows + 1)
ref_m, ref_w = neilpy_amd.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
bad = 0
t0 = time.time()
for i in range(a.reps):
    m, w = neilpy_amd.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
    if not (torch.equal(m, ref_m) and torch.equal(w, ref_w)):
        bad += 1
        print("rep %d differs: %d mask cells, %d when_dropped cells" % (i, int((m != ref_m).sum()), int((w != ref_w).sum())), flush=True)
    if (i + 1) % 50 == 0:
        print("%d reps, %d differing, %.0f s" % (i + 1, bad, time.time() - t0), flush=True)
print("DONE %s %dx%d windows 1..%d: %d reps, %d differing (objects %d)" % (a.dtype, a.size, a.size, a.windows, a.reps, bad, int(ref_m.sum())))
sys.exit(1 if bad else 0)
