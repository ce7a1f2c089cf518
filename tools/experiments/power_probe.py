#!/usr/bin/env python3
"""Package power and shader clock (amdgpu hwmon files, sampled from a thread: tools/gpu_power.py) while progressive_filter
runs one class of windows in a loop, beside a device copy and the spring inpainter's LSQR (timing experiment: is the step
power-bound?  profiles/r05_power_bound.md).     python tools/experiments/power_probe.py [--n 16384] [--seconds 4]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=16384)
ap.add_argument("--seconds", type=float, default=4.0)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd as nz  # noqa: E402

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gpu_power import Sampler  # noqa: E402

smp = Sampler(0, 0.01)
print("hwmon:", smp.dir, "cap %s W" % smp.cap_w())
Z = torch.from_numpy(nz.synth_dem(a.n, seed=20240)).cuda()
smp.start()
classes = [("idle", None), ("windows 1..3 (chain3)", np.arange(1, 4)), ("windows 6..10 (singles)", np.arange(6, 11)),
           ("windows 11..14 (fused)", np.arange(11, 15)), ("windows 15..20", np.arange(15, 21)), ("windows 30..38", np.arange(30, 39)),
           ("windows 39..50", np.arange(39, 51)), ("windows 1..50", np.arange(1, 51)), ("torch copy", "copy"),
           ("LSQR (inpaint 8193^2 fp64, 74 % holes)", "lsqr"), ("idle", None)]
g = torch.Generator(device="cuda").manual_seed(5)
A = torch.from_numpy(nz.synth_dem(8193, seed=20240, dtype=np.float64)).cuda()
A[torch.rand(A.shape, device="cuda", generator=g) < 0.74] = float("nan")
marks = []
for name, win in classes:
    t0 = time.time()
    n = 0
    if win is None:
        time.sleep(a.seconds / 2)
    else:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        while time.time() - t0 < a.seconds:
            for _ in range(4):
                if isinstance(win, str) and win == "lsqr":
                    nz.inpaint_nans_by_springs(A)
                elif isinstance(win, str):
                    Z2 = Z.clone()
                else:
                    nz.progressive_filter(Z, win, 1, .15)
                n += 1
            torch.cuda.synchronize()
        ev1.record()
        torch.cuda.synchronize()
        ms = ev0.elapsed_time(ev1) / n
    marks.append((name, t0, time.time(), None if win is None else ms))
smp.stop()
for name, t0, t1, ms in marks:
    r = smp.between(t0 + 0.8, t1)
    print("%-40s %s  %s" % (name, "%8.3f ms / call" % ms if ms else " " * 18,
                           "package %.0f W (max %.0f), sclk %.2f GHz, %d samples" % (r["socket_w"], r["socket_w_max"], r["sclk_ghz"] or 0, r["samples"])
                           if r else "no samples"))
