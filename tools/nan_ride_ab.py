#!/usr/bin/env python3
"""The public progressive_filter call with the NaN scan riding in the first chained launch (default) against a separate
count pass (SMRF_NAN_RIDE=0), alternating in one process (developer tool).

    python tools/nan_ride_ab.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import neilpy_amd  # noqa: E402

Z = torch.from_numpy(neilpy_amd.synth_dem(16384, seed=20240)).cuda()
win = np.arange(1, 51)
for mode in ("1", "0", "1", "0"):
    os.environ["SMRF_NAN_RIDE"] = mode
    neilpy_amd._lib.reload_switches()
    for _ in range(2):
        neilpy_amd.progressive_filter(Z, win, 1, .15)
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        m = neilpy_amd.progressive_filter(Z, win, 1, .15)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("SMRF_NAN_RIDE=%s: median %.3f ms  objects %d" % (mode, float(np.median(ts)), int(m.sum())), flush=True)
