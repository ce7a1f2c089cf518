#!/usr/bin/env python3
"""Per-stage timing of the full device SMRF path on synthetic lidar points (secondary metric of SURVEY 8d).

    python tools/smrf_stages.py --points 20000000 --extent 8192 --windows 18
"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--points", type=int, default=20_000_000)
ap.add_argument("--extent", type=float, default=8192.0)
ap.add_argument("--windows", type=int, default=18)
ap.add_argument("--cellsize", type=float, default=1.0)
a = ap.parse_args()

import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import api, _lib  # noqa: E402

x, y, z = neilpy_amd.synth_points(a.points, a.extent, seed=20241)
t0 = time.perf_counter()
xd, yd, zd = api._points_to_device(x, y, z)
torch.cuda.synchronize()
stages = {"upload_ms": (time.perf_counter() - t0) * 1e3}
lib = _lib.load()


def timed(name, fn):
    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    stages[name] = (time.perf_counter() - t) * 1e3
    return out


cs = int(a.cellsize) if a.cellsize == int(a.cellsize) else a.cellsize
windows = np.arange(a.windows) + 1
Zmin, empty, t = timed("create_dem_ms", lambda: api._create_dem_device(xd, yd, zd, cs, "min", None))
rows, cols = Zmin.shape
stages["grid"] = [rows, cols]
stages["empty_fraction"] = float(empty.float().mean().item())
timed("inpaint1_ms", lambda: api._springs_device(Zmin, "inpaint1"))
stages["inpaint1"] = dict(api.last_stats["inpaint1"])


def low_filter():
    neg = torch.empty_like(Zmin)
    _lib.check(lib.smrf_negate_f64(api._ptr(Zmin), api._ptr(neg), Zmin.numel(), api._stream()))
    return api._progressive_filter_device(neg, np.array([1]), 5 * (np.array([1]) * cs), False, nan_aware=0)[0]


low = timed("low_outlier_filter_ms", low_filter)
obj = timed("progressive_filter_f64_ms",
            lambda: api._progressive_filter_device(Zmin, windows, .15 * (windows * cs), False, nan_aware=0)[0])
object_cells = torch.empty_like(obj)
timed("mask_apply_ms", lambda: _lib.check(lib.smrf_mask_apply_f64(api._ptr(Zmin), api._ptr(empty), api._ptr(low), api._ptr(obj),
                                                                  api._ptr(object_cells), Zmin.numel(), api._stream())))
stages["object_fraction"] = float(object_cells.float().mean().item())
timed("inpaint2_ms", lambda: api._springs_device(Zmin, "inpaint2"))
stages["inpaint2"] = dict(api.last_stats["inpaint2"])
t_all = time.perf_counter()
out = neilpy_amd.smrf(xd, yd, zd, cellsize=cs, windows=a.windows)
torch.cuda.synchronize()
stages["smrf_total_ms"] = (time.perf_counter() - t_all) * 1e3
stages["points"] = a.points
stages["Mpoints_per_s"] = a.points / stages["smrf_total_ms"] / 1e3
stages["object_points"] = int(out[3].sum().item())          # CUDA tensors in -> CUDA tensors out
print(json.dumps(stages))
