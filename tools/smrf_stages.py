#!/usr/bin/env python3
"""Per-stage timing of the full device SMRF path on synthetic lidar points (secondary metric of SURVEY 8d).

    python tools/smrf_stages.py --points 20000000 --extent 8192 --windows 18

``run()`` is also what bench.py's ``secondary`` block calls.  Stages are timed one by one (synchronise, wall clock)
on device-resident points, then the public ``neilpy_amd.smrf`` call is timed as a whole on the same tensors.
LSQR figures: ``ms_per_iteration`` and ``gbps`` = 99 B x raster cells / that time - the bytes one iteration of the
matrix-free solver moves over its planes since round 5 (u two planes read + written and read again, v read twice and
written, w read + written, x read + written every second iteration, plus the hole bytes: DESIGN 4.3,
profiles/r05_lsqr_split.md; round 4's form moved 106, counter-verified at 106.9 on a raster with 74 % holes,
profiles/r04_lsqr_traffic.md) - an upper bound on rasters with few holes.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

LSQR_BYTES_PER_CELL_ITER = 99.0


def run(points=20_000_000, extent=8192.0, windows=18, cellsize=1.0, seed=20241, warm=False):
    import torch
    import neilpy_amd
    from neilpy_amd import api, _lib

    x, y, z = neilpy_amd.synth_points(points, extent, seed=seed)
    t0 = time.perf_counter()
    xd, yd, zd = api._points_to_device(x, y, z)
    torch.cuda.synchronize()
    stages = {"upload_ms": (time.perf_counter() - t0) * 1e3}
    del x, y, z
    lib = _lib.load()

    def timed(name, fn):
        for _ in range(2 if warm else 1):                  # --warm: every stage twice, the second time counts (kernels loaded,
            torch.cuda.synchronize()                       # occupancy queried, allocator warm: what a long-running caller sees)
            t = time.perf_counter()
            out = fn()
            torch.cuda.synchronize()
            stages[name] = (time.perf_counter() - t) * 1e3
        return out

    cs = int(cellsize) if cellsize == int(cellsize) else cellsize
    win = np.arange(windows) + 1
    Zmin, empty, t = timed("create_dem_ms", lambda: api._create_dem_device(xd, yd, zd, cs, "min", None))
    rows, cols = Zmin.shape
    stages["grid"] = [rows, cols]
    stages["empty_fraction"] = float(empty.float().mean().item())

    def lsqr(name):
        if warm:                                           # (the solver fills Zmin in place: its warm-up run gets a copy)
            api._springs_device(Zmin.clone(), name)
        torch.cuda.synchronize()
        t = time.perf_counter()
        api._springs_device(Zmin, name)
        torch.cuda.synchronize()
        stages[name + "_ms"] = (time.perf_counter() - t) * 1e3
        st = dict(api.last_stats[name])
        st["ms_per_iteration"] = stages[name + "_ms"] / max(1, st["itn"])
        st["gbps"] = LSQR_BYTES_PER_CELL_ITER * rows * cols / (st["ms_per_iteration"] * 1e-3) / 1e9
        stages[name] = st

    lsqr("inpaint1")

    def low_filter():
        neg = torch.empty_like(Zmin)
        _lib.check(lib.smrf_negate_f64(api._ptr(Zmin), api._ptr(neg), Zmin.numel(), api._stream()))
        return api._progressive_filter_device(neg, np.array([1]), 5 * (np.array([1]) * cs), False, nan_aware=0)[0]

    low = timed("low_outlier_filter_ms", low_filter)
    obj = timed("progressive_filter_f64_ms",
                lambda: api._progressive_filter_device(Zmin, win, .15 * (win * cs), False, nan_aware=0)[0])
    object_cells = torch.empty_like(obj)
    timed("mask_apply_ms", lambda: _lib.check(lib.smrf_mask_apply_f64(api._ptr(Zmin), api._ptr(empty), api._ptr(low),
                                                                      api._ptr(obj), api._ptr(object_cells), Zmin.numel(),
                                                                      api._stream())))
    stages["object_fraction"] = float(object_cells.float().mean().item())
    lsqr("inpaint2")
    del Zmin, empty, low, obj, object_cells
    torch.cuda.synchronize()
    t_all = time.perf_counter()
    out = neilpy_amd.smrf(xd, yd, zd, cellsize=cs, windows=windows)
    torch.cuda.synchronize()
    stages["smrf_total_ms"] = (time.perf_counter() - t_all) * 1e3
    stages["points"] = points
    stages["Mpoints_per_s"] = points / stages["smrf_total_ms"] / 1e3
    stages["object_points"] = int(out[3].sum().item())          # CUDA tensors in -> CUDA tensors out
    stages["lsqr_bytes_model"] = ("%.0f B x raster cells per iteration (DESIGN 4.3: 12 plane touches + hole bytes; round 4's form: 106, "
                                  "counter-verified at 106.9 B, profiles/r04_lsqr_traffic.md); ms_per_iteration is stage wall time / "
                                  "iterations, i.e. with the solver's set-up pass and its host polls") % LSQR_BYTES_PER_CELL_ITER
    return stages


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, default=20_000_000)
    ap.add_argument("--extent", type=float, default=8192.0)
    ap.add_argument("--windows", type=int, default=18)
    ap.add_argument("--cellsize", type=float, default=1.0)
    ap.add_argument("--warm", action="store_true", help="time the second run of every stage (a first run loads kernels and queries occupancies: ~5 ms)")
    a = ap.parse_args()
    print(json.dumps(run(a.points, a.extent, a.windows, a.cellsize, warm=a.warm)))
