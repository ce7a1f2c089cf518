#!/usr/bin/env python3
"""Per-window device time of progressive_filter for several library builds on several raster shapes, all interleaved
in one process (developer tool; the per-radius switches in csrc/*.inc are written from its output).

    python tools/window_ab.py --libs neilpy_amd/_lib/variants/tw128.so --shapes 2048x16384,4096x4096 --windows 50 [--fused 0]

Every build is called through ``smrf_progressive_filter_timed_f32|f64`` (an event per window boundary), ``--reps`` times
after one warm-up, builds alternating; the table holds the median per window and the sum.  ``--fused 0|1|2`` sets
SMRF_FUSED for the run (0: every window as two ring passes).  Masks are compared between builds (count + equality).
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--libs", default="")
ap.add_argument("--shapes", default="16384x16384")
ap.add_argument("--windows", type=int, default=50)
ap.add_argument("--first", type=int, default=1)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
ap.add_argument("--fused", default=None)
ap.add_argument("--json", default=None)
a = ap.parse_args()
if a.fused is not None:
    os.environ["SMRF_FUSED"] = a.fused
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
name_fn = "smrf_progressive_filter_timed_" + a.dtype
fns = {"cur": getattr(lib, name_fn)}
for path in [v for v in a.libs.split(",") if v]:
    o = C.CDLL(os.path.abspath(path))
    f = getattr(o, name_fn)
    f.restype, f.argtypes = fns["cur"].restype, fns["cur"].argtypes
    fns[os.path.basename(path).replace(".so", "")] = f
npdt = np.float32 if a.dtype == "f32" else np.float64
esz = 4 if a.dtype == "f32" else 8
win = np.arange(a.first, a.windows + 1).astype(np.int32)
thr = (.15 * (win * 1)).astype(np.float64)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = {}
for shape in a.shapes.split(","):
    rows, cols = (int(v) for v in shape.split("x"))
    Z = torch.from_numpy(neilpy_amd.synth_dem(cols, seed=20240, dtype=npdt, rows=rows)).cuda()
    nbytes = lib.smrf_progressive_filter_workspace_bytes(rows, cols, esz)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    masks = {k: torch.empty((rows, cols), dtype=torch.uint8, device="cuda") for k in fns}
    ts = {k: [] for k in fns}
    order = list(fns.items())
    for i in range(a.reps + 1):
        # the builds take turns at going first: a run's clocks depend on what ran just before it (a build timed always
        # after the slowest one read 3-4 % slow on launches both builds share, round 4)
        for name, fn in order[i % len(order):] + order[:i % len(order)]:
            ms = np.zeros(len(win), dtype=np.float32)
            rc = fn(C.c_void_p(Z.data_ptr()), rows, cols, win.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p),
                    len(win), C.c_void_p(masks[name].data_ptr()), None, C.c_void_p(ws.data_ptr()), nbytes, 0, 0, st,
                    ms.ctypes.data_as(C.c_void_p), None)
            assert rc == 0, (name, rc)
            if i:
                ts[name].append(ms)
    med = {k: np.median(np.stack(v), axis=0) for k, v in ts.items()}
    same = {k: bool(torch.equal(masks[k], masks["cur"])) for k in fns}
    print("== %s %s, windows %d..%d, SMRF_FUSED=%s: masks equal to cur: %s" % (shape, a.dtype, a.first, a.windows, a.fused, same),
          flush=True)
    print("radius " + " ".join("%9s" % k for k in fns))
    for j, r in enumerate(win):
        print("%6d " % r + " ".join("%9.4f" % med[k][j] for k in fns))
    print("   sum " + " ".join("%9.3f" % med[k].sum() for k in fns), flush=True)
    out[shape] = {k: [float(x) for x in med[k]] for k in fns}
    del Z, ws, masks
    torch.cuda.empty_cache()
if a.json:
    with open(a.json, "w") as f:
        json.dump({"windows": [int(w) for w in win], "dtype": a.dtype, "fused": a.fused, "ms": out}, f)
