#!/usr/bin/env python3
"""Per-window device time of progressive_filter for several values of one of the library's SMRF_* switches (default
SMRF_RING_SLOPE: segments of unequal length, morph_ring.h ring_launch_np) in ONE process and ONE library: the switch is
re-read between calls (smrf_switches_reload), the values take turns at going first.  Masks are compared with the first value's.

    python tools/experiments/switch_sweep.py --slopes 0,40,60,80 --shapes 16384x16384 --first 15 --windows 50 [--fused 0]
    python tools/experiments/switch_sweep.py --switch SMRF_SEG_RULE --slopes 2,0 --shapes 8193x8193,10000x12000
"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("--switch", default="SMRF_RING_SLOPE")
ap.add_argument("--slopes", default="0,40,60,80", help="the values of the switch")
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
fn = getattr(lib, "smrf_progressive_filter_timed_" + a.dtype)
npdt = np.float32 if a.dtype == "f32" else np.float64
esz = 4 if a.dtype == "f32" else 8
win = np.arange(a.first, a.windows + 1).astype(np.int32)
thr = (.15 * (win * 1)).astype(np.float64)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
slopes = [int(v) for v in a.slopes.split(",")]
out = {}
for shape in a.shapes.split(","):
    rows, cols = (int(v) for v in shape.split("x"))
    Z = torch.from_numpy(neilpy_amd.synth_dem(cols, seed=20240, dtype=npdt, rows=rows)).cuda()
    nbytes = lib.smrf_progressive_filter_workspace_bytes(rows, cols, esz)
    ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    masks = {k: torch.empty((rows, cols), dtype=torch.uint8, device="cuda") for k in slopes}
    ts = {k: [] for k in slopes}
    for i in range(a.reps + 1):
        for k in slopes[i % len(slopes):] + slopes[:i % len(slopes)]:
            os.environ[a.switch] = str(k)
            _lib.reload_switches()
            ms = np.zeros(len(win), dtype=np.float32)
            rc = fn(C.c_void_p(Z.data_ptr()), rows, cols, win.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p),
                    len(win), C.c_void_p(masks[k].data_ptr()), None, C.c_void_p(ws.data_ptr()), nbytes, 0, 0, st,
                    ms.ctypes.data_as(C.c_void_p), None)
            assert rc == 0, (k, rc)
            if i:
                ts[k].append(ms)
    med = {k: np.median(np.stack(v), axis=0) for k, v in ts.items()}
    same = {k: bool(torch.equal(masks[k], masks[slopes[0]])) for k in slopes}
    print("== %s %s, windows %d..%d, SMRF_FUSED=%s: %s: masks equal to value %d: %s" % (shape, a.dtype, a.first, a.windows, a.fused,
                                                                                     a.switch, slopes[0], same), flush=True)
    print("radius " + " ".join("%9d" % k for k in slopes) + "   best")
    for j, r in enumerate(win):
        best = min(slopes, key=lambda k: med[k][j])
        print("%6d " % r + " ".join("%9.4f" % med[k][j] for k in slopes) + "   %4d %+5.1f %%" % (
            best, 100 * (med[best][j] / med[slopes[0]][j] - 1)))
    print("   sum " + " ".join("%9.3f" % med[k].sum() for k in slopes), flush=True)
    out[shape] = {str(k): [float(x) for x in med[k]] for k in slopes}
    del Z, ws, masks
    torch.cuda.empty_cache()
if a.json:
    with open(a.json, "w") as f:
        json.dump({"windows": [int(w) for w in win], "dtype": a.dtype, "fused": a.fused, "ms": out}, f)
