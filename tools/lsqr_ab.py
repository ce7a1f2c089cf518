#!/usr/bin/env python3
"""A/B the spring-inpaint LSQR between library builds in one process (developer tool).

    python tools/lsqr_ab.py --size 8192 --occupancy 0.09 --libs neilpy_amd/_lib/variants/x.so
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--occupancy", type=float, default=0.09)
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--pattern", default="random", choices=["random", "blocks", "objects"],
                help="random: every cell a hole with probability 1 - occupancy; blocks: 64 x 64 blocks are holes with that "
                     "probability (clustered, as after cutting objects out of a DTM); objects: the cells progressive_filter "
                     "flags on the DEM (windows 1..18) are the holes, as in smrf's second inpaint")
ap.add_argument("--libs", default="")
ap.add_argument("--iters", type=int, default=-1, help="iteration limit (timing-only variant builds whose results differ: give every build the same count)")
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
n = a.size
g = torch.Generator(device="cuda").manual_seed(7)
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240).astype(np.float64)).cuda()
if a.pattern == "random":
    Z[torch.rand((n, n), device="cuda", generator=g) >= a.occupancy] = float("nan")
elif a.pattern == "blocks":
    nb = (n + 63) // 64
    coarse = torch.rand((nb, nb), device="cuda", generator=g) >= a.occupancy
    Z[coarse.repeat_interleave(64, 0).repeat_interleave(64, 1)[:n, :n]] = float("nan")
else:
    Z[neilpy_amd.progressive_filter(Z.float(), np.arange(1, 19), 1, .15)] = float("nan")
print("pattern %s: %.1f %% holes" % (a.pattern, 100.0 * float(torch.isnan(Z).float().mean())), flush=True)
fns = {"cur": (lib.smrf_springs_lsqr_f64, lib.smrf_springs_workspace_bytes)}
for path in [v for v in a.libs.split(",") if v]:
    o = C.CDLL(os.path.abspath(path))
    f, w = o.smrf_springs_lsqr_f64, o.smrf_springs_workspace_bytes
    f.restype, f.argtypes = fns["cur"][0].restype, fns["cur"][0].argtypes
    w.restype, w.argtypes = fns["cur"][1].restype, fns["cur"][1].argtypes
    fns[os.path.basename(path)] = (f, w)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
res = {}
order = list(fns.items())
for i in range(a.reps + 1):
    for name, (f, w) in order[i % len(order):] + order[:i % len(order)]:   # the builds take turns at going first
        nbytes = w(n, n)
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        A = Z.clone()
        istop, itn, nunk = C.c_int(0), C.c_int64(0), C.c_int64(0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = f(C.c_void_p(A.data_ptr()), n, n, 1e-6 if a.iters < 0 else 0.0, 1e-6 if a.iters < 0 else 0.0, 1e8 if a.iters < 0 else 0.0,
               a.iters, C.byref(istop), C.byref(itn), C.byref(nunk),
               C.c_void_p(ws.data_ptr()), nbytes, st)
        assert rc == 0
        e1.record()
        torch.cuda.synchronize()
        if i:
            res.setdefault(name, []).append(e0.elapsed_time(e1))
        res[name + "_info"] = (istop.value, itn.value, nunk.value, float(A.sum().item()))
        if name == "cur" and "ref" not in res:
            res["ref"] = A.clone()
        elif "ref" in res:
            res[name + "_equal"] = bool(torch.equal(A, res["ref"]))
            res[name + "_maxdiff"] = float((A - res["ref"]).abs().max().item())
        del ws, A
for name in fns:
    t = float(np.median(res[name]))
    istop, itn, nunk, s = res[name + "_info"]
    print("%-12s %.1f ms  istop %d itn %d  %.3f ms/iter  %.0f GB/s at 106 B/cell/iter  sum %.6f" %
          (name, t, istop, itn, t / max(itn, 1), n * n * 106.0 * itn / t / 1e6, s) +
          ("" if name + "_equal" not in res else "  bit-equal to cur: %s (max |diff| %.3g)" % (res[name + "_equal"], res[name + "_maxdiff"])),
          flush=True)
