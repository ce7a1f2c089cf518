#!/usr/bin/env python3
"""One-off search for parity failures: many random smrf / progressive_filter / inpaint / create_dem cases,
HIP path against the oracle (developer tool; the committed, seeded subset is tests/test_gpu_fuzz.py).

    python tools/fuzz_campaign.py --cases 300 --seed 1
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=200)
ap.add_argument("--seed", type=int, default=1)
ap.add_argument("--kinds", default="smrf,smrf,pf,inpaint,dem", help="comma list drawn from uniformly: smrf pf inpaint dem pssm fda")
ap.add_argument("--only", type=int, default=-1, help="replay: draw every case (same random stream) but run only this one")
ap.add_argument("--dump", default="", help="with --only on an smrf case: save its inputs and both results to this .npz")
a = ap.parse_args()
import neilpy_amd as nz  # noqa: E402
from oracle import smrf_oracle as orc  # noqa: E402

class _Skip(Exception):
    pass


def _smrf_tie_margin(x, y, z, kw, got, want):
    """Smallest distance to its threshold, in the ORACLE's own stages, over the raster cells / points whose flag differs.
    LSQR sums its norms in another order than SciPy's BLAS, the inpainted surfaces differ by ~1e-13, and a comparison
    that sits on its threshold to that accuracy may fall either way; anything farther from the threshold is a real miss."""
    st = orc.smrf(x, y, z, return_stages=True, **kw)[-1]
    cs = kw.get("cellsize", 1)
    win = kw.get("windows", 5)
    win = np.arange(win) + 1 if np.isscalar(win) else np.asarray(win)
    cells = np.argwhere(got[2] != want[2])
    margins = []
    if len(cells):
        A = -st["inpaint1"]
        low = np.abs((A - orc.opening(A, orc.disk(1))) - 5 * cs)
        last = st.get("inpaint1b", st["inpaint1"]).copy()
        pf = np.full(last.shape, np.inf)
        thr = kw.get("slope_threshold", .15) * win * cs
        for i, w in enumerate(win):
            t = orc.opening(last, orc.disk(w))
            pf = np.minimum(pf, np.abs((last - t) - thr[i]))
            last = t if len(win) > 1 else last
        margins += [float(min(low[r, c], pf[r, c])) for r, c in cells]
    pts = np.flatnonzero(np.asarray(got[3]) != np.asarray(want[3]))
    if len(pts) and not len(cells):
        req = kw.get("elevation_threshold", .5) + kw.get("elevation_scaler", 1.25) * st["slope_values"][pts]
        margins += [float(v) for v in np.abs(np.abs(st["elevation_values"][pts] - z[pts]) - req)]
    # a cell that flipped moves the DTM (and points) around it: those follow from the tie, the cell margin covers them
    return max(margins) if margins else None


rng = np.random.default_rng(a.seed)
bad = []
ties = []
counts = {}
t0 = time.time()
for k in range(a.cases):
    kind = str(rng.choice(a.kinds.split(",")))
    counts[kind] = counts.get(kind, 0) + 1
    try:
        if kind == "smrf":
            npts = int(rng.integers(200, 6000))
            ext = (float(rng.uniform(12, 90)), float(rng.uniform(12, 90)))
            x0, y0 = float(rng.choice([0, 1000, 512345.0, 7.25e6])), float(rng.choice([0, 5000, 5403210.0, 1.1e6]))
            x = np.round(rng.uniform(x0, x0 + ext[0], npts), int(rng.integers(1, 4)))
            y = np.round(rng.uniform(y0, y0 + ext[1], npts), int(rng.integers(1, 4)))
            z = 50 + 0.05 * (x - x0) + 3 * np.sin((y - y0) / 7.0)
            z = np.round(z + np.where(rng.random(npts) < .25, rng.uniform(2, 15, npts), rng.normal(0, .05, npts)), 2)
            z[rng.random(npts) < .004] -= rng.uniform(5, 30)
            cs = rng.choice([1, 2, .5, .3, 1.5, .7, 3])
            cs = int(cs) if float(cs).is_integer() else float(cs)
            win = int(rng.integers(1, 10)) if rng.random() < .7 else rng.integers(1, 9, size=int(rng.integers(1, 5)))
            kw = dict(cellsize=cs, windows=win, slope_threshold=float(rng.choice([.1, .15, .3])),
                      elevation_threshold=float(rng.choice([.3, .5, 1.0])), elevation_scaler=float(rng.choice([0, 1.25, 2])),
                      low_outlier_fill=bool(rng.random() < .3))
            if a.only >= 0 and k != a.only:
                raise _Skip
            try:
                want = orc.smrf(x, y, z, **kw)
            except ValueError as e:                       # e.g. fewer than 4 rows for the spline: both must raise
                try:
                    nz.smrf(x, y, z, **kw)
                    bad.append((k, kind, "oracle raised %r, hip did not" % (e,)))
                except ValueError:
                    pass
                continue
            got = nz.smrf(x, y, z, **kw)
            if a.dump:
                np.savez_compressed(a.dump, x=x, y=y, z=z, kw=np.array(repr(kw)), dtm_hip=got[0], dtm_ref=want[0], obj_hip=got[2],
                                    obj_ref=want[2], pts_hip=np.asarray(got[3]), pts_ref=np.asarray(want[3]))
            ok = (got[0].shape == want[0].shape and np.array_equal(got[2], want[2]) and
                  np.array_equal(np.asarray(got[3]), np.asarray(want[3])) and np.abs(got[0] - want[0]).max() <= 1e-7)
            if not ok:
                info = dict(kw=kw, npts=npts, cells=int((got[2] != want[2]).sum()),
                            pts=int((np.asarray(got[3]) != np.asarray(want[3])).sum()), dtm=float(np.abs(got[0] - want[0]).max()))
                margin = _smrf_tie_margin(x, y, z, kw, got, want)
                if margin is not None and margin <= 1e-9:
                    ties.append((k, kind, dict(info, margin=margin)))     # a threshold tie at LSQR rounding level
                else:
                    bad.append((k, kind, dict(info, margin=margin)))
        elif kind == "pf":
            shape = (int(rng.integers(1, 140)), int(rng.integers(1, 200)))
            dt = rng.choice([np.float32, np.float64])
            Z = (rng.normal(0, 1, shape).cumsum(0).cumsum(1) * .05 + 200 + (rng.random(shape) < .06) * rng.uniform(1, 25, shape)).astype(dt)
            win = rng.integers(0, 40, size=int(rng.integers(1, 7)))
            if 4 * min(shape) <= win.max():               # scipy's own reflect bug regime
                win = np.minimum(win, max(0, 4 * min(shape) - 1))
            cs = float(rng.choice([1, .5, 2]))
            if a.only >= 0 and k != a.only:
                raise _Skip
            m, w = nz.progressive_filter(Z, win, cs, .15, return_when_dropped=True)
            m2, w2 = orc.progressive_filter(Z, win, cs, .15, return_when_dropped=True)
            if not (np.array_equal(m, m2) and np.array_equal(w, w2)):
                bad.append((k, kind, dict(shape=shape, win=win.tolist(), dt=str(dt), diff=int((m != m2).sum()))))
        elif kind == "inpaint":
            shape = (int(rng.integers(1, 120)), int(rng.integers(1, 120)))
            A = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * .1 + 100
            A[rng.random(shape) >= rng.uniform(.02, .95)] = np.nan
            if a.only >= 0 and k != a.only:
                raise _Skip
            want, istop, itn = orc.inpaint_nans_by_springs(A, return_info=True)
            got = nz.inpaint_nans_by_springs(A)
            st = nz.last_stats["inpaint"]
            if (st["istop"], st["itn"]) != (istop, itn) or np.abs(got - want).max(initial=0) > 1e-7:
                bad.append((k, kind, dict(shape=shape, got=(st["istop"], st["itn"]), want=(istop, itn),
                                          err=float(np.abs(got - want).max(initial=0)))))
        elif kind == "pssm":
            shape = (int(rng.integers(2, 300)), int(rng.integers(2, 300)))
            Z = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * float(rng.choice([.01, .3, 5])) + 100
            cs = float(rng.choice([1, .5, 2, 5, .3]))
            ve = float(rng.choice([2.3, 1.0, 4.0]))
            if a.only >= 0 and k != a.only:
                raise _Skip
            P = nz.pssm(Z, cellsize=cs, ve=ve, apply_colormap=False)
            P2 = orc.pssm_classes(Z, cs, ve)
            if not np.array_equal(P, P2):
                bad.append((k, kind, dict(shape=shape, cs=cs, ve=ve, diff=int((P != P2).sum()))))
        elif kind == "fda":
            shape = (int(rng.integers(2, 70)), int(rng.integers(2, 70)))
            A = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * .1 + 100
            A[rng.random(shape) >= rng.uniform(.3, .95)] = np.nan
            if a.only >= 0 and k != a.only:
                raise _Skip
            want, istop, itn = orc.inpaint_nans_by_fda(A, return_info=True)
            got = nz.inpaint_nans_by_fda(A)
            st = nz.last_stats["inpaint_fda"]
            scale = max(1.0, float(np.nanmax(np.abs(want)))) if np.isfinite(want).any() else 1.0
            err = float(np.abs(got - want).max(initial=0)) / scale
            if st["istop"] != istop or abs(st["itn"] - itn) > max(3, itn // 100) or not err <= 2e-5:
                bad.append((k, kind, dict(shape=shape, got=(st["istop"], st["itn"]), want=(istop, itn), relerr=err)))
        else:
            npts = int(rng.integers(1, 4000))
            x0 = float(rng.choice([0, -250.5, 512345.0]))
            x = np.round(rng.uniform(x0, x0 + rng.uniform(3, 80), npts), int(rng.integers(0, 4)))
            y = np.round(rng.uniform(10, 10 + rng.uniform(3, 80), npts), int(rng.integers(0, 4)))
            z = rng.normal(0, 10, npts)
            cs = rng.choice([1, 2, .5, .3, .25, 5])
            cs = int(cs) if float(cs).is_integer() else float(cs)
            bt = str(rng.choice(["min", "max"]))
            if a.only >= 0 and k != a.only:
                raise _Skip
            I, t = nz.create_dem(x, y, z, cs, bt)
            I2, t2 = orc.create_dem(x, y, z, cs, bt)
            if not (I.shape == I2.shape and np.array_equal(I, I2, equal_nan=True) and tuple(t)[:6] == tuple(t2)[:6]):
                bad.append((k, kind, dict(npts=npts, cs=cs, bt=bt, shapes=(I.shape, I2.shape))))
    except _Skip:
        continue
    except Exception as e:  # noqa: BLE001
        bad.append((k, kind, "exception %r" % (e,)))
    if (k + 1) % 25 == 0:
        print("%d cases, %d bad, %.0f s" % (k + 1, len(bad), time.time() - t0), flush=True)
print("DONE %d cases %s, %d bad, %d threshold ties" % (a.cases, counts, len(bad), len(ties)))
for b in ties[:40]:
    print("TIE", b)
for b in bad[:40]:
    print("BAD", b)
