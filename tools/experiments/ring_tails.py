#!/usr/bin/env python3
"""When and where the workgroups of one ring launch ran (timing experiment).  Needs a library built with -DSMRF_RING_DBG_TS:
every workgroup leaves its start / end time (100 MHz counter), XCD, HW_ID, dispatch id and segment length in the first six
cells of its segment's first row (the results are wrong there).

    python -m neilpy_amd.build --variant ts --defs=-DSMRF_RING_DBG_TS
    NEILPY_AMD_LIB=neilpy_amd/_lib/variants/ts.so python tools/experiments/ring_tails.py --shape 16384x16384 --slopes 0,60 15 30 40 50
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
ap = argparse.ArgumentParser()
ap.add_argument("radii", nargs="*", type=int, default=[15, 30, 40, 50])
ap.add_argument("--shape", default="16384x16384")
ap.add_argument("--slopes", default="0")
ap.add_argument("--dil", default="0")
ap.add_argument("--dtype", default="f32")
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd as nz  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
rows, cols = (int(v) for v in a.shape.split("x"))
npdt = np.float32 if a.dtype == "f32" else np.float64
Z = torch.from_numpy(nz.synth_dem(cols, seed=20240, rows=rows, dtype=npdt)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
fn = getattr(lib, "smrf_disk_filter_" + a.dtype)
strips = (cols + 255) // 256
for r in a.radii:
    for dil in [int(v) for v in a.dil.split(",")]:
        for slope in [int(v) for v in a.slopes.split(",")]:
            os.environ["SMRF_RING_SLOPE"] = str(slope)
            _lib.reload_switches()
            out = torch.empty_like(Z)
            for rep in range(3):
                out.fill_(-1.0)
                assert fn(C.c_void_p(Z.data_ptr()), C.c_void_p(out.data_ptr()), rows, cols, cols, 0, rows, 0, rows, r, dil, 0, 1, st) == 0
            torch.cuda.synchronize()
            o = out.cpu().numpy().astype(np.float64)
            recs = []
            for s in range(strips):
                x0 = s * 256
                if x0 + 6 > cols:
                    continue
                blk = o[:, x0:x0 + 6]
                t0, t1, xcc, hw, wid, ln = (blk[:, k] for k in range(6))
                ok = ((xcc >= 0) & (xcc <= 7) & (xcc == np.floor(xcc)) & (t0 == np.floor(t0)) & (t1 == np.floor(t1)) & (t0 > 0) & (t1 >= t0)
                      & (ln > 0) & (ln == np.floor(ln)) & (wid >= 0) & (wid == np.floor(wid)) & (hw == np.floor(hw)) & (hw >= 0) & (t1 - t0 < 1e6))
                for y in np.where(ok)[0]:
                    recs.append((t0[y], t1[y], xcc[y], hw[y], wid[y], ln[y], y, s))
            R_ = np.array(recs)
            t0, t1, xcc, hw, wid, ln = (R_[:, k] for k in range(6))
            hw = hw.astype(int)
            cu = (xcc.astype(int) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)    # XCD, SE, SH, CU
            base = t0.min()
            s_us, e_us = (t0 - base) / 100.0, (t1 - base) / 100.0
            dur = e_us - s_us
            span = e_us.max()
            n = len(dur)
            print("R=%d %s slope %d, %s: %d workgroups on %d CUs (%s per CU); span %.1f us; last start %.1f; ends: first %.1f median %.1f "
                  "p90 %.1f last %.1f; mean resident fraction of the span %.3f"
                  % (r, "dilation" if dil else "erosion", slope, a.shape, n, len(set(cu)),
                     "/".join("%d:%d" % (k, v) for k, v in sorted(zip(*np.unique(np.unique(cu, return_counts=True)[1], return_counts=True)))),
                     span, s_us.max(), e_us.min(), np.median(e_us), np.percentile(e_us, 90), e_us.max(), dur.sum() / (span * n)))
            # age rank of a workgroup on its CU (by dispatch id) against dispatch id / 256
            rank = np.zeros(n, int)
            for c in set(cu):
                idx = np.where(cu == c)[0]
                rank[idx[np.argsort(wid[idx])]] = np.arange(len(idx))
            cls = (wid // 256).astype(int)
            print("   rank on the CU == dispatch id / 256 for %.1f %% of the workgroups" % (100.0 * (rank == cls).mean()))
            for k in range(rank.max() + 1):
                m = rank == k
                print("   rank %d: %5d workgroups, rows %6.0f, duration %7.1f us (%.3f us / row), end %7.1f .. %7.1f us"
                      % (k, m.sum(), ln[m].mean(), dur[m].mean(), dur[m].sum() / ln[m].sum(), e_us[m].min(), e_us[m].max()))
            print("   by XCD: duration " + " ".join("%.0f" % dur[xcc == k].mean() for k in range(8)) + "; last end "
                  + " ".join("%.0f" % e_us[xcc == k].max() for k in range(8)), flush=True)
