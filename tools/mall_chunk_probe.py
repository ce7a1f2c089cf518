#!/usr/bin/env python3
"""Developer probe (round 4): does running a window's two ring passes CHUNK BY CHUNK keep the eroded plane in the 256 MB
Infinity Cache and beat the whole-raster passes on the HBM-bound windows (R = 15..23, 22 B per cell and window)?

    python tools/mall_chunk_probe.py [--size 16384] [--radii 15,18,21,23,30] [--chunks 1024,2048,4096]

Per radius: the opening + flag of one window as (a) two whole-raster launches (smrf_disk_filter + smrf_pf_dilate_flag, what
progressive_filter runs) and (b) per chunk of C rows: the erosion of rows [c0 - R, c1 + R) into ONE reused chunk buffer, then
the dilation + flag of rows [c0, c1) from it - the row-band form of the same C entries.  Results are compared bit for bit.
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--radii", default="15,18,21,23,30")
ap.add_argument("--chunks", default="1024,2048,4096")
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
n = a.size
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240)).cuda()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
E = torch.empty_like(Z)
O = torch.empty_like(Z)
O2 = torch.empty_like(Z)
mask = torch.zeros((n, n), dtype=torch.uint8, device="cuda")
mask2 = torch.zeros_like(mask)
p = lambda t: C.c_void_p(t.data_ptr())


def whole(r, out, m):
    _lib.check(lib.smrf_disk_filter_f32(p(Z), p(E), n, n, n, 0, n, 0, n, r, 0, 0, 0, st))
    _lib.check(lib.smrf_pf_dilate_flag_f32(p(E), p(Z), p(out), p(m), None, 0.15 * r, 1, n, n, n, 0, n, 0, n, r, 0, 0, st))


def chunked(r, c, out, m, buf):
    for c0 in range(0, n, c):
        c1 = min(n, c0 + c)
        q0, q1 = max(0, c0 - r), min(n, c1 + r)                    # eroded rows the chunk's dilation needs
        lo, hi = max(0, q0 - r), min(n, q1 + r)                    # rows of `last` their erosion needs
        _lib.check(lib.smrf_disk_filter_f32(C.c_void_p(Z.data_ptr() + lo * n * 4), p(buf), n, n, n, lo, hi - lo, q0, q1 - q0, r,
                                            0, 0, 0, st))
        _lib.check(lib.smrf_pf_dilate_flag_f32(p(buf), C.c_void_p(Z.data_ptr() + c0 * n * 4), C.c_void_p(out.data_ptr() + c0 * n * 4),
                                               C.c_void_p(m.data_ptr() + c0 * n), None, 0.15 * r, 1, n, n, n, q0, q1 - q0, c0, c1 - c0,
                                               r, 0, 0, st))


def timed(fn):
    ts = []
    for i in range(a.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i:
            ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for r in [int(v) for v in a.radii.split(",")]:
    mask.zero_()
    t_whole = timed(lambda: whole(r, O, mask))
    line = "R=%2d whole %.3f ms" % (r, t_whole)
    for c in [int(v) for v in a.chunks.split(",")]:
        buf = torch.empty((c + 2 * r, n), dtype=torch.float32, device="cuda")
        mask2.zero_()
        t = timed(lambda: chunked(r, c, O2, mask2, buf))
        same = bool(torch.equal(O, O2)) and bool(torch.equal(mask, mask2))
        line += " | chunks of %d rows (%.0f MB eroded): %.3f ms%s" % (c, (c + 2 * r) * n * 4 / 1e6, t, "" if same else " DIFFERENT")
        del buf
    print(line, flush=True)
