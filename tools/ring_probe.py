#!/usr/bin/env python3
"""Developer probe: time single disk erosions / dilations per radius on the GPU (events, median).

    python tools/ring_probe.py --n 16384 --radii 1,8,18,50 [--reps 5] [--dilate] [--dtype f32]
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", "--n", dest="n", type=int, default=16384)
    ap.add_argument("--radii", default="1,8,18,32,50")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dilate", action="store_true")
    ap.add_argument("--flag", action="store_true", help="time the dilation + flag step (smrf_pf_dilate_flag)")
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--libs", default="", help="comma list of extra libsmrf_hip builds to interleave (A/B in one process)")
    a = ap.parse_args()
    import torch
    from neilpy_amd import _lib
    lib = _lib.load()
    n = a.n
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    g = torch.Generator(device="cuda").manual_seed(1)
    Z = torch.rand((n, n), dtype=dt, device="cuda", generator=g) * 50 + 300
    out = torch.empty_like(Z)
    fname = ("smrf_pf_dilate_flag_" if a.flag else "smrf_disk_filter_") + a.dtype
    fns = {"cur": getattr(lib, fname)}
    last = (Z + 0.4) if a.flag else None
    mask = torch.zeros((n, n), dtype=torch.uint8, device="cuda") if a.flag else None
    for path in [v for v in a.libs.split(",") if v]:
        other = C.CDLL(os.path.abspath(path))
        f = getattr(other, fname)
        f.restype, f.argtypes = fns["cur"].restype, fns["cur"].argtypes
        fns[os.path.basename(path)] = f
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    elem = Z.element_size()
    # box calibration: plain device copy of the same plane (read + write)
    ts = []
    for i in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out.copy_(Z); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("copy  %.3f ms  %.0f GB/s" % (min(ts), n * n * 2 * elem / min(ts) / 1e6), flush=True)
    for r in [int(v) for v in a.radii.split(",")]:
        ts = {k: [] for k in fns}
        for i in range(a.reps + 1):
            for name, fn in fns.items():           # interleaved rounds: same box, same clocks
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                if a.flag:
                    rc = fn(C.c_void_p(Z.data_ptr()), C.c_void_p(last.data_ptr()), C.c_void_p(out.data_ptr()),
                            C.c_void_p(mask.data_ptr()), None, 0.15 * r, 3, n, n, n, 0, n, 0, n, r, 0, 0, st)
                else:
                    rc = fn(C.c_void_p(Z.data_ptr()), C.c_void_p(out.data_ptr()), n, n, n, 0, n, 0, n, r,
                            int(a.dilate), 0, 0, st)
                assert rc == 0, rc
                e1.record()
                torch.cuda.synchronize()
                if i:
                    ts[name].append(e0.elapsed_time(e1))
        for name in fns:
            t = float(np.median(ts[name]))
            print("r=%2d %-8s %.3f ms  %.0f GB/s (2 plane passes)  %.1f Gcell/s" %
                  (r, name, t, n * n * 2 * elem / t / 1e6, n * n / t / 1e6), flush=True)


if __name__ == "__main__":
    main()
