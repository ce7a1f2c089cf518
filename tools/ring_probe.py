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
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--radii", default="1,8,18,32,50")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--dilate", action="store_true")
    ap.add_argument("--dtype", default="f32")
    a = ap.parse_args()
    import torch
    import neilpy_amd
    from neilpy_amd import _lib
    lib = _lib.load()
    n = a.n
    dt = torch.float32 if a.dtype == "f32" else torch.float64
    g = torch.Generator(device="cuda").manual_seed(1)
    Z = torch.rand((n, n), dtype=dt, device="cuda", generator=g) * 50 + 300
    out = torch.empty_like(Z)
    fn = getattr(lib, "smrf_disk_filter_" + a.dtype)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    elem = Z.element_size()
    for r in [int(v) for v in a.radii.split(",")]:
        ts = []
        for i in range(a.reps + 1):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            _lib.check(fn(C.c_void_p(Z.data_ptr()), C.c_void_p(out.data_ptr()), n, n, n, 0, n, 0, n, r,
                          int(a.dilate), 0, 0, st))
            e1.record()
            torch.cuda.synchronize()
            if i:
                ts.append(e0.elapsed_time(e1))
        t = float(np.median(ts))
        print("r=%2d  %.3f ms  %.0f GB/s (2 plane passes)  %.1f Gcell/s" %
              (r, t, n * n * 2 * elem / t / 1e6, n * n / t / 1e6), flush=True)


if __name__ == "__main__":
    main()
