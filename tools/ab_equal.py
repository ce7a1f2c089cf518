#!/usr/bin/env python3
"""Bit-equality of variant builds with the current library on every ring radius (developer tool, GPU box).

    python tools/ab_equal.py neilpy_amd/_lib/variants/a.so [b.so ...]

Erosion and dilation of a 300 x 777 raster (3 strips; segments that are mostly warm-up rows) by disk(1..64) through
smrf_disk_filter_{f32,f64}, with SMRF_RING_DUAL = 0 (shifting ring) and 1 (the in-place instances forced on the short
segments), the variant against the current library and both against the library's own direct (footprint-gather) kernel.
"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from neilpy_amd import _lib  # noqa: E402

lib = _lib.load()
rng = np.random.default_rng(0)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
rows, cols = 300, 777
all_ok = True
for path in sys.argv[1:]:
    other = C.CDLL(os.path.abspath(path))
    ok = True
    for dt, sfx in ((torch.float32, "f32"), (torch.float64, "f64")):
        Z = torch.from_numpy(rng.normal(0, 1, (rows, cols)).cumsum(0)).to(dt).cuda()
        fa = getattr(lib, "smrf_disk_filter_" + sfx)
        fb = getattr(other, "smrf_disk_filter_" + sfx)
        fb.restype, fb.argtypes = fa.restype, fa.argtypes
        for dual in ("0", "1"):
            os.environ["SMRF_RING_DUAL"] = dual
            lib.smrf_switches_reload()
            if hasattr(other, "smrf_switches_reload"):
                other.smrf_switches_reload()
            for r in range(1, 65):
                for dil in (0, 1):
                    a, b, d = torch.empty_like(Z), torch.empty_like(Z), torch.empty_like(Z)
                    p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
                    assert fa(p(Z), p(a), rows, cols, cols, 0, rows, 0, rows, r, dil, 0, 1, st) == 0
                    assert fb(p(Z), p(b), rows, cols, cols, 0, rows, 0, rows, r, dil, 0, 1, st) == 0
                    assert fa(p(Z), p(d), rows, cols, cols, 0, rows, 0, rows, r, dil, 0, 2, st) == 0    # direct kernel
                    if not torch.equal(a, b) or not torch.equal(b, d):
                        ok = False
                        print("MISMATCH %s %s R=%d %s dual=%s: variant==current %s, current==direct %s, variant==direct %s (%d cells differ)"
                              % (os.path.basename(path), sfx, r, "dilation" if dil else "erosion", dual, torch.equal(a, b),
                                 torch.equal(a, d), torch.equal(b, d), int((b != d).sum())), flush=True)
    os.environ.pop("SMRF_RING_DUAL", None)
    lib.smrf_switches_reload()
    print("%s equals current and the direct kernel on every radius: %s" % (os.path.basename(path), ok), flush=True)
    all_ok = all_ok and ok
sys.exit(0 if all_ok else 1)
