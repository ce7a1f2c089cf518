#!/usr/bin/env python3
"""One progressive_filter step plus a calibration read, for rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE.

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python3 tools/pmc_traffic.py
The count_nan kernel reads exactly 4*n*n bytes with the ring kernels' access shape (one dword per
lane, 256 B per wave instruction): its FETCH_SIZE calibrates the counter (MI355X_MICROARCH, HBM).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--windows", type=int, default=50)
ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import api  # noqa: E402

n = a.size
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240, dtype=np.float32 if a.dtype == "f32" else np.float64)).cuda()
windows = np.arange(1, a.windows + 1)
thr = .15 * (windows * 1)
assert not api._has_nan(Z)                       # calibration read: n*n elements (4 or 8 bytes each)
mask, _ = api._progressive_filter_device(Z, windows, thr, False, nan_aware=0)
torch.cuda.synchronize()
print("objects", int(mask.sum().item()))
