#!/usr/bin/env python3
"""Shader clock a ring workgroup runs at (timing experiment): needs a library built with -DSMRF_RING_DBG_CLOCK (the
kernel leaves its s_memtime / s_memrealtime deltas in the output's first two cells).
    NEILPY_AMD_LIB=neilpy_amd/_lib/variants/clk.so python tools/experiments/ring_clock.py [radii...]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import neilpy_amd as nz
Z = torch.from_numpy(nz.synth_dem(16384, seed=20240)).cuda()
for r in [int(v) for v in sys.argv[1:]] or [15, 30, 50]:
    for rep in range(3):
        e = nz.erosion(Z, radius=r, impl=1)
    torch.cuda.synchronize()
    c, q = float(e[0, 0]), float(e[0, 1])
    print("R=%d erosion: %.0f shader cycles in %.1f us (100 MHz counter) -> %.3f GHz" % (r, c, q / 100.0, c / q * 0.1))
