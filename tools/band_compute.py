#!/usr/bin/env python3
"""Compute-only time of ONE rank's band of the sharded progressive_filter (developer tool).

The halo exchange is stubbed out (margins hold stale rows; results are not checked), so one GPU
can show what each of N ranks would spend in kernels, with and without window grouping:

    python tools/band_compute.py --size 16384 --windows 50 --world 8 --rank 3 [--exchange-us 100]

``--exchange-us T`` makes the stub occupy the stream it is posted on for T microseconds (a device-side spin,
what a send/recv pair looks like to the stream): with ``overlap`` the spin sits on the side stream beside the
interior of the group's last dilation, without it on the main stream in front of the group.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=16384)
ap.add_argument("--windows", type=int, default=50)
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--rank", type=int, default=3)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--exchange-us", type=float, default=0.0)
a = ap.parse_args()
import torch  # noqa: E402
import neilpy_amd  # noqa: E402
from neilpy_amd import sharded  # noqa: E402

n = a.size
b0, b1 = sharded.band_rows(n, a.world, a.rank)
Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240, row_range=(b0, b1))).cuda()
win = np.arange(1, a.windows + 1)
thr = .15 * (win * 1)
calls = [0]


# cycles of torch.cuda._sleep per microsecond on this device
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda._sleep(1000)
e0.record(); torch.cuda._sleep(20_000_000); e1.record(); torch.cuda.synchronize()
cyc_per_us = 20_000_000 / (e0.elapsed_time(e1) * 1e3)


def no_exchange(*args, **kw):
    calls[0] += 1
    if a.exchange_us > 0:
        torch.cuda._sleep(int(a.exchange_us * cyc_per_us))       # on the current stream (the side stream when overlapped)


sharded._exchange = no_exchange
for budget, overlap in ((0, False), (None, False), (None, True), (256, True), (512, False), (512, True)):
    state = {}
    ts, tq = [], []
    for i in range(a.reps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sharded.progressive_filter_sharded(Z, n, win, thr, rank=a.rank, world_size=a.world, state=state, halo_budget=budget,
                                           overlap=overlap)
        t1 = time.perf_counter()                           # all launches enqueued: the host's share
        torch.cuda.synchronize()
        if i:
            ts.append(time.perf_counter() - t0)
            tq.append(t1 - t0)
    print("world %d rank %d band %d rows, halo budget %s, overlap %s, exchange stub %.0f us: %d exchanges, %.2f ms per call "
          "(host enqueue %.2f ms; 1/%d of the single-GPU step would be the ideal)"
          % (a.world, a.rank, b1 - b0, budget, overlap, a.exchange_us, state["exchanges"], 1e3 * float(np.median(ts)),
             1e3 * float(np.median(tq)), a.world), flush=True)
