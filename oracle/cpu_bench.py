"""CPU throughput of the oracle's progressive_filter over all cores -- TEST/BENCH INFRASTRUCTURE.

    python -m oracle.cpu_bench --crop 192 --windows 50 --workers 16 --seed 20240

Every worker process runs the oracle (scipy.ndimage grey_erosion / grey_dilation, the primitive
the reference reaches through skimage, oracle/smrf_oracle.py) on its own crop x crop tile of the
benchmark DEM with the full window list; the aggregate is cells of all tiles / wall time.  Tiles
are independent (no halo traffic), which flatters the CPU: a real strip decomposition would
re-compute or exchange 2r rows per window.  Prints one JSON line.  Called by bench.py's
cpu_baseline leg in a child process (the parent has initialised the GPU and must not fork).
"""
import argparse
import json
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _tile(args):
    k, crop, windows, seed = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from neilpy_amd.synth import synth_dem
    from oracle import smrf_oracle as orc
    Z = synth_dem(crop, seed=seed + k, dtype=np.float32)
    win = np.arange(1, windows + 1)
    t0 = time.perf_counter()
    m = orc.progressive_filter(Z, win, 1, .15)
    return time.perf_counter() - t0, int(m.sum())


def usable_cores(cap=16):
    """Cores this process may really use: affinity, then the cgroup CPU quota, then ``cap`` (a GPU
    box hands each one-GPU job about 16 of its host's cores whatever ``sched_getaffinity`` lists)."""
    n = len(os.sched_getaffinity(0))
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: [t.strip(), None])):
        try:
            quota, period = parse(open(path).read())
            if period is None:
                period = open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()
            if quota not in ("max", "-1"):
                n = min(n, max(1, int(int(quota) / int(period))))
            break
        except (OSError, ValueError):
            continue
    return max(1, min(n, cap))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--crop", type=int, default=192)
    ap.add_argument("--windows", type=int, default=50)
    ap.add_argument("--workers", type=int, default=0)
    ap.add_argument("--seed", type=int, default=20240)
    a = ap.parse_args()
    workers = a.workers or usable_cores()
    ctx = mp.get_context("spawn")
    with ctx.Pool(workers) as pool:
        pool.map(_tile, [(k, 32, 2, a.seed) for k in range(workers)])          # start the workers, import scipy
        t0 = time.perf_counter()
        res = pool.map(_tile, [(k, a.crop, a.windows, a.seed) for k in range(workers)], chunksize=1)
        wall = time.perf_counter() - t0
    cells = workers * a.crop * a.crop
    print(json.dumps(dict(value=cells / wall / 1e6, unit="Mcells/s", cores=workers, seconds=round(wall, 3),
                          per_tile_seconds=round(float(np.mean([r[0] for r in res])), 3),
                          sample="%d independent %dx%d fp32 tiles of synth_dem, windows 1..%d, one oracle process per core"
                                 % (workers, a.crop, a.crop, a.windows))))


if __name__ == "__main__":
    main()
