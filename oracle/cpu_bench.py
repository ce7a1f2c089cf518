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


# ---- one raster, row strips with halo rows: the "all cores" form SURVEY 8d asks for -----------------------------
_SHM = {}


def _attach(names, shape, dtype):
    from multiprocessing import shared_memory
    for key, name in names.items():
        if key not in _SHM:
            shm = shared_memory.SharedMemory(name=name)
            dt = np.uint8 if key == "mask" else dtype
            _SHM[key] = (shm, np.ndarray(shape, dtype=dt, buffer=shm.buf))
    return {k: v[1] for k, v in _SHM.items()}


def _strip_phase(args):
    """erosion (phase 0) or dilation + flag (phase 1) of rows [s0, s1) with r halo rows either side read from the
    shared planes; scipy reflects at the array ends, which are the raster's true borders only for the edge strips
    (interior strips crop the halo rows, so their reflected part never reaches the kept rows)."""
    phase, names, shape, dtype_name, s0, s1, r, thr = args
    os.environ["OMP_NUM_THREADS"] = "1"
    from oracle import smrf_oracle as orc
    a = _attach(names, shape, np.dtype(dtype_name))
    lo, hi = max(0, s0 - r), min(shape[0], s1 + r)
    fp = orc.disk(r)
    if phase == 0:
        a["eroded"][s0:s1] = orc.erosion(a["last"][lo:hi], fp)[s0 - lo:s0 - lo + (s1 - s0)]
    else:
        opened = orc.dilation(a["eroded"][lo:hi], fp)[s0 - lo:s0 - lo + (s1 - s0)]
        a["opened"][s0:s1] = opened
        a["mask"][s0:s1] |= ((a["last"][s0:s1] - opened) > np.float64(thr)).astype(np.uint8)
    return 0


def strips(n, windows, workers, seed):
    """progressive_filter of ONE n x n raster on `workers` processes: per window two parallel phases over row
    strips (erosion, then dilation + flag), the planes in shared memory.  Exact (checked against the
    single-process oracle when n <= 512)."""
    from multiprocessing import shared_memory
    from neilpy_amd.synth import synth_dem
    Z = synth_dem(n, seed=seed, dtype=np.float32)
    shp, names, keep = Z.shape, {}, []
    for key, dt in (("last", Z.dtype), ("eroded", Z.dtype), ("opened", Z.dtype), ("mask", np.uint8)):
        shm = shared_memory.SharedMemory(create=True, size=int(np.prod(shp)) * np.dtype(dt).itemsize)
        keep.append(shm)
        names[key] = shm.name
    arr = {k: np.ndarray(shp, dtype=(np.uint8 if k == "mask" else Z.dtype), buffer=m.buf) for k, m in zip(names, keep)}
    arr["last"][:] = Z
    arr["mask"][:] = 0
    edges = np.linspace(0, n, workers * 4 + 1).astype(int)          # 4 strips per worker: load balance
    ctx = mp.get_context("spawn")
    try:
        with ctx.Pool(workers) as pool:
            pool.map(_strip_phase, [(0, names, shp, Z.dtype.name, 0, 1, 0, 0.0)] * workers)   # start workers, attach
            t0 = time.perf_counter()
            for i, r in enumerate(range(1, windows + 1)):
                thr = .15 * (r * 1)
                for phase in (0, 1):
                    pool.map(_strip_phase, [(phase, names, shp, Z.dtype.name, int(edges[k]), int(edges[k + 1]), r, thr)
                                            for k in range(len(edges) - 1)], chunksize=1)
                arr["last"][:] = arr["opened"]
            wall = time.perf_counter() - t0
        mask = arr["mask"].astype(bool).copy()
    finally:
        for m in keep:
            m.close()
            m.unlink()
    if n <= 512:
        from oracle import smrf_oracle as orc
        assert np.array_equal(mask, orc.progressive_filter(Z, np.arange(1, windows + 1), 1, .15)), "strip form differs"
    return dict(value=n * n / wall / 1e6, unit="Mcells/s", cores=workers, seconds=round(wall, 3), object_cells=int(mask.sum()),
                sample="one %dx%d fp32 synth_dem(seed=%d), windows 1..%d, %d processes x row strips with r halo rows per "
                       "phase (erosion | dilation + flag), planes in shared memory" % (n, n, seed, windows, workers))


def single(n, windows, seed):
    from neilpy_amd.synth import synth_dem
    from oracle import smrf_oracle as orc
    Z = synth_dem(n, seed=seed, dtype=np.float32)
    t0 = time.perf_counter()
    m = orc.progressive_filter(Z, np.arange(1, windows + 1), 1, .15)
    wall = time.perf_counter() - t0
    return dict(value=n * n / wall / 1e6, unit="Mcells/s", cores=1, seconds=round(wall, 3), object_cells=int(m.sum()),
                sample="one %dx%d fp32 synth_dem(seed=%d), windows 1..%d, single process, single thread" % (n, n, seed, windows))


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
    ap.add_argument("--mode", default="tiles", choices=["tiles", "strips", "single"],
                    help="tiles: independent crop x crop tiles, one per core (bench.py's all_cores leg); strips: ONE "
                         "n x n raster split into row strips with halo rows; single: one raster, one thread")
    ap.add_argument("--n", type=int, default=2048, help="raster edge for --mode strips / single")
    a = ap.parse_args()
    workers = a.workers or usable_cores()
    if a.mode == "strips":
        print(json.dumps(strips(a.n, a.windows, workers, a.seed)))
        return
    if a.mode == "single":
        print(json.dumps(single(a.n, a.windows, a.seed)))
        return
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
