"""The benchmark's window list (1..50) and every ring radius against the reference / the oracle, under -m gpu.

* ``progressive_filter_w50_mid.npz`` / ``_big.npz``: the REFERENCE's own progressive_filter (neilpy/neilpy.py:1659-1680,
  imported by tests/golden/make_golden.py pf_w50) on fp32 synth_dem rasters of 768 x 1024 and 2048 x 2048 cells with
  ``windows = arange(1, 51)`` - the window list bench.py times at 16384^2.  1024 columns are 4 strips of 256 and both
  rasters are cut into several row segments per strip on the device, so the multi-strip / multi-segment / XCD-remapped
  execution the benchmark runs is compared with the reference bit for bit, under every routing of the small disks and
  through the row-band driver.
* every radius 1..64 of the ring kernels (erosion and dilation, fp32 and fp64) against the oracle.
"""
import hashlib

import numpy as np
import pytest

from conftest import golden, unpack
from sharded_one_gpu import run_bands

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nz(gpu_device):
    import neilpy_amd
    neilpy_amd.load_library()
    return neilpy_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import smrf_oracle
    return smrf_oracle


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def w50(which, nz):
    g = golden("progressive_filter_w50_%s.npz" % which)
    rows, cols = (int(v) for v in g["shape"])
    Z = nz.synth_dem(cols, seed=int(g["seed"]), dtype=np.float32, rows=rows)
    assert sha(Z) == str(g["Z_sha1"]), "synth_dem no longer generates the raster the golden was made from"
    return g, Z


@pytest.mark.parametrize("fused,chain,dual", [(None, None, None), ("0", None, None), ("2", None, None), ("2", "0", None),
                                              (None, "0", None), (None, None, "1"), ("0", None, "1")])
def test_w50_mid_reference_golden(nz, monkeypatch, fused, chain, dual):
    """default routing (chained small windows, fused, two-pass), two-pass everywhere (SMRF_FUSED=0), every chain / fused
    kernel that exists whatever the raster size (SMRF_FUSED=2), the same without chains (SMRF_CHAIN=0: every small window
    through its own fused launch); SMRF_RING_DUAL=1: the in-place ring instances the 16384^2 benchmark runs on its long
    segments (csrc/ring_inpl.inc, 17 radii) forced on this raster - erosion and dilation + flag step of each"""
    for name, val in (("SMRF_FUSED", fused), ("SMRF_CHAIN", chain), ("SMRF_RING_DUAL", dual)):
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)
    g, Z = w50("mid", nz)
    windows = g["windows"]
    assert list(windows) == list(range(1, 51))
    mask, wd = nz.progressive_filter(Z, windows, 1, .15, return_when_dropped=True)
    want = unpack(g["mask_bits"], Z.shape)
    assert int(mask.sum()) == int(g["object_cells"])
    assert np.array_equal(mask, want)
    assert np.array_equal(wd, g["when_dropped"])
    assert np.array_equal(nz.progressive_filter(Z, windows, 1, .15), want)
    last = Z
    for w in windows:
        last = nz.opening(last, nz.disk(int(w)))
    assert np.array_equal(last, g["opened_last"])             # the 50th opened surface, bit for bit


@pytest.mark.parametrize("which,world,chain", [("mid", 4, None), ("big", 8, None), ("mid", 4, "0")])
def test_w50_row_band_driver(nz, gpu_device, monkeypatch, which, world, chain):
    """the sharded driver's bands (4 x 192 rows, 8 x 256 rows) with all 50 windows against the reference's mask; the small
    windows of a group as chained / table-free launches (default) or one fused launch each (SMRF_CHAIN=0)"""
    import torch
    if chain is None:
        monkeypatch.delenv("SMRF_CHAIN", raising=False)
    else:
        monkeypatch.setenv("SMRF_CHAIN", chain)
    g, Z = w50(which, nz)
    Zd = torch.from_numpy(Z).to(gpu_device)
    mask, when, groups = run_bands(nz, Zd, g["windows"], world, return_when_dropped=True)
    assert len(groups) > 1
    assert np.array_equal(mask.cpu().numpy(), unpack(g["mask_bits"], Z.shape))
    assert sha(when.cpu().numpy()) == str(g["when_dropped_sha1"])


def test_w50_big_reference_golden(nz, gpu_device):
    """2048^2, windows 1..50 (the raster of the CPU baseline): mask, when_dropped and the last opened
    surface of the reference, single device, default routing"""
    import torch
    g, Z = w50("big", nz)
    Zd = torch.from_numpy(Z).to(gpu_device)
    mask, wd = nz.progressive_filter(Zd, g["windows"], 1, .15, return_when_dropped=True)
    m = mask.cpu().numpy()
    assert int(m.sum()) == int(g["object_cells"])
    assert np.array_equal(m, unpack(g["mask_bits"], Z.shape))
    w = wd.cpu().numpy()
    assert sha(w) == str(g["when_dropped_sha1"])
    assert np.array_equal(np.bincount(w[m].ravel(), minlength=50), g["when_dropped_hist"])
    last = Zd
    for r in g["windows"]:
        last = nz.opening(last, radius=int(r))
    assert sha(last.cpu().numpy()) == str(g["opened_last_sha1"])


def test_dual_ring_instances_vs_oracle(nz, orc, monkeypatch):
    """the radii with two ring instances (csrc/ring_inpl.inc: shifting ring for short segments, in-place 3- or 4-wave ring
    for long ones): both forced in turn on a small raster, against the oracle"""
    rng = np.random.default_rng(65)
    shape = (130, 520)
    Z = (rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.05 + 200 + (rng.random(shape) < 0.05) * rng.uniform(1, 25, shape)).astype(np.float32)
    for r in (21, 22, 23, 30, 33, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50):
        fp = orc.disk(r)
        we, wd = orc.erosion(Z, fp), orc.dilation(Z, fp)
        for mode in ("0", "1"):
            monkeypatch.setenv("SMRF_RING_DUAL", mode)
            assert np.array_equal(nz.erosion(Z, radius=r, impl=1), we), (r, mode, "erosion")
            assert np.array_equal(nz.dilation(Z, radius=r, impl=1), wd), (r, mode, "dilation")
    monkeypatch.delenv("SMRF_RING_DUAL")


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_radius_vs_oracle(nz, orc, dtype):
    """ring kernels, every instantiated radius, erosion and dilation, 3 strips wide (about 70 s of oracle per dtype)"""
    rng = np.random.default_rng(64)
    shape = (150, 600)
    base = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.05 + 200
    Z = (base + (rng.random(shape) < 0.05) * rng.uniform(1, 25, shape)).astype(dtype)
    for r in range(1, 65):
        fp = orc.disk(r)
        e = nz.erosion(Z, radius=r, impl=1)
        assert e.dtype == dtype and np.array_equal(e, orc.erosion(Z, fp)), (r, "erosion")
        d = nz.dilation(Z, radius=r, impl=1)
        assert np.array_equal(d, orc.dilation(Z, fp)), (r, "dilation")


def test_window_routes(nz, gpu_device, monkeypatch):
    """how progressive_filter runs each window (smrf_progressive_filter_timed_*'s route report): chained small windows,
    table-free single launches, fused openings, two ring passes - by radius and raster size; every route the same bits"""
    import torch
    from neilpy_amd import api, _lib
    Z = torch.from_numpy(nz.synth_dem(512, seed=3)).to(gpu_device)
    win = np.arange(1, 17)
    thr = .15 * (win * 1)
    C = _lib.ROUTE_CHAIN

    def run():
        t = {}
        m, _ = api._progressive_filter_device(Z, win, thr, False, nan_aware=0, timing=t)
        assert len(t["window_ms"]) == 16 and float(np.sum(t["window_ms"])) > 0
        return m, [int(v) for v in t["route"]]

    for name in ("SMRF_FUSED", "SMRF_CHAIN"):
        monkeypatch.delenv(name, raising=False)
    m0, r0 = run()                                          # a small raster: chains 1-3, singles 4..8, fused none above 8
    assert r0 == [C, C + 1, C + 2, C, C, C, C, C] + [_lib.ROUTE_TWO_PASS] * 8
    monkeypatch.setenv("SMRF_FUSED", "2")                   # every launch kind that exists, whatever the size
    m2, r2 = run()
    assert r2 == [C, C + 1, C + 2, C, C + 1, C, C, C, C, C] + [_lib.ROUTE_FUSED] * 4 + [_lib.ROUTE_TWO_PASS] * 2
    monkeypatch.setenv("SMRF_CHAIN", "0")
    m3, r3 = run()
    assert r3 == [_lib.ROUTE_FUSED] * 8 + [_lib.ROUTE_TWO_PASS] + [_lib.ROUTE_FUSED] * 5 + [_lib.ROUTE_TWO_PASS] * 2
    monkeypatch.setenv("SMRF_FUSED", "0")
    m4, r4 = run()
    assert r4 == [_lib.ROUTE_TWO_PASS] * 16
    assert torch.equal(m0, m2) and torch.equal(m0, m3) and torch.equal(m0, m4)
