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

from conftest import golden, switch, unpack
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


@pytest.mark.parametrize("fused,chain,dual,nt", [(None, None, None, None), ("0", None, None, None), ("2", None, None, None),
                                                 ("2", "0", None, None), (None, "0", None, None), (None, None, "1", None),
                                                 ("0", None, "1", None),
                                                 (None, None, None, "1"), ("0", None, None, "1"), ("2", None, None, "1"),
                                                 ("2", "0", None, "1"), ("0", None, "1", "1"), (None, None, "1", "1")])
def test_w50_mid_reference_golden(nz, monkeypatch, fused, chain, dual, nt):
    """default routing (chained small windows, fused, two-pass), two-pass everywhere (SMRF_FUSED=0), every chain / fused
    kernel that exists whatever the raster size (SMRF_FUSED=2), the same without chains (SMRF_CHAIN=0: every small window
    through its own fused launch); SMRF_RING_DUAL=1: the in-place ring instances the 16384^2 benchmark runs on its long
    segments (csrc/ring_inpl.inc, 21 radii) forced on this raster - erosion and dilation + flag step of each;
    SMRF_NT=1: the streaming-store epilogues of the ring, fused and chained kernels, which the library only takes by
    itself on planes of 192 MiB and up (morph.hip nt_rule: the 16384^2 benchmark), forced on this raster under each routing"""
    for name, val in (("SMRF_FUSED", fused), ("SMRF_CHAIN", chain), ("SMRF_RING_DUAL", dual), ("SMRF_NT", nt)):
        if val is None:
            switch(monkeypatch, name, None)
        else:
            switch(monkeypatch, name, val)
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
        switch(monkeypatch, "SMRF_CHAIN", None)
    else:
        switch(monkeypatch, "SMRF_CHAIN", chain)
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
    for r in (21, 22, 23, 24, 25, 26, 27, 30, 33, 39, 40, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50):
        fp = orc.disk(r)
        we, wd = orc.erosion(Z, fp), orc.dilation(Z, fp)
        for mode in ("0", "1"):
            switch(monkeypatch, "SMRF_RING_DUAL", mode)
            assert np.array_equal(nz.erosion(Z, radius=r, impl=1), we), (r, mode, "erosion")
            assert np.array_equal(nz.dilation(Z, radius=r, impl=1), wd), (r, mode, "dilation")
    switch(monkeypatch, "SMRF_RING_DUAL", None)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_every_radius_vs_oracle(nz, orc, dtype, monkeypatch):
    """ring kernels, every instantiated radius, erosion and dilation, 3 strips wide (about 70 s of oracle per dtype);
    each with ordinary and with streaming stores (SMRF_NT=0 / 1: the epilogue the 16384^2 benchmark takes)"""
    rng = np.random.default_rng(64)
    shape = (150, 600)
    base = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.05 + 200
    Z = (base + (rng.random(shape) < 0.05) * rng.uniform(1, 25, shape)).astype(dtype)
    for r in range(1, 65):
        fp = orc.disk(r)
        we, wd = orc.erosion(Z, fp), orc.dilation(Z, fp)
        for nt in ("0", "1"):
            switch(monkeypatch, "SMRF_NT", nt)
            e = nz.erosion(Z, radius=r, impl=1)
            assert e.dtype == dtype and np.array_equal(e, we), (r, "erosion", nt)
            d = nz.dilation(Z, radius=r, impl=1)
            assert np.array_equal(d, wd), (r, "dilation", nt)
    switch(monkeypatch, "SMRF_NT", None)


def test_buffer_segment_clamp_vs_oracle(nz, orc, gpu_device, monkeypatch):
    """The buffer-addressed ring kernels (csrc/ring_buf.inc) carry a workgroup's row offsets as 32-bit scalars the hardware
    does not range-check, so ring_launch_np clamps a segment's span below 2 GiB (morph_ring.h, "buffer addressing: a
    workgroup's row offsets are 32-bit").  On the benchmark rasters that clamp only engages at cfg5 size (32769 columns of
    fp64 / 16384^2 never), where no oracle reaches.  Here it is forced on a raster the oracle computes in a second: 150 x 600
    cells held with a row pitch of 8 MiB (ld = 2^21 floats through the C ABI), one segment asked for (SMRF_RING_SEG = all
    rows), so the unclamped span would be 150 rows = 1.2 GiB + warm-up and the clamped segments run with byte offsets up to
    ~1.9 GiB; erosion, dilation + flag step, with ordinary and streaming stores."""
    import ctypes as C
    import torch
    from neilpy_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(66)
    rows, cols, ld = 150, 600, 1 << 21
    base = rng.normal(0, 1, (rows, cols)).cumsum(0).cumsum(1) * 0.05 + 200
    Zh = (base + (rng.random((rows, cols)) < 0.05) * rng.uniform(1, 25, (rows, cols))).astype(np.float32)
    big = lambda dt: torch.zeros((rows, ld), dtype=dt, device=gpu_device)
    Zd, Ed, Od = big(torch.float32), big(torch.float32), big(torch.float32)
    Md, Wd = big(torch.uint8), big(torch.uint8)
    Zd[:, :cols] = torch.from_numpy(Zh).to(gpu_device)
    switch(monkeypatch, "SMRF_RING_SEG", str(rows))
    thr = 0.6
    for r in (15, 21, 29, 38, 49):                            # radii with buffer addressing on (ring_buf.inc), both ring forms
        fp = orc.disk(r)
        we = orc.erosion(Zh, fp)
        wo = orc.dilation(we, fp)
        want = (Zh - wo).astype(np.float64) > thr
        for nt in ("0", "1"):
            switch(monkeypatch, "SMRF_NT", nt)
            Ed.zero_(); Od.zero_(); Md.zero_(); Wd.zero_()
            _lib.check(lib.smrf_disk_filter_f32(C.c_void_p(Zd.data_ptr()), C.c_void_p(Ed.data_ptr()), rows, cols, ld, 0, rows,
                                                0, rows, r, 0, 0, 1, st))
            assert np.array_equal(Ed[:, :cols].cpu().numpy(), we), (r, nt, "erosion")
            _lib.check(lib.smrf_pf_dilate_flag_f32(C.c_void_p(Ed.data_ptr()), C.c_void_p(Zd.data_ptr()), C.c_void_p(Od.data_ptr()),
                                                   C.c_void_p(Md.data_ptr()), C.c_void_p(Wd.data_ptr()), thr, 7, rows, cols, ld,
                                                   0, rows, 0, rows, r, 0, 1, st))
            assert np.array_equal(Od[:, :cols].cpu().numpy(), wo), (r, nt, "dilation")
            assert np.array_equal(Md[:, :cols].cpu().numpy().astype(bool), want), (r, nt, "mask")
            assert np.array_equal(Wd[:, :cols].cpu().numpy(), want.astype(np.uint8) * 7), (r, nt, "when")
            assert int(Ed[:, cols:].abs().sum()) == 0 and int(Md[:, cols:].sum()) == 0   # nothing written beyond the raster's columns


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
        switch(monkeypatch, name, None)
    m0, r0 = run()                            # a small raster: chains 1-3, singles 4..9 (9: any size since round 5), fused none above 8
    assert r0 == [C, C + 1, C + 2, C, C, C, C, C, C] + [_lib.ROUTE_TWO_PASS] * 7
    switch(monkeypatch, "SMRF_FUSED", "2")                   # every launch kind that exists, whatever the size
    m2, r2 = run()
    assert r2 == [C, C + 1, C + 2, C, C + 1, C, C, C, C, C] + [_lib.ROUTE_FUSED] * 4 + [_lib.ROUTE_TWO_PASS] * 2
    switch(monkeypatch, "SMRF_CHAIN", "0")
    m3, r3 = run()
    assert r3 == [_lib.ROUTE_FUSED] * 8 + [_lib.ROUTE_TWO_PASS] + [_lib.ROUTE_FUSED] * 5 + [_lib.ROUTE_TWO_PASS] * 2
    switch(monkeypatch, "SMRF_FUSED", "0")
    m4, r4 = run()
    assert r4 == [_lib.ROUTE_TWO_PASS] * 16
    assert torch.equal(m0, m2) and torch.equal(m0, m3) and torch.equal(m0, m4)


def test_window_routes_f64(nz, orc, gpu_device, monkeypatch):
    """fp64 routing (round 4: table-free single launches at R = 4, 5, 7, 8 with their neighbour reads taken in groups,
    csrc/morph_chain.h chain_stage_grouped, and the chain 1, 2, 3 on large rasters): which launch every window takes, all routes
    the same bits, and the oracle"""
    import torch
    from neilpy_amd import api, _lib
    Zh = nz.synth_dem(384, seed=5, dtype=np.float64)
    Z = torch.from_numpy(Zh).to(gpu_device)
    win = np.arange(1, 11)
    thr = .15 * (win * 1)
    C = _lib.ROUTE_CHAIN

    def run():
        t = {}
        m, w = api._progressive_filter_device(Z, win, thr, True, nan_aware=0, timing=t)
        return m, w, [int(v) for v in t["route"]]

    for name in ("SMRF_FUSED", "SMRF_CHAIN"):
        switch(monkeypatch, name, None)
    m0, w0, r0 = run()     # a small raster (round 5's thresholds): chain 1, 2, 3; singles 4, 5; fused 6; two passes from 7 (single 7 from 4 Mi cells)
    assert r0 == [C, C + 1, C + 2, C, C, _lib.ROUTE_FUSED] + [_lib.ROUTE_TWO_PASS] * 4
    switch(monkeypatch, "SMRF_FUSED", "2")                  # every launch kind that exists, whatever the size
    m2, w2, r2 = run()
    assert r2 == [C, C + 1, C + 2, C, C, _lib.ROUTE_FUSED, C, C] + [_lib.ROUTE_TWO_PASS] * 2   # chain 1, 2, 3; singles 4, 5, 7, 8
    switch(monkeypatch, "SMRF_FUSED", "0")
    m4, w4, r4 = run()
    assert r4 == [_lib.ROUTE_TWO_PASS] * 10
    assert torch.equal(m0, m2) and torch.equal(m0, m4) and torch.equal(w0, w2) and torch.equal(w0, w4)
    want, want_w = orc.progressive_filter(Zh, win, 1, .15, return_when_dropped=True)
    assert np.array_equal(m2.cpu().numpy().astype(bool), want) and np.array_equal(w2.cpu().numpy(), want_w)


def test_unequal_segments_give_the_same_bits(nz, gpu_device, monkeypatch):
    """round 5 (morph_ring.h ring_launch_np, profiles/r05_segment_balance.md): a one-round ring launch cuts the rows into
    segments whose length depends on the residency class of their workgroups (SMRF_RING_SLOPE, permille of the mean length
    per class; built in: 60 for fp32 rasters whose strip count divides 256).  Placement only: erosion, dilation and the
    window step of a 6000 x 4096 raster (16 strips: the built-in rule applies, 2..8 classes per radius) and of a
    6000 x 4100 one (17 strips: the classes do not fall on whole rows of segments; only an explicit slope cuts it
    unequally) are the same bits with equal segments (0), the built-in rule, 60 and an exaggerated slope whose lengths
    are mostly rounding."""
    import torch
    for cols in (4096, 4100):
        Z = torch.from_numpy(nz.synth_dem(cols, seed=31, rows=6000)).to(gpu_device)
        Zd = Z[:3000].double()
        radii = (2, 5, 13, 18, 25, 39, 50, 64)
        want = {}
        for slope in (0, None, 60, 300):
            switch(monkeypatch, "SMRF_RING_SLOPE", slope)
            for r in radii:
                got = [nz.erosion(Z, radius=r, impl=1), nz.dilation(Z, radius=r, impl=1)]
                if r in (5, 18, 39):
                    got += [nz.erosion(Zd, radius=r, impl=1), nz.dilation(Zd, radius=r, impl=1)]
                if slope == 0:
                    want[r] = got
                else:
                    assert all(torch.equal(g, w) for g, w in zip(got, want[r])), (cols, slope, r)
            win = np.array([3, 12, 20, 33, 47])
            m = nz.progressive_filter(Z, win, 1, .15)
            if slope == 0:
                want["pf"] = m
                switch(monkeypatch, "SMRF_RING_ROUNDS", 2)       # (more than one round: equal segments whatever the slope)
                assert torch.equal(nz.erosion(Z, radius=18, impl=1), want[18][0])
                switch(monkeypatch, "SMRF_RING_ROUNDS", None)
            else:
                assert torch.equal(m, want["pf"]), (cols, slope)
        for r in (5, 18):                                         # and the equal-segment result is the direct kernel's
            switch(monkeypatch, "SMRF_RING_SLOPE", 0)
            assert torch.equal(nz.erosion(Z[:1500], radius=r, impl=2), nz.erosion(Z[:1500], radius=r, impl=1))
        # the row-band form of the C ABI (halo rows given, out_row0 > 0): the classes start at the band's first output row
        import ctypes as C
        from neilpy_amd import _lib
        lib = _lib.load()
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        b0, b1 = 1000, 5000
        for r in (18, 39):
            lo, hi = b0 - r, b1 + r
            src = Z[lo:hi]
            for slope in (None, 300):
                switch(monkeypatch, "SMRF_RING_SLOPE", slope)
                out = torch.empty((b1 - b0, cols), dtype=Z.dtype, device=gpu_device)
                _lib.check(lib.smrf_disk_filter_f32(C.c_void_p(src.data_ptr()), C.c_void_p(out.data_ptr()), 6000, cols, cols, lo,
                                                    hi - lo, b0, b1 - b0, r, 0, 0, _lib.IMPL_RING, st))
                assert torch.equal(out, want[r][0][b0:b1]), (cols, r, slope)


def test_segment_count_rounds_down(nz, gpu_device, monkeypatch):
    """round 5: a launch sized for one round of resident workgroups holds floor(slots / strips) rows of segments, not the
    nearest count - 33 strips x 16 segments = 528 workgroups on 512 slots ran 16 of them in a second round that lasted
    as long again (8193^2 fp32, all windows: 25.7 -> 21.8 ms; fp64: 99.9 -> 64.5 ms); and a raster too small to give
    every slot a long segment is cut so that its workgroups fill the CUs k times exactly (smrf_pick_nseg,
    seg_rule.h).  SMRF_SEG_RULE=1 is rounds 1-4's rule, 2 the rounded-down full round with the 4R minimum: the same
    bits under all three, on a raster whose strip count (33) makes them differ at most radii and on a small one."""
    import torch
    Z = torch.from_numpy(nz.synth_dem(8193, seed=32, rows=3000)).to(gpu_device)
    win = np.array([1, 2, 3, 6, 9, 12, 14, 18, 33, 50])
    want = nz.progressive_filter(Z, win, 1, .15)
    e = {r: nz.erosion(Z, radius=r, impl=1) for r in (7, 18, 50)}
    Zs = Z[:700, :1500].contiguous()
    want_s = nz.progressive_filter(Zs, win, 1, .15)
    for rule in (1, 2):
        switch(monkeypatch, "SMRF_SEG_RULE", rule)
        assert torch.equal(nz.progressive_filter(Z, win, 1, .15), want), rule
        assert torch.equal(nz.progressive_filter(Zs, win, 1, .15), want_s), rule
        for r, w in e.items():
            assert torch.equal(nz.erosion(Z, radius=r, impl=1), w), (rule, r)
