"""Edge cases of the drop-in API on the GPU, each against the oracle (tiny, ragged and degenerate inputs)."""
import numpy as np
import pytest

from conftest import switch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nz(gpu_device):
    import neilpy_amd
    return neilpy_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import smrf_oracle
    return smrf_oracle


@pytest.mark.parametrize("shape", [(1, 1), (1, 2), (2, 1), (2, 3), (3, 3), (1, 257), (259, 1), (4, 513)])
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_progressive_filter_tiny_rasters(nz, orc, shape, dtype):
    rng = np.random.default_rng(shape[0] * 1000 + shape[1])
    Z = (rng.normal(0, 2, shape) + 50).astype(dtype)
    lim = 4 * min(shape)                     # scipy's reflect table is only valid below this radius
    windows = np.array([w for w in (1, 2, 3) if w < lim] or [0])
    m, w = nz.progressive_filter(Z, windows, 1, .05, return_when_dropped=True)
    m2, w2 = orc.progressive_filter(Z, windows, 1, .05, return_when_dropped=True)
    assert m.shape == shape and np.array_equal(m, m2) and np.array_equal(w, w2)


def test_progressive_filter_empty_and_single_window(nz, orc):
    Z = np.zeros((0, 5), np.float32)
    assert nz.progressive_filter(Z, np.array([1, 2])).shape == (0, 5)
    rng = np.random.default_rng(3)
    Z = (rng.normal(0, 3, (40, 33)) + 10).astype(np.float32)
    for windows in (np.array([4]), np.array([], dtype=int), np.array([2, 2, 2])):
        assert np.array_equal(nz.progressive_filter(Z, windows, 2.5, .3), orc.progressive_filter(Z, windows, 2.5, .3))


def test_progressive_filter_integer_and_noncontiguous_input(nz, orc):
    rng = np.random.default_rng(4)
    Zi = rng.integers(0, 50, (30, 41)).astype(np.int32)
    assert np.array_equal(nz.progressive_filter(Zi, np.arange(1, 5)), orc.progressive_filter(Zi.astype(np.float64), np.arange(1, 5)))
    Zf = (rng.normal(0, 3, (60, 80)) + 10).astype(np.float64)
    view = Zf[::2, 5:60:3]                                     # strided view
    assert np.array_equal(nz.progressive_filter(view, np.arange(1, 4)), orc.progressive_filter(view, np.arange(1, 4)))
    assert np.array_equal(Zf[::2, 5:60:3], view)


def test_inf_values_and_negative_zero(nz, orc):
    rng = np.random.default_rng(5)
    Z = (rng.normal(0, 3, (25, 37)) + 10).astype(np.float32)
    Z[3, 4], Z[10, 20], Z[0, 0] = np.inf, -np.inf, -0.0
    for r in (1, 3, 6):
        assert np.array_equal(nz.erosion(Z, radius=r), orc.erosion(Z, orc.disk(r)))
        assert np.array_equal(nz.dilation(Z, radius=r), orc.dilation(Z, orc.disk(r)))


def test_create_dem_degenerate(nz, orc):
    for x, y, z in ((np.array([5.0]), np.array([7.0]), np.array([1.5])),
                    (np.array([5.0, 5.0, 5.0]), np.array([7.0, 7.0, 7.0]), np.array([3.0, np.nan, 1.0])),
                    (np.array([0.5, -0.5]), np.array([-0.5, 0.5]), np.array([np.inf, -np.inf]))):
        for bt in ("min", "max"):
            I, t = nz.create_dem(x, y, z, cellsize=1, bin_type=bt)
            I2, t2 = orc.create_dem(x, y, z, cellsize=1, bin_type=bt)
            assert np.array_equal(I, I2, equal_nan=True) and tuple(t)[:6] == tuple(t2)[:6]
    with pytest.raises(ValueError):
        nz.create_dem(np.array([]), np.array([]), np.array([]))


def test_create_dem_float32_and_series_inputs(nz, orc):
    import pandas as pd
    rng = np.random.default_rng(6)
    x = np.round(rng.uniform(0, 30, 500), 2)
    y = np.round(rng.uniform(0, 20, 500), 2)
    z = np.round(rng.normal(5, 1, 500), 2)
    df = pd.DataFrame(dict(x=x, y=y, z=z))
    I, t = nz.create_dem(df.x, df.y, df.z, 2, 'min')
    I2, t2 = orc.create_dem(x, y, z, 2, 'min')
    assert np.array_equal(I, I2, equal_nan=True)


def test_inpaint_tiny_and_inplace_tensor(nz, orc, gpu_device):
    import torch
    for A in (np.array([[np.nan]]), np.array([[1.0, np.nan]]), np.array([[np.nan], [2.0], [np.nan]]),
              np.array([[1.0, 2.0], [np.nan, 4.0]])):
        got = nz.inpaint_nans_by_springs(A)
        want = orc.inpaint_nans_by_springs(A)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-9)
    rng = np.random.default_rng(7)
    A = rng.normal(0, 1, (50, 60))
    A[rng.random(A.shape) < 0.5] = np.nan
    t = torch.from_numpy(A.copy()).to(gpu_device)
    assert nz.inpaint_nans_by_springs(t, inplace=True) is None
    np.testing.assert_allclose(t.cpu().numpy(), orc.inpaint_nans_by_springs(A), rtol=0, atol=1e-8)
    t2 = torch.from_numpy(A.copy()).to(gpu_device)
    out = nz.inpaint_nans_by_springs(t2)
    assert out is not t2 and torch.isnan(t2).any() and not torch.isnan(out).any()


def test_smrf_small_and_error_paths(nz, orc):
    x, y, z = nz.synth_points(3000, 40.0, seed=9)
    a = nz.smrf(x, y, z, cellsize=2, windows=3, return_extras=True)
    b = orc.smrf(x, y, z, cellsize=2, windows=3, return_extras=True)
    assert np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-7)
    for k in ("drop_raster", "when_dropped"):
        assert np.array_equal(a[4][k], b[4][k])
    np.testing.assert_allclose(a[4]["above_ground_height"], b[4]["above_ground_height"], rtol=0, atol=1e-7)
    with pytest.raises(ValueError):                     # fewer than 4 rows: the spline needs 4 (SciPy raises too)
        nz.smrf(np.array([0.0, 9.0]), np.array([0.0, 0.4]), np.array([1.0, 2.0]), cellsize=1)
    with pytest.raises(NotImplementedError):
        nz.create_dem(x, y, z, use_binned_statistic=True)


@pytest.mark.parametrize("with_nan", [False, True])
@pytest.mark.parametrize("world,shape,windows,budget", [(3, (330, 300), [1, 2, 3, 5, 4, 8], 24), (2, (200, 131), [2, 6, 1], None),
                                                        (4, (512, 260), [1, 2, 3, 4, 5, 6, 7, 8, 9, 10], 40)])
def test_sharded_driver_bands_on_one_gpu(nz, orc, gpu_device, world, shape, windows, budget, with_nan):
    """neilpy_amd.sharded's row-band driver with the HIP band operators, every rank run in turn on this GPU: the grouped
    halo exchange, the edge-first split of a group's last dilation (side stream) and - ``with_nan`` - scipy's NaN rule
    across band borders (the sharded path used to ignore NaNs).  A neighbour's message is replaced by the
    same rows of the surface entering the group, computed on the whole raster.  Masks and when_dropped bit-exact
    against the oracle."""
    import torch
    from neilpy_amd import sharded
    rng = np.random.default_rng(world * 100 + shape[0])
    Z = nz.synth_dem(shape[1], seed=9, rows=shape[0]).astype(np.float32)
    if with_nan:
        Z[rng.random(shape) < .02] = np.nan
        Z[shape[0] // world - 1, 5:40] = np.nan                       # a run of NaNs on a band's last row
    win = np.asarray(windows)
    thr = .15 * (win * 1)
    want_m, want_w = orc.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
    Zd = torch.from_numpy(Z).to(gpu_device)
    min_band = min(sharded.band_rows(shape[0], world, k)[1] - sharded.band_rows(shape[0], world, k)[0] for k in range(world))
    groups = sharded.window_groups([int(w) for w in win], min_band, budget)
    entering, last = [], Zd
    for grp in groups:
        entering.append(last)
        for i in grp:
            last = nz.opening(last, radius=int(win[i]))
    real = sharded._exchange
    try:
        for k in range(world):
            b0, b1 = sharded.band_rows(shape[0], world, k)
            calls = []

            def fake_exchange(dist, group, rank, world_size, send_up, recv_up, send_down, recv_down):
                src = entering[len(calls)]
                calls.append(send_up.shape[0])
                # what this rank sends must be what the whole-raster surface holds in those rows (NaNs compare equal)
                assert torch.equal(torch.nan_to_num(send_up, nan=-7.0), torch.nan_to_num(src[b0:b0 + send_up.shape[0]], nan=-7.0))
                assert torch.equal(torch.nan_to_num(send_down, nan=-7.0), torch.nan_to_num(src[b1 - send_down.shape[0]:b1], nan=-7.0))
                if recv_up is not None:
                    recv_up.copy_(src[b0 - recv_up.shape[0]:b0])
                if recv_down is not None:
                    recv_down.copy_(src[b1:b1 + recv_down.shape[0]])
            sharded._exchange = fake_exchange
            ops = sharded.HipBandOps()
            if with_nan:
                ops.has_nan = lambda band: True                           # the all-reduced flag of a real run
            mask, when = sharded.progressive_filter_sharded(Zd[b0:b1].contiguous(), shape[0], win, thr, rank=k, world_size=world,
                                                            ops=ops, return_when_dropped=True, halo_budget=budget,
                                                            overlap=(k % 2 == 0))       # both forms of the driver
            torch.cuda.synchronize()
            assert calls == [sum(2 * int(win[i]) for i in g) for g in groups]
            assert np.array_equal(mask.cpu().numpy().astype(bool), want_m[b0:b1]), (k, "mask")
            assert np.array_equal(when.cpu().numpy(), want_w[b0:b1]), (k, "when")
    finally:
        sharded._exchange = real


@pytest.mark.parametrize("nbands", [1, 3, 8, 64])
def test_point_bucket_kernels_route_every_point_to_its_band(nz, gpu_device, nbands):
    """smrf_points_band_count / _pack (the all-to-all gridding of sharded.create_dem_sharded): the packed runs are a
    permutation of the cloud, every point sits in the run of the band that owns floor(row), and binning run k
    into band k gives the rows of the single-device raster bit for bit."""
    import torch
    from neilpy_amd import sharded
    x, y, z = nz.synth_points(300_000, 700.0, seed=5)
    x[:1000] = np.round(x[:1000]) + .5                       # points exactly on cell edges
    y[1000:2000] = np.round(y[1000:2000]) - .5
    I, t = nz.create_dem(x, y, z, cellsize=1, bin_type="min")
    ny, nx = I.shape
    inv = tuple(~t)[:6]
    xd, yd, zd = (torch.from_numpy(v).to(gpu_device) for v in (x, y, z))
    ops = sharded.HipPointOps()
    counts, px, py, pz = ops.bucket(xd, yd, zd, inv, ny, nbands)
    counts = counts.cpu().numpy()
    assert counts.sum() == len(x)
    key = lambda a, b, c: np.sort(a * 1e6 + b * 1e-3 + c)    # noqa: E731
    assert np.array_equal(key(px.cpu().numpy(), py.cpu().numpy(), pz.cpu().numpy()), key(x, y, z))
    row = np.floor((px.cpu().numpy() * inv[3] + py.cpu().numpy() * inv[4]) + inv[5])      # the device's order of operations
    start = 0
    for k in range(nbands):
        b0, b1 = sharded.band_rows(ny, nbands, k)
        run = slice(start, start + int(counts[k]))
        assert np.all((row[run] >= b0) & (row[run] < b1)), k
        band, empty, n_out = ops.bin_band(px[run].contiguous(), py[run].contiguous(), pz[run].contiguous(), inv, (ny, nx),
                                          b0, b1 - b0, "min")
        assert n_out == 0 and np.array_equal(band.cpu().numpy(), I[b0:b1], equal_nan=True), k
        start += int(counts[k])


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("shape", [(1, 1), (1, 300), (300, 1), (5, 7), (33, 240), (64, 241), (100, 481), (257, 515), (700, 260)])
def test_fused_small_disk_opening_equals_two_pass_and_oracle(nz, orc, shape, dtype, monkeypatch):
    """windows 1..8 and 10..14 take the fused opening + flag kernel (morph_fused.h); SMRF_FUSED=0 forces the two-pass ring kernels.
    Same bits from both, and from the oracle where scipy's reflect table is valid (radius < 4 * min(shape)): strip
    seams every 256 - 2R columns, rasters narrower than a strip, shorter than a segment's 4R warm-up rows."""
    rng = np.random.default_rng(shape[0] * 7 + shape[1])
    Z = (nz.synth_dem(max(shape[1], 8), seed=4, rows=max(shape[0], 8))[:shape[0], :shape[1]] +
         rng.normal(0, .2, shape)).astype(dtype)
    windows = np.array([1, 2, 3, 4, 5, 6, 7, 8, 3, 1, 10, 11, 12, 13, 14, 9])
    switch(monkeypatch, "SMRF_FUSED", "2")                    # radii 10..14 too, which small rasters do not take by default
    m1, w1 = nz.progressive_filter(Z, windows, 1, .1, return_when_dropped=True)
    switch(monkeypatch, "SMRF_FUSED", "0")
    m0, w0 = nz.progressive_filter(Z, windows, 1, .1, return_when_dropped=True)
    switch(monkeypatch, "SMRF_FUSED", None)
    assert np.array_equal(m1, m0) and np.array_equal(w1, w0)
    if 14 < 4 * min(shape):
        m2, w2 = orc.progressive_filter(Z, windows, 1, .1, return_when_dropped=True)
        assert np.array_equal(m1, m2) and np.array_equal(w1, w2)
