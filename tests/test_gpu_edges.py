"""Edge cases of the drop-in API on the GPU, each against the oracle (tiny, ragged and degenerate inputs)."""
import numpy as np
import pytest

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
