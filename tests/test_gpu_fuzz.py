"""Seeded random point clouds through the whole smrf(): HIP path against the oracle (which the goldens
pin to the reference).  Varies point density, cell size (incl. non-integers, where the index
arithmetic of create_dem is delicate), window lists, thresholds and the low-outlier option."""
import numpy as np
import pytest

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


def cloud(rng, npts, extent):
    x = rng.uniform(1000.0, 1000.0 + extent[0], npts)
    y = rng.uniform(5000.0, 5000.0 + extent[1], npts)
    ground = 50 + 0.05 * (x - 1000) + 3 * np.sin((y - 5000) / 7.0)
    z = ground + np.where(rng.random(npts) < .25, rng.uniform(2, 15, npts), rng.normal(0, .05, npts))
    z[rng.random(npts) < .005] -= rng.uniform(5, 30)                       # low outliers
    return np.round(x, 2), np.round(y, 2), np.round(z, 2)


CASES = [
    # seed, points, extent, cellsize, windows, slope, elev_thr, scaler, low_fill
    (1, 3000, (60, 45), 1, 5, .15, .5, 1.25, False),
    (2, 800, (50, 70), 2, 3, .2, .3, 1.0, False),
    (3, 5000, (40, 40), .5, np.array([1, 2, 4, 7]), .15, .5, 1.25, True),
    (4, 2500, (33, 57), .3, 6, .1, .4, 2.0, False),
    (5, 1200, (90, 20), 1.5, np.array([3, 1, 2]), .3, .6, 0.0, False),
    (6, 6000, (70, 70), 1, 12, .15, .5, 1.25, True),
    (7, 400, (25, 25), 1, 2, .15, .5, 1.25, False),
    (8, 2000, (48, 52), .7, 4, .25, 1.0, .5, False),
]


@pytest.mark.parametrize("case", CASES, ids=[str(c[0]) for c in CASES])
def test_smrf_random_cloud(nz, orc, case):
    seed, npts, extent, cellsize, windows, slope, ethr, scaler, low_fill = case
    x, y, z = cloud(np.random.default_rng(seed), npts, extent)
    kw = dict(cellsize=cellsize, windows=windows, slope_threshold=slope, elevation_threshold=ethr,
              elevation_scaler=scaler, low_outlier_fill=low_fill)
    want = orc.smrf(x, y, z, **kw)
    got = nz.smrf(x, y, z, **kw)
    assert got[0].shape == want[0].shape and tuple(got[1])[:6] == tuple(want[1])[:6]
    assert np.array_equal(got[2], want[2])                                  # object cells, bit-exact
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-7)          # DTM
    assert np.array_equal(np.asarray(got[3]), np.asarray(want[3]))          # point flags


def test_return_extras_far_edge_index_error(nz, orc):
    """A point that rounds onto the raster's far edge makes the reference's when_dropped lookup raise
    (neilpy.py:1780, drop_raster[round(r), round(c)]); NumPy and CUDA inputs both raise IndexError here,
    and without return_extras the same cloud runs."""
    import torch
    x, y, z = cloud(np.random.default_rng(2), 800, (50, 70))
    kw = dict(cellsize=2, windows=3, slope_threshold=.2, elevation_threshold=.3, elevation_scaler=1.0)
    with pytest.raises(IndexError):
        orc.smrf(x, y, z, return_extras=True, **kw)
    with pytest.raises(IndexError):
        nz.smrf(x, y, z, return_extras=True, **kw)
    with pytest.raises(IndexError):
        nz.smrf(*(torch.from_numpy(v).cuda() for v in (x, y, z)), return_extras=True, **kw)
    assert len(nz.smrf(x, y, z, **kw)) == 4


@pytest.mark.parametrize("seed,shape,known", [(11, (37, 53), .5), (12, (90, 41), .1), (13, (64, 64), .03), (14, (5, 130), .3),
                                              (15, (120, 7), .6), (16, (77, 77), .9)])
def test_inpaint_random_holes(nz, orc, seed, shape, known):
    rng = np.random.default_rng(seed)
    A = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.1 + 100
    A[rng.random(shape) >= known] = np.nan
    want, istop, itn = orc.inpaint_nans_by_springs(A, return_info=True)
    got = nz.inpaint_nans_by_springs(A)
    st = nz.last_stats["inpaint"]
    assert (st["istop"], st["itn"]) == (istop, itn)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)


@pytest.mark.parametrize("seed,shape,dtype,windows", [(21, (70, 90), np.float32, [1, 2, 3, 5, 9]), (22, (33, 200), np.float64, [4, 2, 11]),
                                                      (23, (150, 64), np.float32, list(range(1, 15))), (24, (9, 9), np.float64, [1, 6])])
def test_progressive_filter_random(nz, orc, seed, shape, dtype, windows):
    rng = np.random.default_rng(seed)
    Z = (rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.05 + 200 + (rng.random(shape) < .06) * rng.uniform(1, 25, shape)).astype(dtype)
    win = np.asarray(windows)
    m, w = nz.progressive_filter(Z, win, .5, .2, return_when_dropped=True)
    m2, w2 = orc.progressive_filter(Z, win, .5, .2, return_when_dropped=True)
    assert np.array_equal(m, m2) and np.array_equal(w, w2)
