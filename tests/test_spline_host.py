"""Host set-up of the bicubic spline (neilpy_amd/spline.py): the banded LU fast-forwards over the repeating interior rows of
an equally spaced axis - bit-identical to the row-by-row elimination."""
import numpy as np
import pytest

from neilpy_amd import spline


def plain_lu(bands):
    m = bands.shape[1]
    a = [bands[b].copy() for b in range(5)]
    l2, l1, d, u1, u2 = (np.zeros(m) for _ in range(5))
    for i in range(m):
        e, c, dg, f, g = a[0][i], a[1][i], a[2][i], a[3][i], a[4][i]
        if i >= 2:
            l2[i] = e / d[i - 2]
            c = c - l2[i] * u1[i - 2]
            dg = dg - l2[i] * u2[i - 2]
        if i >= 1:
            l1[i] = c / d[i - 1]
            dg = dg - l1[i] * u1[i - 1]
            f = f - l1[i] * u2[i - 1]
        d[i], u1[i], u2[i] = dg, f, g
    return np.stack([l2, l1, d, u1, u2])


@pytest.mark.parametrize("m", list(range(4, 80)) + [136, 304, 1000, 4971, 8193])
def test_banded_lu_equals_the_plain_elimination_on_pixel_centres(m):
    bands = spline.collocation_bands(np.arange(0.5, m + .5))
    assert np.array_equal(spline.banded_lu(bands), plain_lu(bands))


def test_banded_lu_on_unequal_sites():
    rng = np.random.default_rng(1)
    for m in (5, 17, 200):
        bands = spline.collocation_bands(np.sort(rng.random(m)) * m)
        assert np.array_equal(spline.banded_lu(bands), plain_lu(bands))


def test_lu_solves_the_collocation_system():
    m = 300
    x = np.arange(0.5, m + .5)
    bands = spline.collocation_bands(x)
    A = np.zeros((m, m))
    for b in range(5):
        for i in range(m):
            j = i + b - 2
            if 0 <= j < m:
                A[i, j] = bands[b, i]
    l2, l1, d, u1, u2 = spline.banded_lu(bands)
    L = np.eye(m) + np.diag(l1[1:], -1) + np.diag(l2[2:], -2)
    U = np.diag(d) + np.diag(u1[:-1], 1) + np.diag(u2[:-2], 2)
    assert np.abs(L @ U - A).max() < 1e-15
