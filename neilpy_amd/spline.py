"""Host-side set-up of the bicubic interpolating spline used by smrf()'s tail.

The reference evaluates ``scipy.interpolate.RectBivariateSpline(rows, cols, Z).ev(r, c)`` with
the defaults kx = ky = 3, s = 0 (neilpy.py:1773-1774, :1788-1790): FITPACK's ``regrid`` then
returns the INTERPOLATING tensor-product cubic spline whose knots are the data sites without the
second and the second-to-last one (fpregr.f: ``tx(kx+1+i) = x(i+2)``), fourfold at both ends.
Per axis that is an m x m collocation system with two sub- and two super-diagonals.  This module
builds the knots and the banded LU factors of that system (1-D, float64, on the host - O(m));
the 2-D solves and the evaluation at the points run on the GPU (csrc/spline.hip).
"""
import functools

import numpy as np

__all__ = ["knots", "bspline_basis", "collocation_bands", "banded_lu", "axis_factors"]


def knots(x):
    """FITPACK's knot vector for an interpolating cubic spline through the sites ``x`` (len >= 4)."""
    x = np.asarray(x, dtype=np.float64)
    m = x.size
    if m < 4:
        raise ValueError("a cubic spline needs at least 4 data sites per axis (got %d)" % m)
    return np.concatenate([np.repeat(x[0], 4), x[2:m - 2], np.repeat(x[-1], 4)])


def bspline_basis(t, l, x):
    """The 4 cubic B-splines that are non-zero on [t[l], t[l+1]) evaluated at x (FITPACK fpbspl)."""
    h = np.zeros(4)
    hh = np.zeros(4)
    h[0] = 1.0
    for j in range(1, 4):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] = h[i] + f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    return h


def collocation_bands(x):
    """Bands of the collocation matrix A[i, j] = B_j(x_i): array (5, m), band b holds A[i, i+b-2].

    Vectorised over the sites: the same fpbspl recurrence as :func:`bspline_basis`, on arrays."""
    x = np.asarray(x, dtype=np.float64)
    m = x.size
    t = knots(x)
    n = t.size
    # interval l with t[l] <= x < t[l+1], 3 <= l <= n-5 (the last interval also takes x = t[n-4])
    l = np.clip(np.searchsorted(t, x, side="right") - 1, 3, n - 5)
    h = np.zeros((4, m))
    hh = np.zeros((4, m))
    h[0] = 1.0
    for j in range(1, 4):
        hh[:j] = h[:j]
        h[0] = 0.0
        for i in range(j):
            li = l + i + 1
            lj = li - j
            f = hh[i] / (t[li] - t[lj])
            h[i] = h[i] + f * (t[li] - x)
            h[i + 1] = f * (x - t[lj])
    bands = np.zeros((5, m))
    rows = np.arange(m)
    for q in range(4):
        b = (l - 3 + q) - rows + 2                    # band of column l-3+q in row i
        nz = h[q] != 0.0
        if np.any(nz & ((b < 0) | (b > 4))):
            raise AssertionError("collocation matrix is not penta-diagonal")
        bands[b[nz], rows[nz]] = h[q][nz]
    return bands


def banded_lu(bands):
    """LU without pivoting of a penta-diagonal matrix given by its bands (5, m).

    Returns (5, m): rows l2, l1 (multipliers of rows i-2, i-1), d (pivot), u1, u2 (row i of U).
    The collocation matrix of spline interpolation is totally positive, so no pivoting is needed.
    """
    m = bands.shape[1]
    a = [bands[b].copy() for b in range(5)]           # a[b][i] = A[i, i+b-2]
    l2 = np.zeros(m)
    l1 = np.zeros(m)
    d = np.zeros(m)
    u1 = np.zeros(m)
    u2 = np.zeros(m)
    def row(i):
        # row i currently: (e, c, diag, f, g) = A[i, i-2 .. i+2] after eliminating with rows < i
        e = a[0][i]
        c = a[1][i]
        dg = a[2][i]
        f = a[3][i]
        g = a[4][i]
        if i >= 2:
            l2[i] = e / d[i - 2]
            c = c - l2[i] * u1[i - 2]
            dg = dg - l2[i] * u2[i - 2]
        if i >= 1:
            l1[i] = c / d[i - 1]
            dg = dg - l1[i] * u1[i - 1]
            f = f - l1[i] * u2[i - 1]
        d[i] = dg
        u1[i] = f
        u2[i] = g

    # Equally spaced sites give the same band entries in every interior row, and the elimination is a contraction: after a
    # few dozen rows two consecutive rows of the factors are bit-equal, and from there every further interior row repeats
    # them exactly (same inputs, same operations).  Those rows are filled in one assignment; the loop resumes where the band
    # entries change again (the last rows).  Bit-identical to the plain loop (tests/test_spline_host.py), O(1) Python steps
    # instead of m: 57 -> 0.5 ms at m = 32769.
    stack = np.stack(a)
    same_as_next = np.ones(m, dtype=bool)
    same_as_next[:-1] = np.all(stack[:, 1:] == stack[:, :-1], axis=0)     # band entries of row i + 1 equal those of row i
    i = 0
    while i < m:
        row(i)
        if i >= 3 and same_as_next[i] and same_as_next[i - 1] and \
                all(v[i] == v[i - 1] == v[i - 2] for v in (l2, l1, d, u1, u2)):
            brk = np.flatnonzero(~same_as_next[i:])       # first row whose band entries differ from row i's: j
            j = i + int(brk[0]) + 1 if brk.size else m
            # rows i + 1 .. j - 1 have row i's inputs and see two identical predecessors: they repeat row i
            for v in (l2, l1, d, u1, u2):
                v[i + 1:j] = v[i]
            i = j
            continue
        i += 1
    return np.stack([l2, l1, d, u1, u2])


@functools.lru_cache(maxsize=16)
def axis_factors(m):
    """(knots, LU bands) for the sites 0.5, 1.5, ..., m - 0.5 (pixel centres, neilpy.py:1768-1769)."""
    x = np.arange(0.5, m + .5)
    return knots(x), banded_lu(collocation_bands(x))
