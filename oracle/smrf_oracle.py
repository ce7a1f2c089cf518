"""CPU oracle for the SMRF hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy/SciPy restatement of the reference's algorithm for
create_dem -> inpaint_nans_by_springs -> progressive_filter -> smrf
(/root/reference/neilpy/neilpy.py:1110-1166, :1221-1271, :1659-1680, :1685-1808).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module, and only as the checker.  The product package
``neilpy_amd`` never imports it and has no CPU fallback.

Parity pin: every function here is checked in ``tests/test_oracle_golden.py``
against golden vectors produced by running the reference itself in the build
container (``tests/golden/make_golden.py``; numpy 2.2.6 / scipy 1.15.3 /
pandas 2.3.3, see ``tests/golden/meta.json``), including the samp12 figures the
reference's SMRF notebook prints (ipynb :902-905).

Third-party arithmetic the reference reaches on this path and how it is restated:

* ``skimage.morphology.disk/opening`` (not installed here; unpinned in the
  reference's setup.py:29) -> ``disk`` below and SciPy's
  ``ndimage.grey_erosion/grey_dilation(footprint=disk, mode='reflect')``, the
  primitive skimage itself dispatches to.
* ``rasterio.transform.from_origin`` / ``affine.Affine`` -> :class:`Affine`.
* ``pandas`` groupby min/max -> NaN-skipping ``np.minimum.at/np.maximum.at``.
* ``scipy.sparse.linalg.lsqr`` (SciPy 1.15.3 ``_isolve/lsqr.py:97-587``) ->
  :func:`lsqr_springs`, matrix-free on the raster's edge planes.
* ``scipy.interpolate.RectBivariateSpline`` -> used as is (FITPACK).
* ``inpaint_nans_by_fda`` (:1170-1216) -> :func:`fda_system` (own assembly of the
  weighted second-difference equations) + SciPy's ``lsqr`` itself.
"""
from math import sqrt

import numpy as np
import scipy.ndimage as ndi
from scipy import interpolate

EPS = np.finfo(np.float64).eps


# ----------------------------------------------------------------------------
# affine transform (call sites neilpy.py:1141-1142, :1772)
# ----------------------------------------------------------------------------
class Affine(tuple):
    """9-tuple (a, b, c, d, e, f, 0, 0, 1) with the ``affine`` package's arithmetic."""

    def __new__(cls, a, b, c, d, e, f):
        return tuple.__new__(cls, (a, b, c, d, e, f, 0.0, 0.0, 1.0))

    def __invert__(self):
        sa, sb, sc, sd, se, sf = self[:6]
        idet = 1.0 / (sa * se - sb * sd)
        ra, rb, rd, re = se * idet, -sb * idet, -sd * idet, sa * idet
        return Affine(ra, rb, -sc * ra - sf * rb, rd, re, -sc * rd - sf * re)

    def __mul__(self, other):
        sa, sb, sc, sd, se, sf = self[:6]
        if isinstance(other, Affine):
            oa, ob, oc, od, oe, of = other[:6]
            return Affine(sa * oa + sb * od, sa * ob + sb * oe, sa * oc + sb * of + sc,
                          sd * oa + se * od, sd * ob + se * oe, sd * oc + se * of + sf)
        vx, vy = other
        return (vx * sa + vy * sb + sc, vx * sd + vy * se + sf)


def from_origin(west, north, xsize, ysize):
    return Affine(1.0, 0.0, west, 0.0, 1.0, north) * Affine(xsize, 0.0, 0.0, 0.0, -ysize, 0.0)


# ----------------------------------------------------------------------------
# create_dem (neilpy.py:1110-1166)
# ----------------------------------------------------------------------------
def create_dem(x, y, z, cellsize=1, bin_type='max', inpaint=False, edges=None):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    if edges is None:                                               # :1120-1124
        xedges = np.arange(cellsize * np.floor(np.min(x) / cellsize) - .5 * cellsize,
                           cellsize * np.ceil(np.max(x) / cellsize) + 1.5 * cellsize, cellsize)
        yedges = np.arange(cellsize * np.ceil(np.max(y) / cellsize) + .5 * cellsize,
                           cellsize * np.floor(np.min(y) / cellsize) - 1.5 * cellsize, -cellsize)
    else:                                                           # :1125-1132
        xedges, yedges = edges[0], edges[1]
        out = (x < xedges[0]) | (x > xedges[-1]) | (y > yedges[0]) | (y < yedges[-1])
        x, y, z = x[~out], y[~out], z[~out]
        cellsize = np.abs(xedges[1] - xedges[0])
    nx, ny = len(xedges) - 1, len(yedges) - 1                       # :1134
    t = from_origin(xedges[0], yedges[0], cellsize, cellsize)       # :1141
    c, r = ~t * (x, y)                                              # :1142
    c, r = np.floor(c).astype(np.int64), np.floor(r).astype(np.int64)
    if bin_type not in ('max', 'min'):
        raise ValueError('This type not supported.')                # :1158
    if np.any((r < 0) | (r >= ny) | (c < 0) | (c >= nx)):
        raise ValueError('invalid entry in coordinates array')      # np.ravel_multi_index, :1151
    idx = r * nx + c
    ok = ~np.isnan(z)                                               # groupby min/max skip NaN
    acc = np.full(nx * ny, np.inf if bin_type == 'min' else -np.inf)
    hit = np.zeros(nx * ny, dtype=bool)
    (np.minimum if bin_type == 'min' else np.maximum).at(acc, idx[ok], z[ok])
    hit[idx[ok]] = True
    I = np.full(nx * ny, np.nan)
    I[hit] = acc[hit]
    I = I.reshape((ny, nx))
    if inpaint == True:  # noqa: E712  (same truthiness test as the reference, :1163)
        I = inpaint_nans_by_springs(I)
    return I, t


def edges_from_IT(Image, Transform):                                 # neilpy/neilpy.py:1095-1102
    """Cell edges of a raster from its transform: the pixel corners (j, 0) and (0, i) pushed through ``Transform``."""
    r, c = np.shape(Image)[0], np.shape(Image)[1]
    j, i = np.arange(c + 1), np.arange(r + 1)
    x_edges, _ = Transform * (j, np.zeros_like(j))
    _, y_edges = Transform * (np.zeros_like(i), i)
    return x_edges, y_edges


# ----------------------------------------------------------------------------
# inpaint_nans_by_springs (neilpy.py:1227-1271) with LSQR (scipy lsqr.py)
# ----------------------------------------------------------------------------
def _sym_ortho(a, b):                                               # scipy lsqr.py:62-94
    if b == 0:
        return np.sign(a), 0, abs(a)
    elif a == 0:
        return 0, np.sign(b), abs(b)
    elif abs(b) > abs(a):
        tau = a / b
        s = np.sign(b) / sqrt(1 + tau * tau)
        c = s * tau
        r = b / s
    else:
        tau = b / a
        c = np.sign(a) / sqrt(1 + tau * tau)
        s = c * tau
        r = a / c
    return c, s, r


def _norm2(*planes):
    """sqrt(dot(v, v)) over the concatenation of the planes (np.linalg.norm of a 1-D vector)."""
    s = 0.0
    for p in planes:
        q = p.ravel()
        s += float(np.dot(q, q))
    return sqrt(s)


def lsqr_springs(A, atol=1e-6, btol=1e-6, conlim=1e8, iter_lim=None):
    """Solve the reference's spring system for the NaN cells of ``A`` with LSQR.

    Unknowns = NaN cells; one spring (matrix row) per 4-neighbour grid edge with at
    least one NaN endpoint, ``+1`` at the lower flat index and ``-1`` at the higher
    (neilpy.py:1238-1260).  Edges are kept as two planes, ``h[i, j]`` joining
    (i, j)-(i, j+1) and ``v[i, j]`` joining (i, j)-(i+1, j); vectors over unknowns are
    full rasters that stay zero on known cells.  The recurrence, the order of the
    floating-point operations and the stopping rule follow scipy lsqr.py:324-555
    with damp=0.  Returns (filled copy of A, istop, itn).
    """
    A = np.asarray(A, dtype=np.float64)
    m, n = A.shape
    hole = np.isnan(A)
    nunk = int(hole.sum())
    B = A.copy()
    if nunk == 0:
        return B, 0, 0
    if iter_lim is None:
        iter_lim = 2 * nunk
    K = np.where(hole, 0.0, A)
    act_h = hole[:, :-1] | hole[:, 1:]
    act_v = hole[:-1, :] | hole[1:, :]

    def matvec(x):                       # S_nan @ x : x[lo] - x[hi] on active edges
        return (x[:, :-1] - x[:, 1:]) * act_h, (x[:-1, :] - x[1:, :]) * act_v

    def rmatvec(uh, uv):                 # S_nan.T @ u, accumulated in spring (row) order:
        y = np.zeros((m, n))             # up (-), left (-), right (+), down (+)
        y[1:, :] -= uv
        y[:, 1:] -= uh
        y[:, :-1] += uh
        y[:-1, :] += uv
        return y * hole

    # rhs = -S_known @ A_known (neilpy.py:1263)
    uh = (K[:, 1:] - K[:, :-1]) * act_h
    uv = (K[1:, :] - K[:-1, :]) * act_v

    itn = 0
    istop = 0
    ctol = 1 / conlim if conlim > 0 else 0
    anorm = 0
    ddnorm = 0
    xnorm = 0
    xxnorm = 0
    z = 0
    cs2 = -1
    sn2 = 0
    bnorm = _norm2(uh, uv)
    x = np.zeros((m, n))
    beta = bnorm
    if beta > 0:
        uh, uv = (1 / beta) * uh, (1 / beta) * uv
        v = rmatvec(uh, uv)
        alfa = _norm2(v)
    else:
        v = x.copy()
        alfa = 0
    if alfa > 0:
        v = (1 / alfa) * v
    w = v.copy()
    rhobar = alfa
    phibar = beta
    arnorm = alfa * beta
    if arnorm == 0:
        B[hole] = x[hole]
        return B, istop, itn

    while itn < iter_lim:
        itn = itn + 1
        ah, av = matvec(v)
        uh, uv = ah - alfa * uh, av - alfa * uv
        beta = _norm2(uh, uv)
        if beta > 0:
            uh, uv = (1 / beta) * uh, (1 / beta) * uv
            anorm = sqrt(anorm ** 2 + alfa ** 2 + beta ** 2)
            v = rmatvec(uh, uv) - beta * v
            alfa = _norm2(v)
            if alfa > 0:
                v = (1 / alfa) * v
        rhobar1 = rhobar
        cs, sn, rho = _sym_ortho(rhobar1, beta)
        theta = sn * alfa
        rhobar = -cs * alfa
        phi = cs * phibar
        phibar = sn * phibar
        tau = sn * phi
        t1 = phi / rho
        t2 = -theta / rho
        dk = (1 / rho) * w
        x = x + t1 * w
        w = v + t2 * w
        ddnorm = ddnorm + _norm2(dk) ** 2
        delta = sn2 * rho
        gambar = -cs2 * rho
        rhs = phi - delta * z
        zbar = rhs / gambar
        xnorm = sqrt(xxnorm + zbar ** 2)
        gamma = sqrt(gambar ** 2 + theta ** 2)
        cs2 = gambar / gamma
        sn2 = theta / gamma
        z = rhs / gamma
        xxnorm = xxnorm + z ** 2
        acond = anorm * sqrt(ddnorm)
        rnorm = sqrt(phibar ** 2)
        arnorm = alfa * abs(tau)
        test1 = rnorm / bnorm
        test2 = arnorm / (anorm * rnorm + EPS)
        test3 = 1 / (acond + EPS)
        t1 = test1 / (1 + anorm * xnorm / bnorm)
        rtol = btol + atol * anorm * xnorm / bnorm
        if itn >= iter_lim:
            istop = 7
        if 1 + test3 <= 1:
            istop = 6
        if 1 + test2 <= 1:
            istop = 5
        if 1 + t1 <= 1:
            istop = 4
        if test3 <= ctol:
            istop = 3
        if test2 <= atol:
            istop = 2
        if test1 <= rtol:
            istop = 1
        if istop != 0:
            break
    B[hole] = x[hole]
    return B, istop, itn


def inpaint_nans_by_springs(A, inplace=False, neighbors=4, return_info=False):
    B, istop, itn = lsqr_springs(A)
    if inplace:
        A[...] = B
        return (None, istop, itn) if return_info else None
    return (B, istop, itn) if return_info else B


# ----------------------------------------------------------------------------
# inpaint_nans_by_fda (neilpy.py:1170-1216): assembled system, SciPy's own LSQR
# ----------------------------------------------------------------------------
def fda_system(A, return_rows=False):
    """(a, b, nan_list) of the least-squares problem the reference hands to LSQR.

    One equation per raster cell: vertical [1, -2, 1] unless in the first/last row plus horizontal
    [1, -2, 1] unless in the first/last column (:1180-1194; no equation at the four corners).  The
    equations that touch a NaN cell are kept once per NaN entry (:1207-1209: ``nonzero()[0]`` of
    the NaN columns lists a row once per stored entry); b = -(known columns) @ known values (:1206).
    """
    from scipy import sparse
    m, n = A.shape
    if m < 2 or n < 2:
        raise ValueError("negative dimensions are not allowed")          # what np.ones(2*n*(m-2)) raises at :1190
    flat = np.arange(m * n, dtype=np.int64).reshape(m, n)
    inner_r, inner_c = flat[1:-1, :].ravel(), flat[:, 1:-1].ravel()
    rows = np.concatenate([inner_r, inner_r, inner_r, inner_c, inner_c, inner_c])
    cols = np.concatenate([inner_r - n, inner_r + n, inner_r, inner_c - 1, inner_c + 1, inner_c])
    vals = np.concatenate([np.ones(2 * inner_r.size), -2 * np.ones(inner_r.size),
                           np.ones(2 * inner_c.size), -2 * np.ones(inner_c.size)]).astype(np.int8)
    L = sparse.coo_matrix((vals, (rows, cols)), (m * n, m * n), dtype=np.int8).tocsr()   # duplicates summed
    nan = np.isnan(A).ravel()
    nan_list, known = np.flatnonzero(nan), np.flatnonzero(~nan)
    b = -L[:, known] * A.ravel()[known]
    Ln = L[:, nan_list]
    k = np.repeat(np.arange(m * n), np.diff(Ln.indptr))
    if return_rows:                      # also the raster cell (flat index) every kept equation belongs to
        return Ln[k], b[k], nan_list, k
    return Ln[k], b[k], nan_list


def inpaint_nans_by_fda(A, fast=True, inplace=False, return_info=False):
    from scipy.sparse.linalg import lsqr
    a, b, nan_list = fda_system(A)
    res = lsqr(a, b)
    B = A if inplace else A.copy()
    B.ravel()[nan_list] = res[0]
    if inplace:
        return (None, res[1], res[2]) if return_info else None
    return (B, res[1], res[2]) if return_info else B


# ----------------------------------------------------------------------------
# disk / opening / progressive_filter (neilpy.py:1659-1680)
# ----------------------------------------------------------------------------
def disk(radius, dtype=np.uint8):
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)


def erosion(image, footprint):
    return ndi.grey_erosion(image, footprint=np.asarray(footprint), mode='reflect')


def dilation(image, footprint):
    fp = np.asarray(footprint)
    return ndi.grey_dilation(image, footprint=fp[::-1, ::-1], mode='reflect')


def opening(image, footprint):
    return dilation(erosion(image, footprint), footprint)


def progressive_filter(Z, windows, cellsize=1, slope_threshold=.15, return_when_dropped=False):
    last_surface = Z.copy()
    elevation_thresholds = slope_threshold * (windows * cellsize)
    is_object_cell = np.zeros(np.shape(Z), dtype=bool)
    when_dropped = np.zeros(np.shape(Z), dtype=np.uint8)
    for i, window in enumerate(windows):
        this_surface = opening(last_surface, disk(window))          # disk(window) always (:1667-1670)
        new_obj = last_surface - this_surface > elevation_thresholds[i]
        is_object_cell = is_object_cell | new_obj
        when_dropped[new_obj] = i
        if len(windows) > 1:
            last_surface = this_surface
    if return_when_dropped:
        return is_object_cell, when_dropped
    return is_object_cell


# ----------------------------------------------------------------------------
# smrf (neilpy.py:1685-1808)
# ----------------------------------------------------------------------------
def smrf(x, y, z, cellsize=1, windows=5, slope_threshold=.15, elevation_threshold=.5,
         elevation_scaler=1.25, low_filter_slope=5, low_outlier_fill=False, return_extras=False,
         return_stages=False):
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    z = np.asarray(z, dtype=np.float64)
    stages = {}
    if np.isscalar(windows):
        windows = np.arange(windows) + 1
    Zmin, t = create_dem(x, y, z, cellsize=cellsize, bin_type='min')
    stages['Zmin'] = Zmin.copy()
    is_empty_cell = np.isnan(Zmin)
    Zmin, istop1, itn1 = inpaint_nans_by_springs(Zmin, return_info=True)
    stages['inpaint1'] = Zmin.copy()
    stages['lsqr1'] = (istop1, itn1)
    low_outliers = progressive_filter(-Zmin, np.array([1]), cellsize, slope_threshold=low_filter_slope)
    stages['low_outliers'] = low_outliers
    if low_outlier_fill:
        Zmin[low_outliers] = np.nan
        Zmin = inpaint_nans_by_springs(Zmin)
        stages['inpaint1b'] = Zmin.copy()
    object_cells, drop_raster = progressive_filter(Zmin, windows, cellsize, slope_threshold,
                                                   return_when_dropped=True)
    stages['pf_mask'] = object_cells
    stages['pf_when_dropped'] = drop_raster
    Zpro = Zmin
    object_cells = is_empty_cell | low_outliers | object_cells
    Zpro[object_cells] = np.nan
    Zpro, istop2, itn2 = inpaint_nans_by_springs(Zpro, return_info=True)
    stages['lsqr2'] = (istop2, itn2)
    col_centers = np.arange(0.5, Zpro.shape[1] + .5)
    row_centers = np.arange(0.5, Zpro.shape[0] + .5)
    c, r = ~t * (x, y)
    f1 = interpolate.RectBivariateSpline(row_centers, col_centers, Zpro)
    elevation_values = f1.ev(r, c)
    if return_extras:                                                   # :1777-1780 (an IndexError there is the reference's too)
        when_dropped = drop_raster[np.round(r).astype(int), np.round(c).astype(int)]
    gy, gx = np.gradient(Zpro, cellsize)
    S = np.sqrt(gy ** 2 + gx ** 2)
    f2 = interpolate.RectBivariateSpline(row_centers, col_centers, S)
    slope_values = f2.ev(r, c)
    stages['elevation_values'] = elevation_values
    stages['slope_values'] = slope_values
    stages['slope'] = S
    required_value = elevation_threshold + (elevation_scaler * slope_values)
    is_object_point = np.abs(elevation_values - z) > required_value
    out = (Zpro, t, object_cells, is_object_point)
    if return_extras:
        out = out + (dict(above_ground_height=z - elevation_values, drop_raster=drop_raster,
                          when_dropped=when_dropped),)
    if return_stages:
        out = out + (stages,)
    return out


def pssm_classes(Z, cellsize=1, ve=2.3):                                # neilpy/neilpy.py:846-858
    """uint8 slope classes of pssm(): gradient, slope, vertical exaggeration, degrees / 90, x255, round."""
    gy, gx = np.gradient(Z, cellsize)
    S = np.sqrt(gx ** 2 + gy ** 2)
    P = np.rad2deg(np.arctan(ve * S)) / 90
    return np.round(255 * P).astype(np.uint8)


def pssm(Z, lut, cellsize=1, ve=2.3):                                   # neilpy/neilpy.py:860-865
    """``lut``: the 256 x 4 table of the colormap (matplotlib indexes it with the integer image)."""
    return np.asarray(lut)[pssm_classes(Z, cellsize, ve)]


def worldfile_lines(t):                                                 # neilpy/neilpy.py:1564-1570
    x_ul, y_ul = t * (.5, .5)
    return ["%0.10f" % v for v in (t[0], t[3], t[1], t[4], x_ul, y_ul)]
