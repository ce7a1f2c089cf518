"""Host side of the SMRF path: neilpy's call signatures over libsmrf_hip (MI355X only).

Mirrors, argument for argument, the reference functions (paths relative to the reference
checkout): ``create_dem`` (neilpy/neilpy.py:1110), ``inpaint_nans_by_springs`` (:1227),
``inpaint_nans_by_fda`` (:1170), ``progressive_filter`` (:1659), ``smrf`` (:1685), the skimage seam they use, ``disk`` /
``opening`` (:43-44, :1670), and the notebooks' shading step ``pssm`` (:846).  Build-specific options are keyword-only and come last.

NumPy in -> NumPy out; ``torch`` CUDA tensor in -> CUDA tensor out (no host copy).  All compute
runs in hand-written HIP kernels behind the C ABI of ``include/smrf_hip.h``; PyTorch only owns
device memory and the stream.  There is no CPU fallback: without the library or a GPU every
function raises :class:`neilpy_amd.SmrfHipError`.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._device import device_scoped as _device_scoped, is_tensor as _is_tensor
from ._xfer import to_device as _h2d, to_host as _d2h
from .affine import from_origin

__all__ = ["disk", "erosion", "dilation", "opening", "progressive_filter", "create_dem",
           "inpaint_nans_by_springs", "inpaint_nans_by_fda", "smrf", "pssm", "last_stats"]

#: statistics of the most recent calls (LSQR istop / itn, unknown counts), SURVEY section 5
last_stats = {}


def _torch():
    import torch
    return torch


def _stream():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _to_device(a, dtype=None):
    """NumPy / tensor -> contiguous CUDA tensor (float32 and float64 kept, others -> float64)."""
    torch = _torch()
    _lib.require_gpu()
    if _is_tensor(a):
        t = a
    else:
        arr = np.asarray(a)
        if dtype is None and arr.dtype not in (np.float32, np.float64):
            arr = arr.astype(np.float64)
        t = _h2d(arr)                                      # pinned staging for large arrays
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    if t.dtype not in (torch.float32, torch.float64):
        t = t.to(torch.float64)
    if not t.is_cuda:
        t = t.cuda()
    return t.contiguous()


def _suffix(t):
    return "f32" if t.dtype == _torch().float32 else "f64"


# ------------------------------------------------------------------------------------------
# disk / erosion / dilation / opening  (skimage.morphology as used at neilpy.py:1667-1670)
# ------------------------------------------------------------------------------------------
def disk(radius, dtype=np.uint8):
    """skimage.morphology.disk: ``x*x + y*y <= radius*radius`` on a (2r+1)^2 grid."""
    L = np.arange(-radius, radius + 1)
    X, Y = np.meshgrid(L, L)
    return np.array((X ** 2 + Y ** 2) <= radius ** 2, dtype=dtype)


def _radius_of(footprint, radius):
    if radius is not None:
        return int(radius)
    fp = np.asarray(footprint)
    if fp.ndim != 2 or fp.shape[0] != fp.shape[1] or fp.shape[0] % 2 != 1:
        raise NotImplementedError("only skimage disk(r) footprints are supported on the device")
    r = fp.shape[0] // 2
    if not np.array_equal(fp != 0, disk(r) != 0):
        raise NotImplementedError("only skimage disk(r) footprints are supported on the device")
    return r


def _has_nan(t):
    lib = _lib.load()
    cnt = C.c_int64(0)
    _lib.check(getattr(lib, "smrf_count_nan_" + _suffix(t))(_ptr(t), t.numel(), C.byref(cnt), _stream()))
    return cnt.value > 0


def _disk_filter(image, radius, dilate, impl, nan_aware=None):
    torch = _torch()
    was_tensor = _is_tensor(image)
    src = _to_device(image)
    if src.dim() != 2:
        raise ValueError("expected a 2-D raster")
    rows, cols = src.shape
    if rows == 0 or cols == 0:
        out = src.clone()
        return out if was_tensor else _d2h(out)
    if nan_aware is None:
        nan_aware = _has_nan(src)
    out = torch.empty_like(src)
    lib = _lib.load()
    fn = getattr(lib, "smrf_disk_filter_" + _suffix(src))
    _lib.check(fn(_ptr(src), _ptr(out), rows, cols, cols, 0, rows, 0, rows, int(radius), int(bool(dilate)),
                  int(bool(nan_aware)), int(impl), _stream()))
    return out if was_tensor else _d2h(out)


@_device_scoped
def erosion(image, footprint=None, *, radius=None, impl=_lib.IMPL_AUTO):
    """Grey erosion by ``disk(r)``, borders ``mode='reflect'`` (scipy.ndimage.grey_erosion)."""
    return _disk_filter(image, _radius_of(footprint, radius), False, impl)


@_device_scoped
def dilation(image, footprint=None, *, radius=None, impl=_lib.IMPL_AUTO):
    """Grey dilation by ``disk(r)``, borders ``mode='reflect'`` (scipy.ndimage.grey_dilation)."""
    return _disk_filter(image, _radius_of(footprint, radius), True, impl)


@_device_scoped
def opening(image, footprint=None, *, radius=None, impl=_lib.IMPL_AUTO):
    """skimage.morphology.opening(image, disk(r)) = dilation(erosion(image))."""
    r = _radius_of(footprint, radius)
    was_tensor = _is_tensor(image)
    src = _to_device(image)
    nan_aware = _has_nan(src) if src.numel() else False
    e = _disk_filter(src, r, False, impl, nan_aware)
    o = _disk_filter(e, r, True, impl, nan_aware)
    return o if was_tensor else _d2h(o)


# ------------------------------------------------------------------------------------------
# progressive_filter  (neilpy.py:1659-1680)
# ------------------------------------------------------------------------------------------
def _progressive_filter_device(Zd, windows, thresholds, want_when, impl=_lib.IMPL_AUTO, nan_aware=-1, timing=None):
    """Device-resident core: CUDA raster in, (uint8 mask, uint8 when_dropped | None) CUDA out.
    ``timing`` (a dict, measurement runs only): the call goes through ``smrf_progressive_filter_timed_*`` and the dict
    receives ``window_ms`` (device time per window) and ``route`` (``_lib.ROUTE_*`` per window)."""
    torch = _torch()
    lib = _lib.load()
    rows, cols = Zd.shape
    mask = torch.empty((rows, cols), dtype=torch.uint8, device=Zd.device)
    when = torch.empty((rows, cols), dtype=torch.uint8, device=Zd.device) if want_when else None
    if rows == 0 or cols == 0:
        return mask, when
    win = np.ascontiguousarray(np.asarray(windows).astype(np.int32))
    thr = np.ascontiguousarray(np.asarray(thresholds, dtype=np.float64))
    if win.ndim != 1 or thr.shape != win.shape:
        raise ValueError("windows must be a 1-D array")
    nbytes = lib.smrf_progressive_filter_workspace_bytes(rows, cols, Zd.element_size())
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Zd.device)
    if timing is not None:
        ms = np.zeros(win.size, dtype=np.float32)
        route = np.zeros(win.size, dtype=np.int32)
        fn = getattr(lib, "smrf_progressive_filter_timed_" + _suffix(Zd))
        _lib.check(fn(_ptr(Zd), rows, cols, win.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p),
                      int(win.size), _ptr(mask), _ptr(when), _ptr(ws), nbytes, int(nan_aware), int(impl), _stream(),
                      ms.ctypes.data_as(C.c_void_p), route.ctypes.data_as(C.c_void_p)))
        timing["window_ms"], timing["route"] = ms, route
        return mask, when
    fn = getattr(lib, "smrf_progressive_filter_" + _suffix(Zd))
    _lib.check(fn(_ptr(Zd), rows, cols, win.ctypes.data_as(C.c_void_p), thr.ctypes.data_as(C.c_void_p),
                  int(win.size), _ptr(mask), _ptr(when), _ptr(ws), nbytes, int(nan_aware), int(impl), _stream()))
    return mask, when


@_device_scoped
def progressive_filter(Z, windows, cellsize=1, slope_threshold=.15, return_when_dropped=False, *,
                       impl=_lib.IMPL_AUTO):
    """Iterative grey opening with growing disks and slope-scaled thresholds -> object mask.

    Same arguments, results and quirks as neilpy.progressive_filter: ``windows`` is an array of
    integer radii, ``disk(window)`` is used for every window (also window 1), ``when_dropped``
    holds the 0-based index of the last window that flagged the cell, ``Z`` is not modified.
    """
    elevation_thresholds = slope_threshold * (windows * cellsize)         # neilpy.py:1661 verbatim
    was_tensor = _is_tensor(Z)
    Zd = _to_device(Z)
    if Zd.dim() != 2:
        raise ValueError("expected a 2-D raster")
    if len(windows) > 256 and return_when_dropped:
        raise OverflowError("when_dropped is uint8: more than 256 windows overflow it (as in the reference)")
    mask, when = _progressive_filter_device(Zd, windows, elevation_thresholds, return_when_dropped, impl)
    if was_tensor:
        mask = mask.view(_torch().bool)                    # the kernels write 0 / 1: a view, not a 268 MB copy
        return (mask, when) if return_when_dropped else mask
    m = _d2h(mask).view(np.bool_)                        # the kernels write 0 / 1
    return (m, _d2h(when)) if return_when_dropped else m


# ------------------------------------------------------------------------------------------
# create_dem  (neilpy.py:1110-1166)
# ------------------------------------------------------------------------------------------
def _dem_edges(xd, yd, cellsize):
    """Cell edges of create_dem's raster from the points' extent (neilpy.py:1117-1124)."""
    torch = _torch()
    lib = _lib.load()
    npts = xd.numel()
    if npts == 0:
        raise ValueError("zero-size array to reduction operation minimum which has no identity")
    ws = torch.empty(4 * 1024, dtype=torch.float64, device=xd.device)
    ext = (C.c_double * 4)()
    _lib.check(lib.smrf_points_extent_f64(_ptr(xd), _ptr(yd), npts, ext, _ptr(ws), ws.numel() * 8, _stream()))
    return _edges_from_extent(*(np.float64(v) for v in ext), cellsize)


def _edges_from_extent(xmin, xmax, ymin, ymax, cellsize):
    """neilpy.py:1117-1124: cell centres on multiples of the cellsize, one spare row / column on the far side"""
    xedges = np.arange(cellsize * np.floor(xmin / cellsize) - .5 * cellsize,
                       cellsize * np.ceil(xmax / cellsize) + 1.5 * cellsize, cellsize)
    yedges = np.arange(cellsize * np.ceil(ymax / cellsize) + .5 * cellsize,
                       cellsize * np.floor(ymin / cellsize) - 1.5 * cellsize, -cellsize)
    return xedges, yedges


def _create_dem_device(xd, yd, zd, cellsize, bin_type, edges):
    """Device core: returns (float64 grid CUDA tensor, uint8 empty mask CUDA tensor, transform)."""
    torch = _torch()
    lib = _lib.load()
    npts = xd.numel()
    h_filter = None
    if edges is None:
        xedges, yedges = _dem_edges(xd, yd, cellsize)
    else:
        xedges, yedges = edges[0], edges[1]
        h_filter = (C.c_double * 4)(float(xedges[0]), float(xedges[-1]), float(yedges[-1]), float(yedges[0]))
        cellsize = np.abs(xedges[1] - xedges[0])
    nx, ny = len(xedges) - 1, len(yedges) - 1
    t = from_origin(xedges[0], yedges[0], cellsize, cellsize)
    inv = ~t
    h_inv = (C.c_double * 6)(*[float(v) for v in tuple(inv)[:6]])
    is_max = 1 if bin_type == 'max' else 0
    keys = torch.empty((max(ny, 0), max(nx, 0)), dtype=torch.int64, device=xd.device)
    n_out = torch.zeros(1, dtype=torch.int64, device=xd.device)
    grid = torch.empty((max(ny, 0), max(nx, 0)), dtype=torch.float64, device=xd.device)
    empty = torch.empty((max(ny, 0), max(nx, 0)), dtype=torch.uint8, device=xd.device)
    if nx < 1 or ny < 1:
        if npts:
            raise ValueError("invalid entry in coordinates array")
        return grid, empty, t
    _lib.check(lib.smrf_grid_clear_u64(_ptr(keys), keys.numel(), _stream()))
    _lib.check(lib.smrf_grid_bin_f64(_ptr(xd), _ptr(yd), _ptr(zd), npts, h_inv, h_filter, _ptr(keys), ny, nx, 0, ny,
                                     is_max, _ptr(n_out), _stream()))
    if int(n_out.item()) > 0:
        raise ValueError("invalid entry in coordinates array")       # np.ravel_multi_index, neilpy.py:1151
    if bin_type not in ('max', 'min'):
        raise ValueError('This type not supported.')                  # neilpy.py:1158
    _lib.check(lib.smrf_grid_finalize_f64(_ptr(keys), _ptr(grid), _ptr(empty), keys.numel(), is_max, _stream()))
    return grid, empty, t


def _points_to_device(x, y, z):
    torch = _torch()
    xd = _to_device(x if _is_tensor(x) else np.asarray(x, dtype=np.float64), torch.float64).reshape(-1)
    yd = _to_device(y if _is_tensor(y) else np.asarray(y, dtype=np.float64), torch.float64).reshape(-1)
    zd = _to_device(z if _is_tensor(z) else np.asarray(z, dtype=np.float64), torch.float64).reshape(-1)
    if not (xd.numel() == yd.numel() == zd.numel()):
        raise ValueError("x, y and z must have the same length")
    return xd, yd, zd


@_device_scoped
def create_dem(x, y, z, cellsize=1, bin_type='max', inpaint=False, edges=None, use_binned_statistic=False):
    """Grid (x, y, z) points to a min-/max-Z raster; empty cells are NaN.  Returns ``(I, t)``.

    Same arguments and results as neilpy.create_dem (float64 grid, row 0 = north, affine
    transform ``t``).  ``use_binned_statistic=True`` is the reference's acknowledged-broken
    branch (:1148-1149) and is not provided.
    """
    if use_binned_statistic:
        raise NotImplementedError("use_binned_statistic=True (neilpy.py:1148) is not supported")
    was_tensor = _is_tensor(x)
    xd, yd, zd = _points_to_device(x, y, z)
    grid, _, t = _create_dem_device(xd, yd, zd, cellsize, bin_type, edges)
    if inpaint == True:  # noqa: E712  (the reference's own test, :1163)
        _springs_device(grid)
    return (grid if was_tensor else _d2h(grid)), t


# ------------------------------------------------------------------------------------------
# inpaint_nans_by_springs  (neilpy.py:1227-1271)
# ------------------------------------------------------------------------------------------
def _springs_device(Ad, key="inpaint"):
    """In-place LSQR spring fill of a contiguous float64 CUDA raster; returns (istop, itn)."""
    torch = _torch()
    lib = _lib.load()
    rows, cols = Ad.shape
    if rows == 0 or cols == 0:
        return 0, 0
    nbytes = lib.smrf_springs_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Ad.device)
    istop, itn, nunk = C.c_int(0), C.c_int64(0), C.c_int64(0)
    _lib.check(lib.smrf_springs_lsqr_f64(_ptr(Ad), rows, cols, 1e-6, 1e-6, 1e8, -1, C.byref(istop), C.byref(itn),
                                         C.byref(nunk), _ptr(ws), nbytes, _stream()))
    last_stats[key] = dict(istop=istop.value, itn=itn.value, n_unknown=nunk.value)
    return istop.value, itn.value


@_device_scoped
def inpaint_nans_by_springs(A, inplace=False, neighbors=4):
    """Fill NaNs by least-squares springs to the 4 neighbours, stopped where SciPy's LSQR stops.

    Same arguments and results as neilpy.inpaint_nans_by_springs (``neighbors`` is accepted and
    ignored there too); ``inplace=True`` writes into ``A`` and returns ``None``.
    """
    torch = _torch()
    if _is_tensor(A):
        if A.dtype != torch.float64:
            raise TypeError("inpaint_nans_by_springs works in float64, as the reference does")
        work = A if (inplace and A.is_cuda and A.is_contiguous()) else _to_device(A).clone()
        _springs_device(work)
        if inplace:
            if work is not A:
                A.copy_(work)
            return None
        return work
    arr = np.asarray(A)
    if arr.ndim != 2:
        raise ValueError("expected a 2-D raster")
    work = _to_device(arr.astype(np.float64, copy=False), torch.float64).clone()
    _springs_device(work)
    out = _d2h(work)
    if inplace:
        A[...] = out
        return None
    return out.astype(arr.dtype, copy=False) if arr.dtype == np.float64 else out


# ------------------------------------------------------------------------------------------
# inpaint_nans_by_fda  (neilpy.py:1170-1216)
# ------------------------------------------------------------------------------------------
def _fda_device(Ad, key="inpaint_fda"):
    """In-place LSQR finite-difference fill of a contiguous float64 CUDA raster; returns (istop, itn)."""
    torch = _torch()
    lib = _lib.load()
    rows, cols = Ad.shape
    if rows == 0 or cols == 0:
        return 0, 0
    nbytes = lib.smrf_fda_workspace_bytes(rows, cols)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=Ad.device)
    istop, itn, nunk = C.c_int(0), C.c_int64(0), C.c_int64(0)
    _lib.check(lib.smrf_fda_lsqr_f64(_ptr(Ad), rows, cols, 1e-6, 1e-6, 1e8, -1, C.byref(istop), C.byref(itn),
                                     C.byref(nunk), _ptr(ws), nbytes, _stream()))
    last_stats[key] = dict(istop=istop.value, itn=itn.value, n_unknown=nunk.value)
    return istop.value, itn.value


@_device_scoped
def inpaint_nans_by_fda(A, fast=True, inplace=False):
    """Fill NaNs by least squares on the second-difference equations, stopped where SciPy's LSQR stops.

    Same arguments and results as neilpy.inpaint_nans_by_fda.  ``fast`` only pre-filters equations
    that cannot touch a NaN in the reference, so both settings give the same raster (and run the
    same kernels here).  ``inplace=True`` writes into ``A`` and returns ``None``.
    """
    torch = _torch()
    if _is_tensor(A):
        if A.dtype != torch.float64:
            raise TypeError("inpaint_nans_by_fda works in float64")
        if A.dim() != 2 or A.shape[0] < 2 or A.shape[1] < 2:
            raise ValueError("negative dimensions are not allowed")
        work = A if (inplace and A.is_cuda and A.is_contiguous()) else _to_device(A).clone()
        _fda_device(work)
        if inplace:
            if work is not A:
                A.copy_(work)
            return None
        return work
    arr = np.asarray(A)
    if arr.ndim != 2:
        raise ValueError("expected a 2-D raster")
    if arr.shape[0] < 2 or arr.shape[1] < 2:
        raise ValueError("negative dimensions are not allowed")      # the reference's np.ones(2*n*(m-2)) at :1190
    work = _to_device(arr.astype(np.float64, copy=False), torch.float64).clone()
    _fda_device(work)
    out = _d2h(work)
    if inplace:
        A[...] = out
        return None
    return out


# ------------------------------------------------------------------------------------------
# smrf  (neilpy.py:1685-1808)
# ------------------------------------------------------------------------------------------
def _classify_points_device(Zpro_d, t, cellsize, xd, yd, zd, elevation_threshold, elevation_scaler):
    """smrf's tail on the device (neilpy.py:1768-1795): slope raster, the two interpolating bicubic
    splines evaluated at the points' fractional pixel coordinates, and the point test.
    Returns ``(elevation, slope, is_object (uint8), row, col)`` as CUDA vectors over the points."""
    torch = _torch()
    lib = _lib.load()
    rows, cols = Zpro_d.shape
    if rows < 4 or cols < 4:
        raise ValueError("the bicubic spline of the point classification needs at least 4 x 4 cells")
    S_d = torch.empty_like(Zpro_d)
    _lib.check(lib.smrf_gradient_slope_f64(_ptr(Zpro_d), _ptr(S_d), rows, cols, float(cellsize), _stream()))   # :1785-1786

    # RectBivariateSpline(row_centers, col_centers, .).ev(r, c) for Zpro and S on the device (:1768-1790)
    from . import spline as _spline
    h_inv = (C.c_double * 6)(*[float(v) for v in tuple(~t)[:6]])
    c_d, r_d = torch.empty_like(xd), torch.empty_like(xd)
    _lib.check(lib.smrf_affine_apply_f64(_ptr(xd), _ptr(yd), xd.numel(), h_inv, _ptr(c_d), _ptr(r_d), _stream()))  # :1772
    tx, lur = _spline.axis_factors(rows)
    ty, luc = _spline.axis_factors(cols)
    tx_d, ty_d = torch.from_numpy(tx).to(Zpro_d.device), torch.from_numpy(ty).to(Zpro_d.device)
    lur_d, luc_d = torch.from_numpy(lur).to(Zpro_d.device), torch.from_numpy(luc).to(Zpro_d.device)
    npts = xd.numel()
    vals = []
    scratch = torch.empty_like(Zpro_d)
    for plane in (Zpro_d, S_d):
        coef = plane.clone()
        _lib.check(lib.smrf_spline_solve_ws_f64(_ptr(coef), _ptr(scratch), rows, cols, _ptr(lur_d), _ptr(luc_d), _stream()))
        out = torch.empty(npts, dtype=torch.float64, device=Zpro_d.device)
        _lib.check(lib.smrf_spline_eval_f64(_ptr(coef), rows, cols, _ptr(tx_d), _ptr(ty_d), _ptr(r_d), _ptr(c_d), npts,
                                            _ptr(out), _stream()))
        vals.append(out)
    elev_d, slope_d = vals
    isobj_d = torch.empty(npts, dtype=torch.uint8, device=Zpro_d.device)
    _lib.check(lib.smrf_classify_points_f64(_ptr(elev_d), _ptr(slope_d), _ptr(zd), npts, float(elevation_threshold),
                                            float(elevation_scaler), _ptr(isobj_d), _stream()))      # :1794-1795
    return elev_d, slope_d, isobj_d, r_d, c_d


@_device_scoped
def smrf(x, y, z, cellsize=1, windows=5, slope_threshold=.15, elevation_threshold=.5,
         elevation_scaler=1.25, low_filter_slope=5, low_outlier_fill=False,
         return_extras=False):
    """Simple Morphological Filter: returns ``(dtm, transform, object_grid, object_vector[, extras])``.

    Same arguments, defaults and results as neilpy.smrf.  Gridding, both inpaints, both
    progressive filters and the slope raster run on the GPU with the rasters resident in HBM
    between stages, and so do the bicubic spline solve / evaluation and the point test of the tail
    (:1768-1795); only 1-D knot vectors and banded LU factors are prepared on the host.
    NumPy / pandas points in -> NumPy results, as the reference; CUDA tensors in -> CUDA tensors
    out (``dtm`` float64, ``object_grid`` and ``object_vector`` bool), nothing crosses PCIe.
    """
    torch = _torch()
    if np.isscalar(windows):
        windows = np.arange(windows) + 1
    xd, yd, zd = _points_to_device(x, y, z)
    Zmin, empty, t = _create_dem_device(xd, yd, zd, cellsize, 'min', None)          # :1741-1742
    _springs_device(Zmin, "inpaint1")                                               # :1743
    low_thr = low_filter_slope * (np.array([1]) * cellsize)
    lib = _lib.load()
    neg = torch.empty_like(Zmin)
    _lib.check(lib.smrf_negate_f64(_ptr(Zmin), _ptr(neg), Zmin.numel(), _stream()))
    low, _ = _progressive_filter_device(neg, np.array([1]), low_thr, False, nan_aware=0)              # :1744
    del neg
    if low_outlier_fill:                                                            # :1747-1749
        _lib.check(lib.smrf_mask_apply_f64(_ptr(Zmin), _ptr(low), None, None, None, Zmin.numel(), _stream()))
        _springs_device(Zmin, "inpaint1b")
    thr = slope_threshold * (windows * cellsize)
    obj, drop = _progressive_filter_device(Zmin, windows, thr, bool(return_extras), nan_aware=0)       # :1752-1755
    object_cells = torch.empty_like(obj)
    _lib.check(lib.smrf_mask_apply_f64(_ptr(Zmin), _ptr(empty), _ptr(low), _ptr(obj), _ptr(object_cells),
                                       Zmin.numel(), _stream()))                    # :1762-1763
    _springs_device(Zmin, "inpaint2")                                               # :1764
    Zpro_d = Zmin
    elev_d, slope_d, isobj_d, r_d, c_d = _classify_points_device(Zpro_d, t, cellsize, xd, yd, zd, elevation_threshold,
                                                                 elevation_scaler)
    # the spline values are diagnostics (tests compare them with FITPACK's); 16 B per point of HBM, so they are only
    # kept alive past the call when the caller asked for the extras
    last_stats.pop("tail", None)
    if return_extras:
        last_stats["tail"] = _DeviceValues(elevation_values=elev_d, slope_values=slope_d)
    if _is_tensor(x) and _is_tensor(y) and _is_tensor(z):
        # CUDA tensors in -> CUDA tensors out: nothing of the result crosses PCIe
        obj_t, pts_t = object_cells.view(torch.bool), isobj_d.view(torch.bool)     # 0 / 1 bytes: views
        if not return_extras:
            return Zpro_d, t, obj_t, pts_t
        ri, ci = torch.round(r_d).long(), torch.round(c_d).long()      # round-half-even, as np.round
        if ri.numel() and (int(ri.min()) < -drop.shape[0] or int(ri.max()) >= drop.shape[0] or
                           int(ci.min()) < -drop.shape[1] or int(ci.max()) >= drop.shape[1]):
            # the reference indexes drop_raster with the rounded pixel coordinates (:1780) and NumPy raises for a
            # point that rounds onto the raster's far edge; a device gather must not be handed such an index
            raise IndexError("index %d is out of bounds for the drop raster of shape %s"
                             % (max(int(ri.max()), int(ci.max())), tuple(drop.shape)))
        extras = {'above_ground_height': zd - elev_d, 'drop_raster': drop, 'when_dropped': drop[ri, ci]}
        return Zpro_d, t, obj_t, pts_t, extras
    Zpro = _d2h(Zpro_d)
    is_object_point = _d2h(isobj_d).view(np.bool_)
    try:                                            # the reference returns a Series when z is one (:1795)
        import pandas as pd
        if isinstance(z, pd.Series):
            is_object_point = pd.Series(is_object_point, index=z.index, name=z.name)
    except ImportError:  # pragma: no cover
        pass
    obj_np = _d2h(object_cells).view(np.bool_)
    if not return_extras:
        return Zpro, t, obj_np, is_object_point
    zh = z if not _is_tensor(z) else _d2h(z)
    elevation_values = _d2h(elev_d)
    drop_raster = _d2h(drop)
    r, c = _d2h(r_d), _d2h(c_d)
    extras = {'above_ground_height': zh - elevation_values, 'drop_raster': drop_raster,
              'when_dropped': drop_raster[np.round(r).astype(int), np.round(c).astype(int)]}
    return Zpro, t, obj_np, is_object_point, extras


class _DeviceValues:
    """Read-only mapping of named device vectors, copied to NumPy on first access."""

    def __init__(self, **tensors):
        self._t, self._np = tensors, {}

    def __getitem__(self, key):
        if key not in self._np:
            self._np[key] = _d2h(self._t[key])
        return self._np[key]

    def keys(self):
        return self._t.keys()

    def __contains__(self, key):
        return key in self._t


# ------------------------------------------------------------------------------------------
# pssm  (neilpy.py:846-867) - the bonemaps of examples/smrf/*.ipynb
# ------------------------------------------------------------------------------------------
_lut_cache = {}


@_device_scoped
def pssm(Z, cellsize=1, ve=2.3, reverse=False, apply_colormap=True):
    """Perceptually scaled slope map, same arguments and results as neilpy.pssm.

    ``apply_colormap=True``: float64 RGBA raster ``(rows, cols, 4)`` looked up in the bone
    colormap (``bone_r`` unless ``reverse``); ``False``: the uint8 slope classes
    ``round(255 * degrees(arctan(ve * slope)) / 90)``.  One fused kernel (gradient, slope, class,
    colour).  The arithmetic is float64 (a float32 or integer raster is widened first; the
    reference keeps float32 rasters in float32, which can move a class by one at a rounding tie).
    """
    torch = _torch()
    was_tensor = _is_tensor(Z)
    Zd = _to_device(Z, torch.float64)
    if Zd.dim() != 2:
        raise ValueError("expected a 2-D raster")
    rows, cols = Zd.shape
    if rows < 2 or cols < 2:
        raise ValueError("Shape of array too small to calculate a numerical gradient, "
                         "at least (edge_order + 1) elements are required.")      # np.gradient's message
    lib = _lib.load()
    P = rgba = lut = None
    if apply_colormap:
        key = (bool(reverse), Zd.device.index)
        if key not in _lut_cache:
            from .colormap import bone_lut
            _lut_cache[key] = torch.from_numpy(np.array(bone_lut(reverse=not reverse))).to(Zd.device)
        lut = _lut_cache[key]
        rgba = torch.empty((rows, cols, 4), dtype=torch.float64, device=Zd.device)
    else:
        P = torch.empty((rows, cols), dtype=torch.uint8, device=Zd.device)
    _lib.check(lib.smrf_pssm_f64(_ptr(Zd), _ptr(P), _ptr(rgba), _ptr(lut), rows, cols, float(cellsize), float(ve),
                                 _stream()))
    out = rgba if apply_colormap else P
    return out if was_tensor else _d2h(out)
