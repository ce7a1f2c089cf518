"""Seeded synthetic inputs for the SMRF hot path.

The generators use ``numpy.random`` PCG64 streams, so the same seed gives the same array on any
host.  ``synth_dem`` draws its random fields per 256-row block from a stream keyed by
``(seed, block)``: any row band of a large DEM can be generated on its own (each rank of a
sharded run builds only its rows) and equals the corresponding rows of the full DEM.
Used by ``bench.py``, the parity tests and ``tests/golden/make_golden.py``; no GPU involved.
"""
import numpy as np

__all__ = ["synth_dem", "synth_points", "terrain"]

_BLOCK = 256


def terrain(xx, yy):
    """Smooth analytic terrain shared by :func:`synth_dem` and :func:`synth_points`."""
    return (40.0 * np.sin(2.0 * np.pi * xx / 1024.0) * np.cos(2.0 * np.pi * yy / 1536.0)
            + 0.01 * xx + 0.005 * yy + 300.0)


def synth_dem(n, seed=20240, dtype=np.float32, rows=None, row_range=None):
    """``rows x n`` DEM (``rows`` defaults to ``n``): terrain + boxes ("buildings", ``rows*n/8192``
    of them, half-sizes 3..39 cells, 3..30 m high) + 3 % salt ("vegetation", 1..20 m) + N(0, 0.03)
    noise.  ``row_range=(r0, r1)`` returns only those rows of the same DEM."""
    m = n if rows is None else int(rows)
    r0, r1 = (0, m) if row_range is None else (int(row_range[0]), int(row_range[1]))
    rng = np.random.default_rng([seed, 0xB0C5])
    k = max(1, (m * n) // 8192)
    cy = rng.integers(0, m, size=k)
    cx = rng.integers(0, n, size=k)
    hy = rng.integers(3, 40, size=k)
    hx = rng.integers(3, 40, size=k)
    hh = rng.uniform(3.0, 30.0, size=k)
    xs = np.arange(n, dtype=np.float64)[None, :]
    ys = np.arange(r0, r1, dtype=np.float64)[:, None]
    base = terrain(xs, ys)
    out = base.copy()
    sel = np.flatnonzero((cy + hy >= r0) & (cy - hy < r1))
    for i in sel:
        y0, y1 = max(r0, cy[i] - hy[i]), min(r1, cy[i] + hy[i] + 1)
        x0, x1 = max(0, cx[i] - hx[i]), min(n, cx[i] + hx[i] + 1)
        if y0 >= y1:
            continue
        np.maximum(out[y0 - r0:y1 - r0, x0:x1], base[y0 - r0:y1 - r0, x0:x1] + hh[i],
                   out=out[y0 - r0:y1 - r0, x0:x1])
    del base
    for blk in range(r0 // _BLOCK, (r1 + _BLOCK - 1) // _BLOCK):
        brng = np.random.default_rng([seed, 0xF1E1D, blk])
        b0, b1 = blk * _BLOCK, min(m, (blk + 1) * _BLOCK)
        veg = brng.random((b1 - b0, n)) < 0.03
        vh = brng.uniform(1.0, 20.0, size=(b1 - b0, n))
        noise = brng.normal(0.0, 0.03, size=(b1 - b0, n))
        field = veg * vh + noise
        lo, hi = max(b0, r0), min(b1, r1)
        out[lo - r0:hi - r0] += field[lo - b0:hi - b0]
    return out.astype(dtype)


def synth_points(npts, extent, seed=20241):
    """``npts`` lidar-like points over ``[0.5, extent-0.5]^2`` (float64 x, y, z)."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(0.5, extent - 0.5, size=npts)
    y = rng.uniform(0.5, extent - 0.5, size=npts)
    z = terrain(x, y)
    obj = rng.random(npts) < 0.2
    z = z + np.abs(rng.normal(0.0, 0.5, size=npts)) * obj * 20.0
    return x, y, z
