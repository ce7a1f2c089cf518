"""Seeded synthetic inputs for the SMRF hot path.

The generators use ``numpy.random.default_rng`` (PCG64), so the same seed gives
the same array on any host.  They are used by ``bench.py``, by the parity tests
and by ``tests/golden/make_golden.py``; nothing here touches the GPU.
"""
import numpy as np

__all__ = ["synth_dem", "synth_points", "terrain"]


def terrain(xx, yy):
    """Smooth analytic terrain shared by :func:`synth_dem` and :func:`synth_points`."""
    return (40.0 * np.sin(2.0 * np.pi * xx / 1024.0) * np.cos(2.0 * np.pi * yy / 1536.0)
            + 0.01 * xx + 0.005 * yy + 300.0)


def synth_dem(n, seed=20240, dtype=np.float32, rows=None):
    """``rows x n`` DEM: terrain + boxes ("buildings") + salt ("vegetation") + noise.

    ``rows`` defaults to ``n``.  Memory is bounded by generating in row strips
    for the random fields, but the box list is drawn first so that a ``rows``
    crop of a larger DEM is *not* the same as a smaller DEM - callers that need
    a crop should generate the full DEM and slice it.
    """
    m = n if rows is None else int(rows)
    rng = np.random.default_rng(seed)
    k = max(1, (m * n) // 8192)
    cy = rng.integers(0, m, size=k)
    cx = rng.integers(0, n, size=k)
    hy = rng.integers(3, 40, size=k)
    hx = rng.integers(3, 40, size=k)
    hh = rng.uniform(3.0, 30.0, size=k)
    out = np.empty((m, n), dtype=np.float64)
    strip = max(1, min(m, (1 << 24) // max(n, 1)))
    xs = np.arange(n, dtype=np.float64)[None, :]
    for r0 in range(0, m, strip):
        r1 = min(m, r0 + strip)
        ys = np.arange(r0, r1, dtype=np.float64)[:, None]
        out[r0:r1] = terrain(xs, ys)
    base = out.copy() if k else out
    for i in range(k):
        y0, y1 = max(0, cy[i] - hy[i]), min(m, cy[i] + hy[i] + 1)
        x0, x1 = max(0, cx[i] - hx[i]), min(n, cx[i] + hx[i] + 1)
        np.maximum(out[y0:y1, x0:x1], base[y0:y1, x0:x1] + hh[i], out=out[y0:y1, x0:x1])
    del base
    for r0 in range(0, m, strip):
        r1 = min(m, r0 + strip)
        veg = rng.random((r1 - r0, n)) < 0.03
        vh = rng.uniform(1.0, 20.0, size=(r1 - r0, n))
        out[r0:r1] += veg * vh
        out[r0:r1] += rng.normal(0.0, 0.03, size=(r1 - r0, n))
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
