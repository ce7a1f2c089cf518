"""Affine transform returned by create_dem / smrf.

The reference returns ``rasterio.transform.from_origin(...)``, an ``affine.Affine``
(neilpy.py:1141).  When the ``affine`` package is importable its class is used, so callers get
the very same type; otherwise this module's 9-tuple with the same arithmetic is used.
"""

try:  # pragma: no cover - depends on the environment
    from affine import Affine as _ExternalAffine
except Exception:  # noqa: BLE001
    _ExternalAffine = None


class Affine(tuple):
    """(a, b, c, d, e, f, 0, 0, 1): x' = a*x + b*y + c, y' = d*x + e*y + f."""

    def __new__(cls, a, b, c, d, e, f):
        return tuple.__new__(cls, (a, b, c, d, e, f, 0.0, 0.0, 1.0))

    a = property(lambda s: s[0])
    b = property(lambda s: s[1])
    c = property(lambda s: s[2])
    d = property(lambda s: s[3])
    e = property(lambda s: s[4])
    f = property(lambda s: s[5])

    @property
    def determinant(self):
        return self[0] * self[4] - self[1] * self[3]

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

    def __repr__(self):
        return "Affine(%r, %r, %r,\n       %r, %r, %r)" % tuple(self[:6])


def from_origin(west, north, xsize, ysize):
    """rasterio.transform.from_origin: translation(west, north) * scale(xsize, -ysize)."""
    if _ExternalAffine is not None:
        return _ExternalAffine.translation(west, north) * _ExternalAffine.scale(xsize, -ysize)
    return Affine(1.0, 0.0, west, 0.0, 1.0, north) * Affine(xsize, 0.0, 0.0, 0.0, -ysize, 0.0)


def edges_from_IT(Image, Transform):
    """neilpy.edges_from_IT (neilpy/neilpy.py:1095-1102): the x and y cell edges of a raster with transform
    ``Transform`` - pixel corners (0..cols, 0) and (0, 0..rows) mapped to world coordinates - in the form
    ``create_dem(..., edges=(x_edges, y_edges))`` takes, so that a second cloud is gridded onto the first one's cells.
    Host arithmetic on rows + cols + 2 values; ``Image`` only gives the shape (array, tensor or anything with ``.shape``)."""
    import numpy as np
    shape = getattr(Image, "shape", None) or np.shape(Image)
    r, c = int(shape[0]), int(shape[1])
    j, i = np.arange(c + 1), np.arange(r + 1)
    sa, sb, sc, sd, se, sf = (float(v) for v in tuple(Transform)[:6])
    # the same expression order as affine's __mul__ on a (vx, vy) pair: vx * a + vy * b + c, vx * d + vy * e + f
    x_edges = j * sa + np.zeros_like(j) * sb + sc
    y_edges = np.zeros_like(i) * sd + i * se + sf
    return x_edges, y_edges


def write_worldfile(affine_matrix, output_file):
    """neilpy.write_worldfile (neilpy/neilpy.py:1564-1570): the six world-file lines (pixel width,
    column rotation, row rotation, pixel height, x and y of the centre of the upper-left pixel),
    ``%0.10f`` each, for the rasters ``pssm`` / ``smrf`` return with transform ``affine_matrix``."""
    x_ul_center, y_ul_center = affine_matrix * (.5, .5)
    pixel_width, row_rotation = affine_matrix[0], affine_matrix[1]
    pixel_height, col_rotation = affine_matrix[4], affine_matrix[3]
    world = [pixel_width, col_rotation, row_rotation, pixel_height, x_ul_center, y_ul_center]
    with open(output_file, "w") as fh:
        for v in world:
            fh.write("%0.10f\n" % v)
