"""The 'bone' colormap tables pssm() looks its uint8 slope classes up in.

The reference calls ``plt.cm.bone_r(P)`` / ``plt.cm.bone(P)`` (neilpy/neilpy.py:861-864): for an
integer image matplotlib indexes the colormap's 256-entry RGBA table directly.  The table is
rebuilt here from the colormap's published piecewise-linear definition, so the device path does
not need matplotlib; tests/test_host_logic.py checks it entry for entry against matplotlib.
"""
import functools

import numpy as np

# (x, y_left, y_right) anchors of each channel: the 'bone' definition (grey with a blue tint)
_BONE = {
    "red": ((0., 0., 0.), (0.746032, 0.652778, 0.652778), (1.0, 1.0, 1.0)),
    "green": ((0., 0., 0.), (0.365079, 0.319444, 0.319444), (0.746032, 0.777778, 0.777778), (1.0, 1.0, 1.0)),
    "blue": ((0., 0., 0.), (0.365079, 0.444444, 0.444444), (1.0, 1.0, 1.0)),
}


def _channel_table(anchors, n):
    """n samples of the piecewise-linear channel through ``anchors`` (x ascending on [0, 1])."""
    a = np.array(anchors, dtype=np.float64)
    x, left, right = a[:, 0] * (n - 1), a[:, 1], a[:, 2]
    xi = (n - 1) * np.linspace(0, 1, n)
    seg = np.searchsorted(x, xi)[1:-1]
    frac = (xi[1:-1] - x[seg - 1]) / (x[seg] - x[seg - 1])
    inner = frac * (left[seg] - right[seg - 1]) + right[seg - 1]
    return np.clip(np.concatenate([[right[0]], inner, [left[-1]]]), 0.0, 1.0)


@functools.lru_cache(maxsize=None)
def _bone_lut_cached(reverse):
    out = np.ones((256, 4), dtype=np.float64)
    for k, ch in enumerate(("red", "green", "blue")):
        anchors = _BONE[ch]
        if reverse:                                       # matplotlib's Colormap.reversed()
            anchors = tuple((1.0 - x, y1, y0) for x, y0, y1 in reversed(anchors))
        out[:, k] = _channel_table(anchors, 256)
    out.setflags(write=False)
    return out


def bone_lut(reverse=False):
    """256 x 4 float64 RGBA table of ``plt.cm.bone`` (``reverse=True``: ``plt.cm.bone_r``)."""
    return _bone_lut_cached(bool(reverse))
