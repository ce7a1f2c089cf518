"""read_las against the reference's read_las (goldens from tests/golden/make_golden.py las)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, golden

LAS = golden("las.npz")
CASES = [str(c) for c in LAS["cases"]]


@pytest.mark.parametrize("tag", CASES)
def test_read_las_matches_reference(tag):
    import neilpy_amd
    header, df = neilpy_amd.read_las(os.path.join(GOLDEN, "las", tag + ".las"))
    want = json.loads(str(LAS[tag + "_header_json"]))
    got = {k: (list(v) if isinstance(v, tuple) else v) for k, v in header.items()}
    assert got == want
    assert list(df.columns) == [str(c) for c in LAS[tag + "_columns"]]
    for c in df.columns:
        w = LAS[tag + "_col_" + c]
        assert df[c].values.dtype == w.dtype, (c, df[c].values.dtype, w.dtype)
        assert np.array_equal(df[c].values, w, equal_nan=True), c


def test_extra_bytes_and_laz(tmp_path):
    import neilpy_amd
    rng = np.random.default_rng(1)
    x, y, z = (np.round(rng.uniform(10, 20, 40), 2) for _ in range(3))
    fn = str(tmp_path / "extra.las")
    neilpy_amd.write_las(fn, x, y, z, fmt=1, extra_bytes=5)            # user-defined extra bytes per record
    header, df = neilpy_amd.read_las(fn)
    assert header["point_data_record_length"] == 33 and len(df) == 40
    np.testing.assert_allclose(df.x.values, x, atol=1e-9)
    raw = bytearray(open(fn, "rb").read())
    raw[104] = 128 + 1                                                # LAZ-compressed format id
    open(fn, "wb").write(bytes(raw))
    with pytest.raises(ValueError, match="LAZ not yet supported."):
        neilpy_amd.read_las(fn)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_read_las_xyz_gpu(tag, gpu_device):
    import neilpy_amd
    header, x, y, z = neilpy_amd.read_las_xyz(os.path.join(GOLDEN, "las", tag + ".las"))
    assert x.is_cuda and x.dtype.is_floating_point
    for t, c in ((x, "x"), (y, "y"), (z, "z")):
        assert np.array_equal(t.cpu().numpy(), LAS[tag + "_col_" + c])   # bit-exact with the reference's floats


@pytest.mark.gpu
def test_smrf_from_las_file(tmp_path, gpu_device):
    """LAS file -> device decode -> smrf equals smrf on the same points given as arrays"""
    import neilpy_amd
    from conftest import load_sample
    x, y, z, g = load_sample("samp24")
    fn = str(tmp_path / "samp24.las")
    neilpy_amd.write_las(fn, x, y, z, fmt=1, scale=(0.01, 0.01, 0.01), offset=(512000.0, 5403000.0, 0.0))
    header, xd, yd, zd = neilpy_amd.read_las_xyz(fn)
    a = neilpy_amd.smrf(xd, yd, zd, 1, 18, .15, .5, 1.25)
    hx, df = neilpy_amd.read_las(fn)
    b = neilpy_amd.smrf(df.x.values, df.y.values, df.z.values, 1, 18, .15, .5, 1.25)
    assert a[0].is_cuda and a[3].is_cuda               # device points in -> device results out
    assert np.array_equal(a[2].cpu().numpy(), b[2]) and np.array_equal(a[3].cpu().numpy(), np.asarray(b[3]))
    assert np.array_equal(a[0].cpu().numpy(), b[0])
