"""pssm() (neilpy/neilpy.py:846-867) and write_worldfile() (:1564-1570): the oracle and the host
tables against the reference's own outputs (tests/golden/pssm.npz, written by make_golden.py
pssm), then the HIP kernel against both."""
import json
import os

import numpy as np
import pytest

from conftest import golden
from oracle import smrf_oracle as orc

G = golden("pssm.npz")
CASES = [str(c) for c in G["cases"]]


def _case(name):
    cellsize, ve = G[name + "_args"]
    cellsize = int(cellsize) if float(cellsize).is_integer() else float(cellsize)
    return G[name + "_Z"], cellsize, float(ve), G[name + "_P"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_classes_golden(name):
    Z, cellsize, ve, P = _case(name)
    got = orc.pssm_classes(Z, cellsize, ve)
    assert got.dtype == np.uint8 and np.array_equal(got, P)


def test_bone_tables_golden():
    from neilpy_amd.colormap import bone_lut
    assert np.array_equal(bone_lut(False), G["lut_bone"])
    assert np.array_equal(bone_lut(True), G["lut_bone_r"])
    assert bone_lut(True).shape == (256, 4) and bone_lut(True).dtype == np.float64


def test_bone_tables_match_matplotlib():
    mpl = pytest.importorskip("matplotlib")
    mpl.use("Agg")
    import matplotlib.pyplot as plt
    from neilpy_amd.colormap import bone_lut
    idx = np.arange(256, dtype=np.uint8)
    assert np.array_equal(bone_lut(False), plt.cm.bone(idx))
    assert np.array_equal(bone_lut(True), plt.cm.bone_r(idx))


def test_oracle_rgba_golden():
    Z, cellsize, ve, _ = _case("hills_c1")
    assert np.array_equal(orc.pssm(Z, G["lut_bone_r"], cellsize, ve), G["hills_c1_rgba"])
    assert np.array_equal(orc.pssm(Z, G["lut_bone"], cellsize, ve), G["hills_c1_rgba_reverse"])


def test_write_worldfile_golden(tmp_path):
    import neilpy_amd
    for rec in G["worldfiles"]:
        rec = json.loads(str(rec))
        t = neilpy_amd.from_origin(*rec["origin"])
        assert orc.worldfile_lines(orc.from_origin(*rec["origin"])) == rec["lines"]
        path = os.path.join(tmp_path, "w.pgw")
        neilpy_amd.write_worldfile(t, path)
        assert open(path).read().split() == rec["lines"]


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_hip_classes_golden(name, gpu_device):
    import neilpy_amd
    Z, cellsize, ve, P = _case(name)
    got = neilpy_amd.pssm(Z, cellsize=cellsize, ve=ve, apply_colormap=False)
    assert got.dtype == np.uint8 and got.shape == P.shape
    assert np.array_equal(got, P)


@pytest.mark.gpu
def test_hip_rgba_golden(gpu_device):
    import neilpy_amd
    Z, cellsize, ve, _ = _case("hills_c1")
    got = neilpy_amd.pssm(Z, cellsize=cellsize)
    assert got.dtype == np.float64 and got.shape == Z.shape + (4,)
    assert np.array_equal(got, G["hills_c1_rgba"])
    assert np.array_equal(neilpy_amd.pssm(Z, cellsize, 2.3, True), G["hills_c1_rgba_reverse"])   # positional, as the notebooks


@pytest.mark.gpu
def test_hip_tensor_in_tensor_out_and_dtypes(gpu_device):
    import torch
    import neilpy_amd
    Z, cellsize, ve, P = _case("plateau_c2")
    out = neilpy_amd.pssm(torch.from_numpy(Z).to(gpu_device), cellsize=cellsize, apply_colormap=False)
    assert out.is_cuda and out.dtype == torch.uint8
    assert np.array_equal(out.cpu().numpy(), P)
    Zi = np.round(Z).astype(np.int32)                      # integer rasters: np.gradient works in float64
    assert np.array_equal(neilpy_amd.pssm(Zi, cellsize=cellsize, apply_colormap=False),
                          orc.pssm_classes(Zi, cellsize))
    with pytest.raises(ValueError):
        neilpy_amd.pssm(np.zeros((1, 8)))


@pytest.mark.gpu
def test_hip_vs_oracle_large_dtm(gpu_device):
    """2048^2 synthetic DTM: classes equal to the oracle in every cell."""
    import neilpy_amd
    Z = neilpy_amd.synth_dem(2048, seed=5).astype(np.float64)
    got = neilpy_amd.pssm(Z, cellsize=1, apply_colormap=False)
    assert np.array_equal(got, orc.pssm_classes(Z, 1))
