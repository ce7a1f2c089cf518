"""The C-ABI library loads and exports every symbol include/smrf_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "smrf_hip.h")).read()
    return sorted(set(re.findall(r"^SMRF_API [\w \*]*?\b(smrf_\w+)\(", text, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from neilpy_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        from neilpy_amd.build import build
        build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), "missing export " + n
    assert sorted(_lib.SIGNATURES) == names, "ctypes SIGNATURES out of sync with smrf_hip.h"
    assert _lib.load().smrf_abi_version() == 1


def test_no_cpu_fallback():
    """without a GPU the product path raises instead of computing on the host"""
    import numpy as np
    import torch
    import neilpy_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(neilpy_amd.SmrfHipError):
        neilpy_amd.progressive_filter(np.zeros((8, 8), np.float32), np.array([1, 2]))
    with pytest.raises(neilpy_amd.SmrfHipError):
        neilpy_amd.smrf(np.arange(9.0), np.arange(9.0), np.arange(9.0))


def test_product_never_imports_oracle():
    """only tests/, smoke() and bench.py's cpu_baseline may touch oracle/"""
    pat = re.compile(r"^\s*(from\s+oracle|import\s+oracle|from\s+\.+oracle)|smrf_oracle|oracle/_ref", re.M)
    pkg = os.path.join(ROOT, "neilpy_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert not pat.search(text), os.path.join(dirpath, f)


def test_device_scope_refuses_mixed_devices():
    """tensors of one call on two devices are refused before any launch (neilpy_amd/_device.py)"""
    from neilpy_amd import SmrfHipError
    from neilpy_amd._device import common_device

    class FakeTensor:                      # what the guard looks at: torch's module, data_ptr, is_cuda, device
        def __init__(self, device, is_cuda=True):
            self.device, self.is_cuda = device, is_cuda

        def data_ptr(self):
            return 0
    FakeTensor.__module__ = "torch"
    a, b, c = FakeTensor("cuda:0"), FakeTensor("cuda:1"), FakeTensor("cpu", is_cuda=False)
    assert common_device([1.0, None, c]) is None
    assert common_device([a, c, "x", a]) == "cuda:0"
    with pytest.raises(SmrfHipError, match="different devices"):
        common_device([a, 3, b])


def test_drop_in_signatures_match_the_reference():
    """Every function neilpy_amd re-exports under a neilpy name takes the reference's parameters - same names, same
    order, same defaults (tests/golden/signatures.json, written by make_golden.py from the imported reference);
    anything this package adds is keyword-only, so a positional neilpy call means the same thing here."""
    import inspect
    import json
    import os

    import neilpy_amd
    from conftest import GOLDEN
    want = json.load(open(os.path.join(GOLDEN, "signatures.json")))
    assert set(want) >= {"smrf", "progressive_filter", "create_dem", "inpaint_nans_by_springs"}
    for name, params in want.items():
        got = list(inspect.signature(getattr(neilpy_amd, name)).parameters.values())
        for i, p in enumerate(params):
            g = got[i]
            assert (g.name, g.kind.name) == (p["name"], p["kind"]), (name, i, g, p)
            assert (None if g.default is inspect.Parameter.empty else repr(g.default)) == p["default"], (name, g, p)
        for g in got[len(params):]:
            assert g.kind is inspect.Parameter.KEYWORD_ONLY, (name, g)


def test_chain_length_size_thresholds():
    """smrf_pf_chain_length is host logic (csrc/chain.hip min_cells): which small windows run as chained / table-free launches
    depends on dtype and raster size.  Round 5 re-measured the thresholds under the new segmentation
    (profiles/r05_segment_balance.md section 7): the chain 4, 5 and the single R = 10 from 20 Mi cells, the single R = 9 and
    the fp64 chain 1, 2, 3 / single R = 5 at any size, the fp64 singles R = 7 / 8 from 4 / 16 Mi cells."""
    import ctypes as C
    import os
    import numpy as np
    from neilpy_amd import _lib
    lib = _lib.load()
    saved = {k: os.environ.pop(k, None) for k in ("SMRF_FUSED", "SMRF_CHAIN")}
    _lib.reload_switches()
    try:
        def n(elem, radii, cells):
            r = np.asarray(radii, dtype=np.int32)
            return lib.smrf_pf_chain_length(elem, r.ctypes.data_as(C.c_void_p), len(r), cells)
        Mi = 1 << 20
        assert n(4, [1, 2, 3, 4], 1 * Mi) == 3 and n(4, [2, 3, 4], 1 * Mi) == 2 and n(4, [1, 2, 4], Mi) == 2
        assert n(4, [4, 5, 6], 16 * Mi) == 1 and n(4, [4, 5, 6], 24 * Mi) == 2
        assert n(4, [9, 10], 1 * Mi) == 1 and n(4, [10, 11], 16 * Mi) == 0 and n(4, [10, 11], 24 * Mi) == 1
        assert n(4, [11], 1 << 40) == 0 and n(4, [6], 1) == 1
        assert n(8, [1, 2, 3], 1 * Mi) == 3 and n(8, [5], 1 * Mi) == 1 and n(8, [4, 5], 1 << 40) == 1     # no fp64 chain 4, 5
        assert n(8, [7], 2 * Mi) == 0 and n(8, [7], 4 * Mi) == 1
        assert n(8, [8], 8 * Mi) == 0 and n(8, [8], 16 * Mi) == 1
        assert n(8, [6], 1 << 40) == 0 and n(8, [9], 1 << 40) == 0
    finally:
        for k, v in saved.items():
            if v is not None:
                os.environ[k] = v
        _lib.reload_switches()
