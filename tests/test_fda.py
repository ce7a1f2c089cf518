"""inpaint_nans_by_fda (neilpy/neilpy.py:1170-1216): oracle against the reference's outputs
(tests/golden/fda.npz, written by make_golden.py fda), then the HIP solver against both."""
import numpy as np
import pytest

from conftest import golden
from oracle import smrf_oracle as orc

G = golden("fda.npz")
CASES = [str(c) for c in G["cases"]]
# The second-difference system is ill conditioned and LSQR's early-stopped iterate is sensitive to
# the order of floating-point sums: SciPy's own answer moves by up to ~8e-6 (relative to the raster's
# magnitude; measured on the 300 x 260 case below) and its iteration count by 1-2 when the same
# equations are merely permuted (test_scipy_itself_moves_under_permutation).  Parity is therefore:
# same istop, itn within ITN_SLACK, values within RTOL = 2e-5 of the reference, i.e. about twice the
# solver's own order-of-summation noise.
RTOL = 2e-5
ITN_SLACK = 3


def _close_stop(got, want):
    return got[0] == int(want[0]) and abs(got[1] - int(want[1])) <= max(ITN_SLACK, int(want[1]) // 100)


@pytest.mark.parametrize("tag", CASES)
def test_oracle_golden(tag):
    A = G[tag + "_in"]
    B, istop, itn = orc.inpaint_nans_by_fda(A, return_info=True)
    assert (istop, itn) == tuple(G[tag + "_lsqr"])
    assert np.array_equal(B, G[tag + "_out"])                      # same SciPy, same system: bit-equal
    assert bool(G[tag + "_slow_equal"])                            # fast=False gave the same raster in the reference
    A2 = A.copy()
    assert orc.inpaint_nans_by_fda(A2, inplace=True) is None and np.array_equal(A2, B)


def test_reference_errors_recorded():
    for shape in ((1, 30), (30, 1)):
        assert str(G["error_%dx%d" % shape]) == "negative dimensions are not allowed"
        with pytest.raises(ValueError, match="negative dimensions"):
            orc.inpaint_nans_by_fda(np.full(shape, np.nan))


def test_scipy_itself_moves_under_permutation():
    """Why parity is a tolerance here: the reference's own solver, same equations in another order."""
    from scipy.sparse.linalg import lsqr
    rng = np.random.default_rng(0)
    a, b, _ = orc.fda_system(G["occ15_in"])
    base = lsqr(a, b)
    moved = 0.0
    for _ in range(3):
        p = rng.permutation(a.shape[0])
        r = lsqr(a[p], b[p])
        assert r[1] == base[1] and abs(r[2] - base[2]) <= ITN_SLACK + 1
        moved = max(moved, float(np.abs(r[0] - base[0]).max()))
    assert 1e-12 < moved / float(np.abs(base[0]).max()) < RTOL


def test_multiplicity_is_nan_entries_per_row():
    rng = np.random.default_rng(5)
    A = rng.normal(size=(9, 7))
    A[rng.random(A.shape) < .3] = np.nan
    a, b, nan_list = orc.fda_system(A)
    # every kept equation appears as many times as it has stored entries in the NaN columns
    dense = a.toarray()
    uniq, counts = np.unique(dense, axis=0, return_counts=True)
    for row, c in zip(uniq, counts):
        assert c % np.count_nonzero(row) == 0


# ------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tag", CASES)
def test_hip_golden(tag, gpu_device):
    import neilpy_amd
    A = G[tag + "_in"]
    want = G[tag + "_out"]
    got = neilpy_amd.inpaint_nans_by_fda(A)
    st = neilpy_amd.last_stats["inpaint_fda"]
    assert _close_stop((st["istop"], st["itn"]), G[tag + "_lsqr"])
    assert got.dtype == np.float64 and got.shape == want.shape
    known = ~np.isnan(A)
    assert np.array_equal(got[known], A[known])                    # known cells untouched
    scale = max(1.0, float(np.nanmax(np.abs(want)))) if want.size else 1.0
    assert np.max(np.abs(got - want), initial=0.0) <= RTOL * scale
    A2 = A.copy()
    assert neilpy_amd.inpaint_nans_by_fda(A2, True, True) is None  # positional fast, inplace
    assert np.array_equal(A2, got)


@pytest.mark.gpu
def test_hip_errors_and_tensor(gpu_device):
    import torch
    import neilpy_amd
    with pytest.raises(ValueError, match="negative dimensions"):
        neilpy_amd.inpaint_nans_by_fda(np.full((1, 30), np.nan))
    A = G["hole18_in"]
    t = torch.from_numpy(A).to(gpu_device)
    out = neilpy_amd.inpaint_nans_by_fda(t)
    assert out.is_cuda and torch.isnan(t).any() and not torch.isnan(out).any()
    assert np.max(np.abs(out.cpu().numpy() - G["hole18_out"])) <= RTOL * float(np.abs(G["hole18_out"]).max())
    assert neilpy_amd.inpaint_nans_by_fda(t, inplace=True) is None and not torch.isnan(t).any()


@pytest.mark.gpu
def test_hip_vs_oracle_random_holes(gpu_device):
    """A larger raster than the goldens: 300 x 260, 40 % NaN plus a 25-cell hole."""
    import neilpy_amd
    rng = np.random.default_rng(77)
    A = neilpy_amd.synth_dem(512, seed=3).astype(np.float64)[:300, :260].copy()
    A[rng.random(A.shape) < .4] = np.nan
    A[100:125, 60:85] = np.nan
    want, istop, itn = orc.inpaint_nans_by_fda(A, return_info=True)
    got = neilpy_amd.inpaint_nans_by_fda(A)
    st = neilpy_amd.last_stats["inpaint_fda"]
    assert _close_stop((st["istop"], st["itn"]), (istop, itn))
    assert np.max(np.abs(got - want)) <= RTOL * float(np.abs(want).max())


@pytest.mark.gpu
@pytest.mark.parametrize("shape,frac,seed", [((9, 7), .3, 1), ((40, 33), .5, 2), ((64, 300), .2, 3), ((2, 12), .4, 4),
                                             ((31, 2), .4, 5), ((120, 90), .85, 6)])
def test_hip_operator_equals_explicit_system(gpu_device, shape, frac, seed):
    """The device A v, A^T u, right-hand side and row multiplicity against the reference's explicit sparse
    system (oracle.fda_system, neilpy.py:1180-1209): a wrong-by-one-equation stencil cannot hide inside the
    LSQR iteration tolerance of the solve tests.  Products agree to rounding (1e-13 relative)."""
    import ctypes as C
    import torch
    from neilpy_amd import _lib
    lib = _lib.load()
    rng = np.random.default_rng(seed)
    A = rng.normal(100.0, 20.0, size=shape)
    A[rng.random(shape) < frac] = np.nan
    a, b, nan_list, k = orc.fda_system(A, return_rows=True)
    m, n = shape
    v_nan = rng.normal(size=nan_list.size)
    V = np.zeros(m * n); V[nan_list] = v_nan
    U = rng.normal(size=m * n)                                   # one value per raster cell = per unique equation
    want_Av = a @ v_nan                                          # per kept (duplicated) equation row
    want_Atu = a.T @ U[k]
    dev = lambda x, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(x)).to(gpu_device).to(dt)   # noqa: E731
    A_d, V_d, U_d = dev(A), dev(V.reshape(shape)), dev(U.reshape(shape))
    rhs_d, Av_d, Atu_d = (torch.zeros(shape, dtype=torch.float64, device=gpu_device) for _ in range(3))
    cnt_d = torch.zeros(shape, dtype=torch.uint8, device=gpu_device)
    nbytes = lib.smrf_fda_workspace_bytes(m, n)
    ws = torch.empty(nbytes, dtype=torch.uint8, device=gpu_device)
    p = lambda t: C.c_void_p(t.data_ptr())                       # noqa: E731
    _lib.check(lib.smrf_fda_apply_f64(p(A_d), m, n, p(V_d), p(U_d), p(rhs_d), p(cnt_d), p(Av_d), p(Atu_d), p(ws), nbytes,
                                      C.c_void_p(torch.cuda.current_stream().cuda_stream)))
    cnt = cnt_d.cpu().numpy().ravel()
    assert np.array_equal(cnt, np.bincount(k, minlength=m * n))  # the reference keeps a row once per NaN entry
    rhs, Av, Atu = (t.cpu().numpy().ravel() for t in (rhs_d, Av_d, Atu_d))
    scale = max(1.0, float(np.abs(b).max(initial=0.0)))
    assert np.max(np.abs(rhs[k] - b), initial=0.0) <= 1e-13 * scale
    assert np.max(np.abs(Av[k] - want_Av), initial=0.0) <= 1e-13 * max(1.0, float(np.abs(want_Av).max(initial=0.0)))
    assert np.max(np.abs(Atu[nan_list] - want_Atu), initial=0.0) <= 1e-13 * max(1.0, float(np.abs(want_Atu).max(initial=0.0)))
