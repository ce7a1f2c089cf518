"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's goldens.

Bit-exact for masks, when_dropped, opened surfaces and create_dem grids; float64 DTM / inpaint
within 1e-7 of the reference (north-star tolerance 1e-5) with LSQR's istop/itn equal.
"""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SAMPLES, free_port, golden, load_sample, switch, unpack, zmin_from_centi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nz(gpu_device):
    import neilpy_amd
    neilpy_amd.load_library()
    return neilpy_amd


@pytest.fixture(scope="module")
def orc():
    from oracle import smrf_oracle
    return smrf_oracle


def rand_dem(rng, shape, dtype):
    base = rng.normal(0, 1, shape).cumsum(0).cumsum(1) * 0.05 + 200
    spikes = (rng.random(shape) < 0.05) * rng.uniform(1, 25, shape)
    return (base + spikes).astype(dtype)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("impl", [1, 2])
def test_disk_filter_vs_oracle(nz, orc, dtype, impl):
    rng = np.random.default_rng(11)
    shapes = [(70, 333), (129, 64), (5, 300), (300, 7), (33, 257)]
    radii = [1, 2, 3, 5, 8, 13, 18, 31, 50, 64]
    for shape in shapes:
        Z = rand_dem(rng, shape, dtype)
        for r in radii:
            if impl == 2 and r > 18 and shape[0] * shape[1] > 20000:
                continue
            if r >= 4 * min(shape):
                # scipy's reflect offsets go out of bounds there (NI_InitFilterOffsets maps offsets
                # that are a multiple of 2*len below -2*len to index -1): the reference's result is
                # uninitialised memory, nothing to compare with.  Covered by ring-vs-direct below.
                a, b = nz.erosion(Z, radius=r, impl=1), nz.erosion(Z, radius=r, impl=2)
                assert np.array_equal(a, b), (shape, r)
                continue
            fp = orc.disk(r)
            e = nz.erosion(Z, radius=r, impl=impl)
            assert e.dtype == dtype
            assert np.array_equal(e, orc.erosion(Z, fp)), (shape, r, "erosion")
            d = nz.dilation(Z, radius=r, impl=impl)
            assert np.array_equal(d, orc.dilation(Z, fp)), (shape, r, "dilation")


def test_ring_equals_direct_all_radii(nz):
    """every ring instantiation (radius 1..64) against the independent direct kernel"""
    rng = np.random.default_rng(5)
    for dtype in (np.float32, np.float64):
        Z = rand_dem(rng, (150, 300), dtype)
        for r in range(1, 65):
            for fn in (nz.erosion, nz.dilation):
                a = fn(Z, radius=r, impl=1)
                b = fn(Z, radius=r, impl=2)
                assert np.array_equal(a, b), (dtype, r, fn.__name__)


def test_radius_beyond_ring_and_zero(nz, orc):
    rng = np.random.default_rng(6)
    Z = rand_dem(rng, (40, 90), np.float32)
    for r in (0, 65, 100):
        assert np.array_equal(nz.opening(Z, orc.disk(r)), orc.opening(Z, orc.disk(r)))
    with pytest.raises(NotImplementedError):
        nz.opening(Z, np.ones((3, 3), np.uint8))


PF = golden("progressive_filter.npz")


@pytest.mark.parametrize("impl", [0, 2])
@pytest.mark.parametrize("tag", [str(c) for c in PF["cases"]])
def test_progressive_filter_golden(nz, tag, impl):
    Z = PF[tag + "_Z"]
    windows = PF[tag + "_windows"]
    cellsize, slope = PF[tag + "_params"]
    if cellsize == int(cellsize):
        cellsize = int(cellsize)
    Z0 = Z.copy()
    mask, wd = nz.progressive_filter(Z, windows, cellsize, slope, return_when_dropped=True, impl=impl)
    assert np.array_equal(Z, Z0, equal_nan=True)
    assert mask.dtype == bool and wd.dtype == np.uint8
    assert np.array_equal(mask, unpack(PF[tag + "_mask_bits"], Z.shape))
    assert np.array_equal(wd, PF[tag + "_when_dropped"])
    assert np.array_equal(nz.progressive_filter(Z, windows, cellsize, slope, impl=impl), mask)
    last = Z
    for w in windows:
        last = nz.opening(last, nz.disk(int(w)), impl=impl)
    assert np.array_equal(last, PF[tag + "_opened_last"], equal_nan=True)


def test_progressive_filter_torch_in_torch_out(nz, gpu_device):
    import torch
    Z = PF["big_f32_w18_Z"]
    t = torch.from_numpy(Z).to(gpu_device)
    m = nz.progressive_filter(t, np.arange(1, 19), 1, .15)
    assert m.is_cuda and m.dtype == torch.bool
    assert np.array_equal(m.cpu().numpy(), unpack(PF["big_f32_w18_mask_bits"], Z.shape))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_progressive_filter_synth_vs_oracle(nz, orc, dtype):
    Z = nz.synth_dem(512, seed=20240, dtype=dtype)[:384]
    windows = np.arange(1, 19)
    m, w = nz.progressive_filter(Z, windows, 1, .15, return_when_dropped=True)
    m2, w2 = orc.progressive_filter(Z, windows, 1, .15, return_when_dropped=True)
    assert np.array_equal(m, m2) and np.array_equal(w, w2)


def test_opening_properties_large(nz, gpu_device):
    """size-independent properties at a size the oracle cannot reach in seconds"""
    import torch
    Z = torch.from_numpy(nz.synth_dem(2048, seed=7)).to(gpu_device)
    for r in (3, 18, 50):
        o = nz.opening(Z, radius=r)
        assert bool((o <= Z).all())                           # anti-extensive
        assert torch.equal(nz.opening(o, radius=r), o)        # idempotent
        e = nz.erosion(Z, radius=r)
        assert bool((e <= o).all())
        if r <= 18:
            assert torch.equal(nz.erosion(Z, radius=r, impl=2), e)


INP = golden("inpaint.npz")


@pytest.mark.parametrize("tag", [str(c) for c in INP["cases"]])
def test_inpaint_golden(nz, tag):
    A = INP[tag + "_in"]
    want = INP[tag + "_out"]
    istop, itn = INP[tag + "_lsqr"]
    A0 = A.copy()
    got = nz.inpaint_nans_by_springs(A)
    assert np.array_equal(A, A0, equal_nan=True)
    st = nz.last_stats["inpaint"]
    assert (st["istop"], st["itn"]) == (int(istop), int(itn))
    assert not np.isnan(got).any()
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-7)
    known = ~np.isnan(A)
    assert np.array_equal(got[known], A[known])
    A2 = A.copy()
    assert nz.inpaint_nans_by_springs(A2, inplace=True) is None
    assert np.array_equal(A2, got)


CD = golden("create_dem.npz")


@pytest.mark.parametrize("tag", [str(c) for c in CD["cases"]])
def test_create_dem_golden(nz, tag):
    kw = json.loads(str(CD[tag + "_kwargs_json"]))
    if kw.get("edges"):
        kw["edges"] = (CD[tag + "_xedges"], CD[tag + "_yedges"])
    I, t = nz.create_dem(CD[tag + "_x"], CD[tag + "_y"], CD[tag + "_z"], **kw)
    want = CD[tag + "_I"]
    assert I.shape == want.shape and I.dtype == np.float64
    assert np.array_equal(np.array(tuple(t)[:6]), CD[tag + "_transform"])
    if kw.get("inpaint"):
        np.testing.assert_allclose(I, want, rtol=0, atol=1e-7)
    else:
        assert np.array_equal(I, want, equal_nan=True)


CDS = golden("create_dem_samples.npz")


@pytest.mark.parametrize("tag", [str(c) for c in CDS["cases"]])
def test_create_dem_samples_noninteger_cellsize(nz, tag):
    """the reference's grids for samp52/54/71 at cellsize 0.3 / 0.7 (59-301 points per case land in another cell
    than (x - west) / cellsize would put them): bit-exact, transform equal"""
    x, y, z, _ = load_sample(tag.split("_")[0])
    I, t = nz.create_dem(x, y, z, cellsize=float(CDS[tag + "_cellsize"]), bin_type="min")
    assert np.array_equal(np.array(t[:6]), CDS[tag + "_transform"])
    assert np.array_equal(I, zmin_from_centi(CDS[tag + "_I_centi"]), equal_nan=True)


EDG = golden("edges.npz")


@pytest.mark.parametrize("tag", [str(c) for c in EDG["cases"]])
def test_create_dem_on_edges_from_IT(nz, tag):
    """create_dem(x, y, z, edges=edges_from_IT(I, t)) - a second gridding onto the first one's cells (neilpy.py:1095-1102,
    :1125-1132): the reference's grid and transform, bit-exact; the grid is also the one create_dem built without edges"""
    x, y, z, _ = load_sample("samp11")
    cs = {"cs1": 1, "cs0p3": .3, "cs2p5": 2.5}[tag]
    I, t = nz.create_dem(x, y, z, cellsize=cs, bin_type="min")
    xe, ye = nz.edges_from_IT(I, t)
    assert np.array_equal(xe, EDG[tag + "_xedges"]) and np.array_equal(ye, EDG[tag + "_yedges"])
    I2, t2 = nz.create_dem(x, y, z, bin_type="min", edges=(xe, ye))
    assert np.array_equal(np.array(tuple(t2)[:6]), EDG[tag + "_roundtrip_transform"])
    assert np.array_equal(I2, zmin_from_centi(EDG[tag + "_roundtrip_I_centi"]), equal_nan=True)
    assert np.array_equal(I2, I, equal_nan=True)


def test_create_dem_errors(nz):
    with pytest.raises(ValueError, match="This type not supported."):
        nz.create_dem(np.array([0., 1.]), np.array([0., 1.]), np.array([0., 1.]), bin_type="mean")
    with pytest.raises(ValueError):
        nz.create_dem(np.array([1.5, 3.0]), np.array([1.5, 2.5]), np.array([1.0, 2.0]),
                      edges=(np.arange(0.0, 4.0), np.arange(3.0, -1.0, -1.0)))


META = json.load(open(os.path.join(GOLDEN, "meta.json")))


def check_smrf(nz, gold, x, y, z, kw, full, stride):
    Zpro, t, obj, pts, extras = nz.smrf(x, y, z, return_extras=True, **kw)
    shape = tuple(gold["shape"])
    assert Zpro.shape == shape and Zpro.dtype == np.float64
    assert np.array_equal(np.array(tuple(t)[:6]), gold["transform"])
    st = nz.last_stats
    assert (st["inpaint1"]["istop"], st["inpaint1"]["itn"]) == tuple(gold["lsqr1"])
    assert (st["inpaint2"]["istop"], st["inpaint2"]["itn"]) == tuple(gold["lsqr2"])
    assert obj.dtype == bool and np.array_equal(obj, unpack(gold["object_cells_bits"], shape))
    assert np.array_equal(extras["drop_raster"], gold["pf_when_dropped"])
    assert np.array_equal(np.asarray(pts), unpack(gold["is_object_point_bits"], pts.shape))
    assert np.array_equal(extras["when_dropped"], gold["when_dropped_pts"])
    if full:
        np.testing.assert_allclose(Zpro, gold["Zpro"], rtol=0, atol=1e-7)
    else:
        np.testing.assert_allclose(Zpro.ravel()[::stride], gold["Zpro_strided"], rtol=0, atol=1e-7)
    assert abs(float(Zpro.sum()) - float(gold["Zpro_sum"][0])) < 1e-3
    tail = nz.last_stats["tail"]                        # device spline vs the reference's FITPACK values
    for key in ("elevation_values", "slope_values"):
        if full:
            np.testing.assert_allclose(tail[key], gold[key], rtol=0, atol=1e-7)
        else:
            np.testing.assert_allclose(tail[key][::stride], gold[key + "_strided"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(float(np.sum(extras["above_ground_height"])), float(gold["above_ground_height_sum"][0]),
                               rtol=0, atol=1e-4)
    return pts


def test_smrf_tensor_in_tensor_out(nz, gpu_device):
    """CUDA tensors in -> CUDA tensors out, equal to the NumPy path (which the goldens pin)."""
    import torch
    x, y, z, g = load_sample("samp24")
    want = nz.smrf(x, y, z, 1, 18, .15, .5, 1.25, return_extras=True)
    xd, yd, zd = (torch.from_numpy(v).to(gpu_device) for v in (x, y, z))
    got = nz.smrf(xd, yd, zd, 1, 18, .15, .5, 1.25, return_extras=True)
    assert got[0].is_cuda and got[0].dtype == torch.float64
    assert got[2].dtype == torch.bool and got[3].dtype == torch.bool and got[3].is_cuda
    assert np.array_equal(got[0].cpu().numpy(), want[0])
    assert tuple(got[1]) == tuple(want[1])
    assert np.array_equal(got[2].cpu().numpy(), want[2])
    assert np.array_equal(got[3].cpu().numpy(), want[3])
    for key in ("above_ground_height", "drop_raster", "when_dropped"):
        assert np.array_equal(got[4][key].cpu().numpy(), want[4][key]), key
    assert len(nz.smrf(xd, yd, zd)) == 4


@pytest.mark.parametrize("name", SAMPLES)
def test_smrf_samples_golden(nz, name):
    x, y, z, g = load_sample(name)
    gold = golden("smrf_%s.npz" % name)
    pts = check_smrf(nz, gold, x, y, z, META["smrf_kwargs"], name in META["full_dtm"], META["stride"])
    err = 100.0 * (1.0 - np.mean(pts == g))
    assert abs(err - META["anchors"][name]["total_error_pct"]) < 1e-9


@pytest.mark.parametrize("tag", ["cs0p5", "cs2", "cs0p3", "lowfill", "winlist"])
def test_smrf_samp11_variants(nz, tag):
    x, y, z, g = load_sample("samp11")
    gold = golden("smrf_samp11_%s.npz" % tag)
    kw = json.loads(str(gold["kwargs_json"]))
    if isinstance(kw["windows"], list):
        kw["windows"] = np.array(kw["windows"])
    check_smrf(nz, gold, x, y, z, kw, False, META["stride"])


def test_create_dem_stage_golden_all_samples(nz):
    for name in SAMPLES:
        x, y, z, g = load_sample(name)
        gold = golden("smrf_%s.npz" % name)
        I, t = nz.create_dem(x, y, z, cellsize=1, bin_type="min")
        assert np.array_equal(I, zmin_from_centi(gold["Zmin_centi"]), equal_nan=True)


def test_band_calls_match_full_raster(nz, gpu_device):
    """the row-band form of the C ABI (halo rows present, reflect only at true borders)"""
    import ctypes as C
    import torch
    from neilpy_amd import _lib
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(8)
    for dtype, sfx in ((np.float32, "f32"), (np.float64, "f64")):
        Zh = rand_dem(rng, (180, 300), dtype)
        Z = torch.from_numpy(Zh).to(gpu_device)
        m = Z.shape[0]
        for r in (1, 5, 18, 33):
            full_e = nz.erosion(Z, radius=r)
            full_o = nz.dilation(full_e, radius=r)
            for b0, b1 in ((0, 90), (90, 180), (40, 130)):
                lo, hi = max(0, b0 - 2 * r), min(m, b1 + 2 * r)
                q0, q1 = max(0, b0 - r), min(m, b1 + r)
                src = Z[lo:hi].contiguous()
                ero = torch.empty((q1 - q0, Z.shape[1]), dtype=Z.dtype, device=gpu_device)
                _lib.check(getattr(lib, "smrf_disk_filter_" + sfx)(
                    C.c_void_p(src.data_ptr()), C.c_void_p(ero.data_ptr()), m, Z.shape[1], Z.shape[1], lo, hi - lo,
                    q0, q1 - q0, r, 0, 0, 0, st))
                assert torch.equal(ero, full_e[q0:q1]), (sfx, r, b0)
                opened = torch.empty((b1 - b0, Z.shape[1]), dtype=Z.dtype, device=gpu_device)
                mask = torch.zeros((b1 - b0, Z.shape[1]), dtype=torch.uint8, device=gpu_device)
                when = torch.zeros_like(mask)
                last = Z[b0:b1].contiguous()
                _lib.check(getattr(lib, "smrf_pf_dilate_flag_" + sfx)(
                    C.c_void_p(ero.data_ptr()), C.c_void_p(last.data_ptr()), C.c_void_p(opened.data_ptr()),
                    C.c_void_p(mask.data_ptr()), C.c_void_p(when.data_ptr()), 0.7, 3, m, Z.shape[1], Z.shape[1],
                    q0, q1 - q0, b0, b1 - b0, r, 0, 0, st))
                assert torch.equal(opened, full_o[b0:b1]), (sfx, r, b0)
                want = ((last - opened).double() > 0.7)
                assert torch.equal(mask.bool(), want) and torch.equal(when, want.to(torch.uint8) * 3)
            # a band that lacks halo rows is refused, not silently reflected
            bad = Z[90:180].contiguous()
            out = torch.empty_like(bad)
            rc = getattr(lib, "smrf_disk_filter_" + sfx)(C.c_void_p(bad.data_ptr()), C.c_void_p(out.data_ptr()), m,
                                                          Z.shape[1], Z.shape[1], 90, 90, 90, 90, r, 0, 0, 0, st)
            assert rc == -1


def test_sharded_two_ranks_on_one_gpu(nz, tmp_path):
    """bench.py's N=2 path (row bands + halo exchange + HIP band kernels), gloo-staged halos, both
    ranks on cuda:0; the mask count must equal the single-device run."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    common = ["--size", "1024", "--windows", "12", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-pmc", "--no-secondary"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    # NO launcher and no WORLD_SIZE in the environment: bench.py starts its own two ranks (fresh children under
    # torch.distributed.run), relays rank 0's line and its exit code
    bare_env = {k: v for k, v in os.environ.items()
                if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--share-gpu"]
                         + common, capture_output=True, text=True, timeout=600, env=bare_env)
    assert two.returncode == 0, two.stderr[-2000:]
    assert len([l for l in two.stdout.splitlines() if l.startswith("{")]) == 1      # ONE line
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["n_gpus"] == 2 and j1["config"]["object_cells"] == j2["config"]["object_cells"]
    # the N = 1 line: median step, per-class split priced at the bytes each class moves (nothing above the peak)
    assert len(j1["step_ms"]) == 3 and abs(j1["ms_per_step"] - sorted(j1["step_ms"])[1]) < 2e-3
    cl = j1["roofline"]["classes"]
    assert sum(c["windows"] for c in cl.values()) == 12 and all(c["gbps"] < 8000.0 for c in cl.values())
    assert len(j1["roofline"]["window_ms"]) == 12
    # the N = 2 line: per-rank compute / exchange split and the exchange schedule
    sh = j2["config"]["sharding"]
    assert sh["bands"] == 2 and sh["backend"] == "gloo" and sh["exchanges_per_step"] == len(sh["groups"])
    assert sorted(r for g in sh["groups"] for r in g) == list(range(1, 13))
    assert len(j2["per_rank"]) == 2
    for pr in j2["per_rank"]:
        assert pr["exchanges"] == sh["exchanges_per_step"] and pr["exchange_ms"] > 0 and pr["compute_ms"] > 0
        assert pr["exchange_bytes_sent"] == sum(sh["halo_rows_per_exchange"]) * 1024 * 4


def test_bench_bare_multi_gpu_preflight(nz):
    """`python bench.py --gpus N` on a host with fewer than N devices: an error JSON on stderr, exit 4, no line, before
    any rank is started (the count is made in a child process; the starting process never touches the GPU)"""
    import json
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    want = torch.cuda.device_count() + 1
    bare_env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--size", "1024", "--windows", "12"],
                       capture_output=True, text=True, timeout=600, env=bare_env)
    assert r.returncode == 4 and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    err = json.loads([l for l in r.stderr.splitlines() if l.startswith("{")][-1])
    assert err["n_gpus"] == want and err["devices_visible"] == want - 1 and "needs" in err["error"]


def test_bench_two_ranks_nccl_needs_two_devices(nz):
    """the first real RCCL run of bench.py's N = 2 path: only where two GPUs are visible (never on the one-GPU box)"""
    import json
    import subprocess
    import sys
    import torch
    from conftest import ROOT
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible: RCCL halo exchange needs two devices")
    common = ["--size", "4096", "--windows", "18", "--steps", "3", "--warmup", "1", "--no-cpu", "--no-pmc", "--no-secondary"]
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common,
                         capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2"] + common, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-2000:]
    j1 = json.loads([l for l in one.stdout.splitlines() if l.startswith("{")][-1])
    j2 = json.loads([l for l in two.stdout.splitlines() if l.startswith("{")][-1])
    assert j2["config"]["sharding"]["backend"] == "nccl"
    assert j1["config"]["object_cells"] == j2["config"]["object_cells"]


def test_spline_matches_scipy(nz, gpu_device):
    """device bicubic spline (solve + ev, clamping outside the knot range) vs RectBivariateSpline"""
    import ctypes as C
    import torch
    from scipy import interpolate
    from neilpy_amd import _lib, spline
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rng = np.random.default_rng(12)
    for rows, cols in ((4, 4), (5, 9), (37, 23), (300, 211)):
        Z = rng.normal(100, 5, (rows, cols))
        f = interpolate.RectBivariateSpline(np.arange(.5, rows + .5), np.arange(.5, cols + .5), Z)
        pr = rng.uniform(-1.0, rows + 1.0, 5000)
        pc = rng.uniform(-1.0, cols + 1.0, 5000)
        pr[:4] = [0.5, rows - 0.5, 0.0, rows]           # knot-range ends and beyond
        pc[:4] = [0.5, cols - 0.5, cols, 0.0]
        pr[4:4 + min(rows, 50)] = np.arange(.5, rows + .5)[:50]   # exactly on data sites / knots
        want = f.ev(pr, pc)
        tx, lur = spline.axis_factors(rows)
        ty, luc = spline.axis_factors(cols)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
        coef, txd, tyd, lurd, lucd, prd, pcd = dev(Z), dev(tx), dev(ty), dev(lur), dev(luc), dev(pr), dev(pc)
        out = torch.empty(pr.size, dtype=torch.float64, device=gpu_device)
        p = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.smrf_spline_solve_f64(p(coef), rows, cols, p(lurd), p(lucd), st))
        np.testing.assert_allclose(coef.cpu().numpy().ravel(), f.tck[2], rtol=0, atol=1e-10)
        _lib.check(lib.smrf_spline_eval_f64(p(coef), rows, cols, p(txd), p(tyd), p(prd), p(pcd), pr.size, p(out), st))
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=0, atol=1e-9)
    rc = lib.smrf_spline_solve_f64(p(coef), 3, 9, p(lurd), p(lucd), st)
    assert rc == -1                                       # fewer than 4 sites: refused like SciPy refuses


def test_chunked_spline_solve_equals_the_sequential_one(nz, gpu_device):
    """smrf_spline_solve_ws_f64 (lines cut into chunks with a 64-entry warm-up, what smrf() uses) against the line-by-line
    smrf_spline_solve_f64: rasters large enough for several chunks per axis, rough data; the cut leaves 0.268^64 of the
    unknown state, so the two agree to rounding (and to SciPy's coefficients within the same 1e-10 as the sequential form)."""
    import ctypes as C
    import torch
    from scipy import interpolate
    from neilpy_amd import _lib, spline
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    rng = np.random.default_rng(5)
    for rows, cols in ((4, 4), (9, 300), (700, 513), (2100, 1500)):
        Z = rng.normal(100, 30, (rows, cols)) + rng.choice([0.0, 500.0], (rows, cols), p=[.97, .03])
        lur, luc = spline.axis_factors(rows)[1], spline.axis_factors(cols)[1]
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
        lurd, lucd = dev(lur), dev(luc)
        seq, chk, scratch = dev(Z), dev(Z), torch.full((rows, cols), float("nan"), dtype=torch.float64, device=gpu_device)
        _lib.check(lib.smrf_spline_solve_f64(p(seq), rows, cols, p(lurd), p(lucd), st))
        _lib.check(lib.smrf_spline_solve_ws_f64(p(chk), p(scratch), rows, cols, p(lurd), p(lucd), st))
        a, b = seq.cpu().numpy(), chk.cpu().numpy()
        assert np.isfinite(b).all()
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max(), (rows, cols, np.abs(a - b).max())
        if rows * cols <= 400_000:
            f = interpolate.RectBivariateSpline(np.arange(.5, rows + .5), np.arange(.5, cols + .5), Z)
            np.testing.assert_allclose(b.ravel(), f.tck[2], rtol=0, atol=1e-9)
    assert lib.smrf_spline_solve_ws_f64(p(chk), p(scratch), 3, 9, p(lurd), p(lucd), st) == -1


def test_smrf_pandas_series_in(nz):
    import pandas as pd
    x, y, z, g = load_sample("samp24")
    df = pd.DataFrame({"x": x, "y": y, "z": z})
    out = nz.smrf(df.x, df.y, df.z, 1, 18, .15, .5, 1.25)          # positional, as the notebooks call it
    gold = golden("smrf_samp24.npz")
    assert isinstance(out[3], pd.Series) and out[3].index.equals(df.index)
    assert np.array_equal(out[3].values, unpack(gold["is_object_point_bits"], (len(x),)))
    assert isinstance(out[0], np.ndarray) and isinstance(out[2], np.ndarray)


@pytest.mark.parametrize("world", [1, 2, 3])
def test_sharded_stages_rehearsal(nz, world):
    """create_dem band -> sharded LSQR -> sharded progressive_filter with the HIP band kernels on
    1, 2 and 3 ranks sharing this GPU (gloo-staged halos) against samp11's goldens"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    script = os.path.join(ROOT, "tools", "sharded_rehearsal.py")
    if world == 1:
        cmd = [sys.executable, script, "--sample", "samp11"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, "--sample", "samp11",
               "--backend", "gloo", "--share-gpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["create_dem_band_ok"] and j["lsqr_itn_ok"] and j["progressive_filter_ok"], j
    assert j["inpaint_max_abs_err"] < 1e-7, j


@pytest.mark.parametrize("world,points", [(1, "replicated"), (3, "replicated"), (3, "sharded")])
def test_smrf_sharded_rehearsal(nz, world, points):
    """the whole smrf() over row bands (neilpy_amd.sharded.smrf_sharded) on 1 and 3 ranks sharing this
    GPU (gloo-staged halos and gathers) against samp11's goldens: object raster and point flags bit-exact,
    both LSQR solves stop at the reference's iteration, DTM within 1e-7"""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    script = os.path.join(ROOT, "tools", "smrf_sharded_rehearsal.py")
    if world == 1:
        cmd = [sys.executable, script, "--sample", "samp11"]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
               "--master-addr", "127.0.0.1", "--master-port", str(free_port()), script, "--sample", "samp11",
               "--backend", "gloo", "--share-gpu", "--points", points]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    j = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert j["object_cells_ok"] and j["is_object_point_ok"] and j["transform_ok"] and j["lsqr_itn_ok"], j
    assert j["dtm_err"] < 1e-7 and j["object_points"] == 16384, j


def test_integration_md_stub_runs(nz, orc):
    """The ctypes binding shown in INTEGRATION.md section 2 is executed as written (only the library path is
    made absolute) and must equal the oracle."""
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    stub = [b for b in blocks if "ctypes.CDLL" in b]
    assert len(stub) == 1
    src = stub[0].replace('"libsmrf_hip.so"', repr(nz.LIB_PATH))
    ns = {}
    exec(compile(src, "INTEGRATION.md", "exec"), ns)
    rng = np.random.default_rng(3)
    Z = rand_dem(rng, (120, 150), np.float32)
    win = np.arange(1, 9)
    m, w = ns["progressive_filter"](Z, win, 1, .15, return_when_dropped=True)
    m2, w2 = orc.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
    assert m.dtype == bool and np.array_equal(m, m2) and np.array_equal(w, w2)
    assert np.array_equal(ns["progressive_filter"](Z.astype(np.float64), win), m2)


@pytest.mark.parametrize("route", ["default", "chain0", "fused0", "direct"])
def test_flag_threshold_float_boundaries(nz, orc, monkeypatch, route):
    """The flag step compares the raster dtype's difference with the float64 threshold in float64 (neilpy.py:1671 under
    NumPy 2).  The fp32 kernels compare against the largest float <= threshold instead (smrf_float_below): spikes whose
    height is the float just below / at / just above thresholds that are and are not float32 numbers must be flagged
    exactly as the oracle flags them, on every route a window can take."""
    for name in ("SMRF_FUSED", "SMRF_CHAIN"):
        switch(monkeypatch, name, None)
    if route == "chain0":
        switch(monkeypatch, "SMRF_CHAIN", "0")
    if route == "fused0":
        switch(monkeypatch, "SMRF_FUSED", "0")
    impl = 2 if route == "direct" else 0
    for slope in (0.3, 0.75, 0.1, 1e-3, 7.0):              # 0.75 and 7.0 are float32 numbers, the others are not
        for window in (1, 2, 4, 9, 11):
            thr = slope * (window * 1)
            f = np.float32(thr)
            hs = [np.nextafter(np.nextafter(f, np.float32(-np.inf)), np.float32(-np.inf)), np.nextafter(f, np.float32(-np.inf)), f,
                  np.nextafter(f, np.float32(np.inf)), np.nextafter(np.nextafter(f, np.float32(np.inf)), np.float32(np.inf))]
            Z = np.zeros((96, 300), dtype=np.float32)
            for k, h in enumerate(hs):
                Z[20 + 12 * k, 40 + 50 * k] = h                # isolated spikes: the opening removes them, diff = h exactly
            win = np.array([window])
            got = nz.progressive_filter(Z, win, 1, slope, impl=impl)
            want = orc.progressive_filter(Z, win, 1, slope)
            assert np.array_equal(got, want), (route, slope, window)
            assert int(want.sum()) == sum(1 for h in hs if float(h) > thr), (slope, window)
            Zd = Z.astype(np.float64)
            assert np.array_equal(nz.progressive_filter(Zd, win, 1, slope, impl=impl), orc.progressive_filter(Zd, win, 1, slope))


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_nan_anywhere_is_found_by_the_first_launch(nz, orc, dtype):
    """progressive_filter finds out about NaNs itself: when the call starts with a chained launch that launch carries
    the scan (csrc/morph.hip), runs as if there were none and the call starts over on the NaN-aware kernels if it met
    one.  A single NaN in any corner, on any edge or in the middle of a 3-strip raster must give the oracle's result."""
    rng = np.random.default_rng(31)
    shape = (300, 700)
    base = rand_dem(rng, shape, dtype)
    win = np.arange(1, 6)
    spots = [(0, 0), (0, 699), (299, 0), (299, 699), (0, 350), (299, 351), (150, 0), (151, 699), (150, 256), (37, 511)]
    for r, c in spots:
        Z = base.copy()
        Z[r, c] = np.nan
        m, w = nz.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
        m2, w2 = orc.progressive_filter(Z, win, 1, .15, return_when_dropped=True)
        assert np.array_equal(m, m2) and np.array_equal(w, w2), (r, c)
    m, w = nz.progressive_filter(base, win, 1, .15, return_when_dropped=True)      # and without one
    m2, w2 = orc.progressive_filter(base, win, 1, .15, return_when_dropped=True)
    assert np.array_equal(m, m2) and np.array_equal(w, w2)


def test_sharded_entry_points_over_one_rank_rccl(nz):
    """the sharded entry points over a REAL RCCL process group of one rank (tools/rccl_single_rank_check.py): process-group
    creation, barrier, the all-reduces of the three stages, all_to_all_single, and the halo exchange's own primitive - a
    batch_isend_irecv of a raster's row block - with this rank as its own neighbour; every result equal to the single-device
    one.  (Two ranks on one device are refused by RCCL: the neighbour exchange between DIFFERENT ranks needs two GPUs.)"""
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()), RANK="0", WORLD_SIZE="1")
    env.pop("GRAFT_REPO_ROOT", None)                 # the tool finds the repository from its own path, not from the environment
    env.pop("PYTHONPATH", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_single_rank_check.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd="/tmp")
    assert r.returncode == 0, r.stderr[-2000:]
    out = r.stdout
    assert "backend nccl" in out
    assert "objects 74932 74932 True" in out
    assert "max |band - single device| 0.0" in out
    assert "create_dem_sharded equal True True" in out
    assert "self send/recv of a row block over RCCL equal True" in out


def test_band_lsqr_on_one_rank_is_the_single_device_solve(nz, gpu_device):
    """round 5: the row-band phases (smrf_springs_band_phase: PH_AV, PH_BETA_RHO, PH_ATUXW, PH_ALFA_TESTS) run the single-device
    solver's own vector kernels, so a band that is the whole raster gives the same bits, the same istop and itn - whether the
    solve stops at an even or an odd iteration (the odd ones owe x one step, added by the scatter)"""
    import torch
    from neilpy_amd import sharded
    seen = set()
    for n, holes, seed in ((257, 0.6, 1), (300, 0.3, 2), (513, 0.8, 3), (199, 0.5, 4), (401, 0.1, 5)):
        g = torch.Generator(device="cuda").manual_seed(seed)
        Z = torch.from_numpy(nz.synth_dem(n, seed=seed).astype(np.float64)).to(gpu_device)
        Z[torch.rand((n, n), device="cuda", generator=g) < holes] = float("nan")
        ref = nz.inpaint_nans_by_springs(Z)
        st = dict(nz.last_stats["inpaint"])
        A = Z.clone()
        istop, itn, nunk = sharded.inpaint_nans_by_springs_sharded(A, n, rank=0, world_size=1)
        assert (istop, itn, nunk) == (st["istop"], st["itn"], st["n_unknown"])
        assert torch.equal(A, ref), (n, holes, float((A - ref).abs().max()))
        seen.add(itn & 1)
    assert seen == {0, 1}, "the cases should stop at both parities"
