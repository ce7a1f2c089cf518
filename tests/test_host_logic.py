"""Host-side logic that needs no GPU: affine arithmetic, disk footprints, synthetic inputs."""
import numpy as np
import pytest

from conftest import golden


def test_affine_matches_golden_transforms():
    from neilpy_amd.affine import from_origin
    g = golden("smrf_samp11_cs0p3.npz")
    t = from_origin(g["transform"][2], g["transform"][5], 0.3, 0.3)
    assert np.array_equal(np.array(tuple(t)[:6]), g["transform"])
    inv = ~t
    col, row = inv * (np.array([g["transform"][2] + 0.3 * 7.5]), np.array([g["transform"][5] - 0.3 * 2.25]))
    assert np.floor(col[0]) == 7 and np.floor(row[0]) == 2
    assert tuple(t)[6:] == (0.0, 0.0, 1.0) and t[0] == 0.3 and t[4] == -0.3


def test_disk_matches_oracle_and_counts():
    import neilpy_amd
    from oracle import smrf_oracle as orc
    for r in (0, 1, 2, 3, 7, 18, 50):
        assert np.array_equal(neilpy_amd.disk(r), orc.disk(r))
    assert neilpy_amd.disk(1).tolist() == [[0, 1, 0], [1, 1, 1], [0, 1, 0]]


def test_synth_dem_band_equals_full():
    from neilpy_amd.synth import synth_dem
    full = synth_dem(300, seed=9, rows=700)
    for r0, r1 in ((0, 700), (0, 1), (255, 257), (100, 613)):
        assert np.array_equal(synth_dem(300, seed=9, rows=700, row_range=(r0, r1)), full[r0:r1])
    assert full.dtype == np.float32 and np.isfinite(full).all()
    assert not np.array_equal(full, synth_dem(300, seed=10, rows=700))


def test_progressive_filter_rejects_list_windows_like_reference():
    """the reference fails on a list of windows at neilpy.py:1661 (list * float); so do we"""
    import neilpy_amd
    with pytest.raises(TypeError):
        neilpy_amd.progressive_filter(np.zeros((4, 4), np.float32), [1, 2], cellsize=.5)


def test_bench_bare_multi_gpu_refuses_without_devices():
    """`python bench.py --gpus N` with no launcher on a host that shows fewer than N devices (64 here: more than any host
    has): exit 4, an error JSON on stderr, nothing on stdout - and the starting process never imports torch itself"""
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 4 and r.stdout.strip() == ""
    err = json.loads([ln for ln in r.stderr.splitlines() if ln.startswith("{")][-1])
    assert err["n_gpus"] == 64 and err["devices_visible"] < 64 and "needs 64 visible HIP devices" in err["error"]
    # a launcher that started another number of ranks than --gpus says is refused the same way
    env2 = dict(env, WORLD_SIZE="3", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True,
                       timeout=900, env=env2)
    assert r.returncode == 4 and "started 3 ranks" in r.stderr
