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


def test_split_rotation_equals_alfa_rot_step():
    """csrc/lsqr_core.h: round 5's single-device spring solver cuts alfa_rot_step in two - rho_step right after beta (the
    plane rotation needs only rhobar and beta) and alfa_rest_step after |v|^2 - so that t1 and 1 / rho exist one vector pass
    early.  Replayed here in Python floats (IEEE doubles, same operations in the same order): every scalar of both forms is
    bit-identical over a long random recurrence, including the b == 0 / a == 0 / |b| > |a| branches of sym_ortho."""
    import math
    import random

    def sgn(a):
        return 1.0 if a > 0 else (-1.0 if a < 0 else 0.0)

    def sym_ortho(a, b):
        if b == 0:
            return sgn(a), 0.0, abs(a)
        if a == 0:
            return 0.0, sgn(b), abs(b)
        if abs(b) > abs(a):
            tau = a / b
            sn = sgn(b) / math.sqrt(1 + tau * tau)
            cs = sn * tau
            return cs, sn, b / sn
        tau = b / a
        cs = sgn(a) / math.sqrt(1 + tau * tau)
        sn = cs * tau
        return cs, sn, a / cs

    def beta_step(s, sum_u):
        bt = math.sqrt(sum_u)
        s["beta"], s["beta_pos"] = bt, bt > 0
        if bt > 0:
            s["inv_beta"] = 1 / bt
            s["anorm"] = math.sqrt(s["anorm"] * s["anorm"] + s["alfa"] * s["alfa"] + bt * bt)
        else:
            s["inv_beta"] = 1.0

    def rest(s, cs, sn, rho, phi, theta):                 # the norm(x) estimate, lsqr.py:474-483
        delta = s["sn2"] * rho
        gambar = -s["cs2"] * rho
        rhs = phi - delta * s["z"]
        zbar = rhs / gambar
        s["xnorm"] = math.sqrt(s["xxnorm"] + zbar * zbar)
        gamma = math.sqrt(gambar * gambar + theta * theta)
        s["cs2"], s["sn2"], s["z"] = gambar / gamma, theta / gamma, rhs / gamma
        s["xxnorm"] = s["xxnorm"] + s["z"] * s["z"]

    def alfa_rot_step(s, sum_v):                          # round 4: everything after |v|^2
        if s["beta_pos"]:
            a = math.sqrt(sum_v)
            s["alfa"], s["inv_alfa"] = a, (1 / a if a > 0 else 1.0)
        cs, sn, rho = sym_ortho(s["rhobar"], s["beta"])
        theta = sn * s["alfa"]
        s["rhobar"] = -cs * s["alfa"]
        phi = cs * s["phibar"]
        s["phibar"] = sn * s["phibar"]
        s["tau"] = sn * phi
        s["t1"], s["t2"], s["inv_rho"] = phi / rho, -theta / rho, 1 / rho
        rest(s, cs, sn, rho, phi, theta)

    def rho_step(s):                                      # round 5, right after beta_step
        cs, sn, rho = sym_ortho(s["rhobar"], s["beta"])
        s["cs"], s["sn"], s["rho"] = cs, sn, rho
        s["phi"] = cs * s["phibar"]
        s["t1"], s["inv_rho"] = s["phi"] / rho, 1 / rho

    def alfa_rest_step(s, sum_v):                         # round 5, after |v|^2
        if s["beta_pos"]:
            a = math.sqrt(sum_v)
            s["alfa"], s["inv_alfa"] = a, (1 / a if a > 0 else 1.0)
        cs, sn, rho, phi = s["cs"], s["sn"], s["rho"], s["phi"]
        theta = sn * s["alfa"]
        s["rhobar"] = -cs * s["alfa"]
        s["phibar"] = sn * s["phibar"]
        s["tau"] = sn * phi
        s["t2"] = -theta / rho
        rest(s, cs, sn, rho, phi, theta)

    rng = random.Random(5)
    init = dict(alfa=0.7, beta=1.3, inv_alfa=1 / 0.7, inv_beta=1 / 1.3, rhobar=0.7, phibar=1.3, anorm=0.0, xxnorm=0.0, z=0.0,
                cs2=-1.0, sn2=0.0, t1=0.0, t2=0.0, inv_rho=0.0, xnorm=0.0, tau=0.0, beta_pos=True)
    a, b = dict(init), dict(init)
    keys = ["alfa", "beta", "rhobar", "phibar", "anorm", "xxnorm", "z", "cs2", "sn2", "t1", "t2", "inv_rho", "xnorm", "tau"]
    for it in range(400):
        su = 0.0 if it == 137 else rng.uniform(1e-6, 10.0) ** 2      # one beta == 0 step (sym_ortho's b == 0 branch)
        sv = rng.uniform(1e-6, 10.0) ** 2
        if it == 211:
            a["rhobar"] = b["rhobar"] = 0.0                           # the a == 0 branch
        beta_step(a, su)
        alfa_rot_step(a, sv)
        beta_step(b, su)
        rho_step(b)
        t1_early, ir_early = b["t1"], b["inv_rho"]                    # what atuxw_kernel reads before alfa exists
        alfa_rest_step(b, sv)
        assert (t1_early, ir_early) == (a["t1"], a["inv_rho"]), it
        assert all(a[k] == b[k] or (a[k] != a[k] and b[k] != b[k]) for k in keys), (it, [(k, a[k], b[k]) for k in keys if a[k] != b[k]])


def test_segment_rule_picks(tmp_path):
    """csrc/seg_rule.h (round 5, profiles/r05_segment_balance.md): how a marching launch is cut into row segments, compiled
    here with g++ and asked for the cases the note measures.  A large raster gets one full round of resident slots, rounded
    DOWN (33 strips on 512 slots: 15 rows of segments = 495 workgroups, never 16 = 528); a small one gets the segment count
    whose workgroups fill the 256 CUs k times exactly, without the old 4R minimum; the old rules stay reachable."""
    import os
    import subprocess
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "pick.cpp"
    src.write_text('#include <cstdio>\n#include "seg_rule.h"\nint main() { int a[8]; '
                   'while (std::scanf("%d %d %d %d %d %d %d %d", a, a + 1, a + 2, a + 3, a + 4, a + 5, a + 6, a + 7) == 8) '
                   'std::printf("%d\\n", smrf_pick_nseg(a[0], a[1], a[2], a[3], a[4], a[5], a[6], a[7])); return 0; }\n')
    exe = tmp_path / "pick"
    r = subprocess.run(["g++", "-O1", "-I", os.path.join(ROOT, "neilpy_amd", "csrc"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-1500:]
    #         rows   strips resident rounds warm batch min_seg rule   expected rows of segments
    cases = [((16384, 64, 4, 1, 30, 6, 60, 0), 16),          # the benchmark raster: 1024 workgroups = every slot
             ((16384, 64, 3, 1, 100, 4, 200, 0), 12),
             ((8193, 33, 2, 1, 80, 4, 160, 0), 15),          # 495 of 512 slots, not 528
             ((8193, 33, 2, 1, 80, 4, 160, 2), 15),
             ((8193, 33, 2, 1, 80, 4, 160, 1), 16),          # rounds 1-4: to nearest -> a second, nearly empty round
             ((30000, 24, 4, 1, 30, 6, 60, 0), 42),          # 1008 of 1024, not 43 x 24 = 1032
             ((1024, 4, 3, 1, 100, 4, 200, 0), 64),          # 256 workgroups, one per CU, 16-row segments
             ((1024, 4, 3, 1, 100, 4, 200, 2), 6),           # the 4R minimum: 24 workgroups
             ((4096, 16, 2, 1, 100, 4, 200, 0), 32),         # 512 workgroups: two per CU exactly (measured best: 128 rows)
             ((2048, 8, 2, 1, 100, 4, 200, 0), 32),          # one per CU, 64-row segments (measured best at R = 50)
             ((2048, 64, 2, 1, 100, 4, 200, 0), 8),          # a 1/8 band of the benchmark: two per CU, 256 rows
             ((100, 400, 3, 1, 100, 4, 200, 0), 1),          # more strips than slots: one row of segments
             ((3, 1, 4, 1, 10, 4, 32, 0), 1),                # fewer rows than a batch
             ((16384, 64, 4, 3, 6, 4, 32, 0), 48),           # the chained launches keep their three rounds on the benchmark raster
             ((16384, 64, 4, 3, 16, 4, 32, 0), 48),
             ((8193, 33, 4, 3, 16, 4, 32, 0), 31),           # ... and take ONE round where that is cheaper (measured -6 %)
             ((4096, 16, 4, 3, 16, 4, 32, 0), 64),
             ((1024, 4, 4, 3, 16, 4, 32, 0), 64),
             ((16384, 64, 4, 3, 16, 4, 32, 2), 48)]
    out = subprocess.run([str(exe)], input="\n".join(" ".join(str(v) for v in c) for c, _ in cases) + "\n", capture_output=True,
                         text=True, check=True).stdout.split()
    assert [int(v) for v in out] == [w for _, w in cases], list(zip(out, cases))
    # whatever the inputs: at least one, never more segments than batches of rows, and rule 0 never overfills the slots
    import random
    rnd = random.Random(5)
    rand = [(rnd.randint(1, 40000), rnd.randint(1, 300), rnd.randint(1, 8), 1, rnd.randint(2, 128), rnd.choice((2, 4, 6, 8)), 32, 0)
            for _ in range(2000)]
    out = subprocess.run([str(exe)], input="\n".join(" ".join(str(v) for v in c) for c in rand) + "\n", capture_output=True,
                         text=True, check=True).stdout.split()
    for c, v in zip(rand, out):
        nseg, (rows, strips, resident, _, _, batch, _, _) = int(v), c
        assert 1 <= nseg <= max(1, rows // batch + 1), (c, nseg)
        assert nseg == 1 or nseg * strips <= resident * 256, (c, nseg)


def test_power_sampler_reads_hwmon_files(tmp_path):
    """tools/gpu_power.py (bench.py's roofline.power): the sampler thread reads power1_input (uW) / freq1_input (Hz) of a
    hwmon directory and reports median / max watts and the median clock of a time span; without the files it says so
    (bench.py then leaves the field out) instead of raising."""
    import importlib.util
    import os
    import time
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("gpu_power", os.path.join(ROOT, "tools", "gpu_power.py"))
    gp = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gp)
    s = gp.Sampler.__new__(gp.Sampler)                       # (no torch device here: point it at a directory by hand)
    gp.Sampler.__init__(s, device_index=0, period=0.005)
    s.dir = str(tmp_path)
    assert not s.ok and s.start().stop().between(0, time.time() + 1) is None
    (tmp_path / "power1_input").write_text("1363000000\n")
    (tmp_path / "power1_cap").write_text("1400000000\n")
    (tmp_path / "freq1_input").write_text("2180000000\n")
    assert s.ok and s.cap_w() == 1400.0
    t0 = time.time()
    s.start()
    time.sleep(0.08)
    (tmp_path / "power1_input").write_text("1378000000\n")
    time.sleep(0.08)
    s.stop()
    r = s.between(t0, time.time())
    assert r["samples"] >= 4 and r["socket_w_max"] == 1378.0 and 1363.0 <= r["socket_w"] <= 1378.0
    assert abs(r["sclk_ghz"] - 2.18) < 1e-9
    assert s.between(t0 - 10, t0 - 5) is None


def test_xcd_tile_placement_is_a_permutation():
    """csrc/morph_ring.h ring_kernel (and lsqr_core.h lsqr_tile): workgroup `id` runs on XCD id % 8 and takes the tile
    t = xcd * (total / 8) + min(xcd, total % 8) + id / 8 of the strip-major tile list, so that an XCD owns a contiguous range
    of strips.  Restated here: every tile is taken exactly once for any grid, and an XCD's tiles are contiguous."""
    import random
    rnd = random.Random(3)
    for gx, gy in [(33, 31), (17, 60), (47, 21), (9, 1), (129, 7)] + [(rnd.randint(9, 200), rnd.randint(1, 80)) for _ in range(40)]:
        total = gx * gy
        seen = set()
        per_xcd = {}
        for i in range(total):
            xcd, slot, q, rem = i & 7, i >> 3, total >> 3, total & 7
            t = xcd * q + min(xcd, rem) + slot
            assert 0 <= t < total
            seen.add((t // gy, t % gy))
            per_xcd.setdefault(xcd, []).append(t)
        assert len(seen) == total, (gx, gy)
        for ts in per_xcd.values():
            assert ts == list(range(ts[0], ts[0] + len(ts)))
