#!/usr/bin/env python3
"""Headline benchmark: Mcells/s through SMRF progressive_filter on a 16384^2 fp32 DEM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 16384] [--windows 50]

A "step" is one call of the public drop-in ``neilpy_amd.progressive_filter(Z, windows, cellsize,
slope_threshold)`` (all windows: erosion + dilation + flagging per window, plus the call's own NaN
scan and bool mask) over the synthetic DEM ``synth_dem(n, seed=20240)`` already resident in HBM.  With N > 1
(one rank per GPU) the DEM's rows are split into N bands and groups of consecutive windows exchange their halo rows
with the neighbouring ranks over RCCL (neilpy_amd/sharded.py): the problem size is fixed, so ``scaling`` is "strong".
Rank 0 prints ONE JSON line.  ``python bench.py --gpus N`` works both ways: started by ``torch.distributed.run`` (WORLD_SIZE
in the environment) it is one of the N ranks; started bare it first counts the visible devices in a child process (an
error JSON on stderr and exit 4 when there are fewer than N), then starts the N ranks itself as fresh child processes
under ``torch.distributed.run`` on 127.0.0.1, relays the line and leaves with their exit code - the starting process
never imports torch or touches a GPU.

Timing: W warm-up steps, then K steps between barrier + synchronize on both sides (``ms_per_step_mean`` = that wall
time / K, max over ranks); every step also sits between two events on the launch stream, ``step_ms`` = the K
device times (max over ranks per step), ``ms_per_step`` their MEDIAN and ``value`` = cells / median.

``roofline`` (dominant kernels: smrf::ring_kernel, two passes per window, and the fused small-disk launches):
  achieved      algorithmic bytes per pass ((5s + 2) / 2 B/cell, a pass = half a window) / median device
                time per pass; frac = achieved / 8 TB/s
  traffic       HBM bytes per pass from the hardware counters (two rocprofv3 --pmc child runs after the timed region);
                frac_traffic = traffic / time per pass / peak - the physical rate; it is below `frac` because the
                fused launches move 10 B/cell per window where the convention credits 22
  classes       one extra step through smrf_progressive_filter_timed_* (an event per window): windows grouped by how
                they ran, each class priced at the bytes it really moves (fused 2s + 2, chain of k windows (2s + 2k) / k,
                two-pass 5s + 2 B/cell/window), so no GB/s in this block can exceed the peak; with the counters
                (third rocprofv3 --pmc child run: GRBM_GUI_ACTIVE SQ_INSTS_VALU) every class also carries
                valu_inst_per_cell, shader_clock_ghz and valu_frac = VALU instructions x 4.1 cycles / (1024 SIMDs x
                clock x the class's time): the fraction of the VALU ISSUE bound, the on-chip bound the large disks sit on
  secondary_bound  the two-pass class's pair: frac of the VALU issue bound beside its fraction of 8 TB/s
  power         package power (W) and shader clock while the step runs back to back for 2.5 s after the timed region,
                beside the power cap (amdgpu hwmon files, tools/gpu_power.py): the step runs AT the cap - the firmware
                lowers the clock to hold it, which is why the issue fraction does not turn into time
``secondary`` (N = 1, after the headline; SURVEY 8d's secondary metric): full device smrf() on 20 M synthetic points
  (per-stage ms, points/s, LSQR ms and GB/s per iteration) and the fp64 progressive_filter 8192^2 windows 1..18 rate
  (fp64 is the dtype the reference's smrf runs in, neilpy.py:1136).
``cpu_baseline``: the NumPy/SciPy oracle (the same scipy.ndimage primitive the reference reaches through skimage),
  single thread, on a bounded crop of the same DEM with the same windows.
N > 1: ``per_rank`` carries every rank's compute_ms / exchange_ms (events around each halo exchange) / bytes sent;
  the run refuses to start on a backend other than nccl (= RCCL) unless --backend gloo is given.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", dest="n", type=int, default=16384, help="DEM is size x size cells")
    ap.add_argument("--windows", type=int, default=50, help="radii 1..windows")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--cpu-crop", type=int, default=192, help="crop edge for the CPU baseline (0 = skip)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the 2.5 s package-power leg after the timed region (roofline.power)")
    ap.add_argument("--no-pmc", action="store_true", help="do not collect roofline.traffic / the VALU counters live (three rocprofv3 "
                    "--pmc child runs, ~1 min); quote profiles/pmc_summary.json instead")
    ap.add_argument("--no-secondary", action="store_true", help="skip the smrf() / fp64 secondary block")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo stages halos through the host: for rehearsing N>1 ranks on one GPU")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal on a one-GPU box)")
    ap.add_argument("--launch-timeout", type=int, default=1500, help="bare --gpus N: seconds the self-started ranks may take")
    return ap.parse_args()


def cpu_baseline(Z_crop, windows, cellsize, slope):
    """The oracle timed on the host: single process, single thread (how the reference runs)."""
    from oracle import smrf_oracle as orc
    t0 = time.perf_counter()
    orc.progressive_filter(Z_crop, windows, cellsize, slope)
    dt = time.perf_counter() - t0
    out = dict(value=Z_crop.size / dt / 1e6, unit="Mcells/s", cores=1, kind="port", seconds=round(dt, 3),
               sample="a %dx%d CROP (top-left) of the same DEM, windows 1..%d: oracle/smrf_oracle.py progressive_filter "
                      "(scipy.ndimage grey_erosion/grey_dilation, disk footprints, 1 thread); SURVEY 8d's full procedure at "
                      "n = 2048 is profiles/r02_cpu_baseline_2048.json (0.0025 Mcells/s, the same rate)"
                      % (Z_crop.shape[0], Z_crop.shape[1], len(windows)))
    # the same oracle on every host core at once (independent tiles; a child process: this one holds the GPU)
    try:
        import subprocess
        r = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--crop", str(Z_crop.shape[0]), "--windows",
                            str(len(windows))], cwd=ROOT, capture_output=True, text=True, timeout=600)
        out["all_cores"] = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001  (the single-thread figure above is the baseline; this one is extra)
        out["all_cores"] = {"error": repr(e)[:200]}
    return out


def power_leg(step, barrier, dev, seconds=2.5):
    """Package power and shader clock while the step runs back to back for ``seconds`` AFTER the timed region (the hwmon
    figure is a firmware average that lags a 0.7 s region): tools/gpu_power.py reads the amdgpu hwmon files of this device
    from a thread.  None where the files are not there.  profiles/r05_power_bound.md: every launch class of the step sits at
    1.33-1.37 kW of the 1.4 kW cap and the firmware lowers the shader clock to hold it - the bound under the VALU issue
    bound."""
    import importlib.util
    try:
        spec = importlib.util.spec_from_file_location("gpu_power", os.path.join(ROOT, "tools", "gpu_power.py"))
        gp = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(gp)
        smp = gp.Sampler(dev.index or 0, 0.01)
        if not smp.ok:
            return None
        smp.start()
        t0 = time.time()
        while time.time() - t0 < seconds:
            for _ in range(3):
                step()
            barrier()
        t1 = time.time()
        smp.stop()
        rec = smp.between(t0 + min(1.0, seconds / 2), t1)
        if rec is None:
            return None
        rec["cap_w"] = smp.cap_w()
        rec["frac_of_cap"] = rec["socket_w"] / rec["cap_w"] if rec["cap_w"] else None
        rec["source"] = ("amdgpu hwmon power1_input / power1_cap / freq1_input of this device, every 10 ms while the step ran back to "
                         "back for %.1f s after the timed region (first second dropped: the figure is a firmware average)" % seconds)
        return rec
    except Exception as e:  # noqa: BLE001  (a measurement aid: the headline does not depend on it)
        return {"error": repr(e)[:200]}


def pmc_live(n, windows, dtype, timeout_s=150):
    """Hardware counters of this workload, measured now: three child runs of tools/pmc_traffic.py (one progressive_filter
    step + a calibration read) under ``rocprofv3 --pmc FETCH_SIZE``, ``--pmc WRITE_SIZE`` and ``--pmc GRBM_GUI_ACTIVE
    SQ_INSTS_VALU`` - separate passes, --kernel-trace only, as MI355X_MICROARCH's HBM section prescribes.
    Returns (traffic, valu): ``traffic`` = tools/profile_summary.pmc's record (HBM bytes per pass, reads scaled by the
    calibration kernel's known byte count); ``valu`` = per launch class the VALU wave-instructions, the GPU-active cycles
    (GRBM_GUI_ACTIVE is summed over the 8 XCDs) and the kernel time of the counter pass.  Either is None when rocprofv3
    is not there, a pass fails or times out: the caller then quotes the committed summary / leaves the fields out."""
    import csv
    import glob
    import importlib.util
    import re
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or dtype != "f32":
        return None, None
    spec = importlib.util.spec_from_file_location("profile_summary", os.path.join(ROOT, "tools", "profile_summary.py"))
    ps = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ps)
    tmp = tempfile.mkdtemp(prefix="smrf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    csvs = {}
    traffic, valu = None, None
    try:
        for tag, counters in (("FETCH_SIZE", ["FETCH_SIZE"]), ("WRITE_SIZE", ["WRITE_SIZE"]),
                              ("VALU", ["GRBM_GUI_ACTIVE", "SQ_INSTS_VALU"])):
            d = os.path.join(tmp, tag)
            r = subprocess.run([exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                                sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), "--size", str(n),
                                "--windows", str(windows)], cwd="/tmp", env=env, capture_output=True, text=True,
                               timeout=timeout_s)
            found = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode == 0 and found:
                csvs[tag] = found[0]
        if "FETCH_SIZE" in csvs and "WRITE_SIZE" in csvs:
            rec = ps.pmc(csvs["FETCH_SIZE"], csvs["WRITE_SIZE"], n, windows, None)
            traffic = rec if rec.get("fetch_calibration") else None
        if "VALU" in csvs:
            valu = {}
            seen = set()
            for row in csv.DictReader(open(csvs["VALU"])):
                k = row["Kernel_Name"]
                m = re.search(r"chain_kernel<\w+, \d+, \d+, (\d+), (\d+), (\d+), (\d+)", k)
                if m:
                    cls = "chain%d" % sum(1 for v in m.groups() if int(v))
                elif "fused_open_kernel" in k:
                    cls = "fused"
                elif "ring_kernel" in k:
                    cls = "two_pass"
                else:
                    continue
                c = valu.setdefault(cls, {"insts": 0.0, "cycles": 0.0, "ns": 0.0})
                if row["Counter_Name"] == "SQ_INSTS_VALU":
                    c["insts"] += float(row["Counter_Value"])
                elif row["Counter_Name"] == "GRBM_GUI_ACTIVE":
                    c["cycles"] += float(row["Counter_Value"]) / 8.0
                if row["Dispatch_Id"] not in seen:
                    seen.add(row["Dispatch_Id"])
                    c["ns"] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
            if not valu or any(c["insts"] <= 0 or c["cycles"] <= 0 or c["ns"] <= 0 for c in valu.values()):
                valu = None
    except Exception:  # noqa: BLE001  (a measurement aid: never fail the bench line over it)
        pass
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return traffic, valu


# cycles one wave64 min / max (v_min3_f32, v_max3_f32, v_min_f32 ...) occupies a SIMD's issue port on gfx950, measured at every
# occupancy from 2 to 8 waves per SIMD (profiles/r02_issue_rate_ubench.md; the guide's 2-cycle rate holds for fma / add / mul)
VALU_CYCLES_PER_MINMAX = 4.1
SIMDS = 1024                                               # 256 CUs x 4


def window_classes(Z, windows, thresholds, cells, elem, peak):
    """One more step with an event at every window boundary: device time per window and how each ran, grouped into
    classes priced at the HBM bytes that class really moves per cell and window."""
    from neilpy_amd import api, _lib
    timing = {}
    api._progressive_filter_device(Z, windows, thresholds, False, nan_aware=0, timing=timing)   # warm (first-call setup)
    api._progressive_filter_device(Z, windows, thresholds, False, nan_aware=0, timing=timing)
    ms, route = timing["window_ms"], timing["route"]
    # chains: route = ROUTE_CHAIN + position; the time of a chain is recorded on its first window, the other members read ~0
    classes = {}
    i = 0
    while i < len(ms):
        r = int(route[i])
        j, t = i + 1, float(ms[i])
        if r >= _lib.ROUTE_CHAIN:
            while j < len(ms) and int(route[j]) > _lib.ROUTE_CHAIN:
                t += float(ms[j])
                j += 1
            k = j - i
            name, bpw = "chain%d" % k, (2.0 * elem + 2.0 * k) / k
        elif r == _lib.ROUTE_FUSED:
            name, bpw = "fused", 2.0 * elem + 2.0
        elif r == _lib.ROUTE_COPY:
            name, bpw = "copy", 3.0 * elem + 2.0
        else:
            name, bpw = ("two_pass" if r == _lib.ROUTE_TWO_PASS else "direct"), 5.0 * elem + 2.0
        c = classes.setdefault(name, {"windows": 0, "ms": 0.0, "bytes_per_cell_window": bpw, "radii": []})
        c["windows"] += j - i
        c["ms"] += t
        c["radii"] += [int(w) for w in windows[i:j]]
        i = j
    for c in classes.values():
        c["gbps"] = c["bytes_per_cell_window"] * c["windows"] * cells / (c["ms"] * 1e-3) / 1e9
        c["frac"] = c["gbps"] / peak
        c["ms"] = round(c["ms"], 4)
        rr = c.pop("radii")
        c["radii"] = "%d..%d" % (min(rr), max(rr)) if rr == list(range(min(rr), max(rr) + 1)) else rr
    return classes, [round(float(v), 4) for v in ms]


def secondary(dev):
    """SURVEY 8d's secondary metric, timed after the headline on the same GPU (N = 1 only)."""
    import importlib.util
    import torch
    import neilpy_amd
    from neilpy_amd import api
    out = {}
    spec = importlib.util.spec_from_file_location("smrf_stages", os.path.join(ROOT, "tools", "smrf_stages.py"))
    st = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(st)
    out["smrf_20M"] = st.run(points=20_000_000, extent=8192.0, windows=18, cellsize=1.0, warm=True)
    out["smrf_20M"]["workload"] = ("neilpy_amd.smrf on synth_points(2e7, 8192.0, seed=20241), cellsize 1, windows 18, "
                                   "device-resident points in, device tensors out; the per-stage figures are each stage's SECOND "
                                   "run (a first run also loads kernels, queries occupancies and grows the allocator: "
                                   "create_dem 15 ms cold, 1.5 ms warm), smrf_total_ms one whole call after them")
    torch.cuda.empty_cache()
    n, nw = 8192, 18
    Z = torch.from_numpy(neilpy_amd.synth_dem(n, seed=20240, dtype=np.float64)).to(dev)
    win = np.arange(1, nw + 1)
    thr = .15 * (win * 1)
    for _ in range(2):
        api._progressive_filter_device(Z, win, thr, False, nan_aware=0)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    evs[0].record()
    for k in range(5):
        m, _w = api._progressive_filter_device(Z, win, thr, False, nan_aware=0)
        evs[k + 1].record()
    torch.cuda.synchronize()
    ms = float(np.median([evs[k].elapsed_time(evs[k + 1]) for k in range(5)]))
    alg = n * n * nw * (5 * 8 + 2)
    out["progressive_filter_f64_8192_w18"] = {
        "ms_per_step": ms, "Mcells_per_s": n * n / ms / 1e3, "achieved_gbps": alg / (ms * 1e-3) / 1e9,
        "frac": alg / (ms * 1e-3) / 1e9 / 8000.0, "algorithmic_bytes_per_cell_window": 42,
        "object_cells": int(m.sum().item()),
        "workload": "progressive_filter core on synth_dem(8192, seed=20240) float64, windows 1..18, median of 5"}
    return out


def fail_line(msg, code, **extra):
    """an error instead of a bench line: one JSON object on stderr (stdout carries bench lines only), non-zero exit"""
    print(json.dumps(dict({"error": msg}, **extra)), file=sys.stderr, flush=True)
    sys.exit(code)


def visible_devices():
    """HIP devices a fresh process of this interpreter sees, counted in a CHILD: the process that starts the ranks never
    imports torch and never touches the GPU (a parent that holds a HIP context beside N ranks is one more process on the
    card, and nothing that has initialised the GPU may exec another program on this pool).  None if the count fails."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                           text=True, timeout=900)
        return int(r.stdout.strip().splitlines()[-1])
    except Exception:  # noqa: BLE001
        return None


def self_launch(a, argv):
    """``python bench.py --gpus N`` with no launcher around it: start the N ranks ourselves - fresh child processes under
    ``torch.distributed.run`` (one per GPU, rendezvous on 127.0.0.1 at a free port), the same command line - relay rank
    0's line and leave with the launcher's exit code.  Nothing GPU-related happens in this process."""
    import socket
    import subprocess
    if not a.share_gpu:
        have = visible_devices()
        if have is None:
            fail_line("--gpus %d: could not count the visible HIP devices (is torch importable?)" % a.gpus, 4, n_gpus=a.gpus)
        if have < a.gpus:
            fail_line("--gpus %d needs %d visible HIP devices, this host shows %d (use --share-gpu --backend gloo to rehearse "
                      "the ranks on one device)" % (a.gpus, a.gpus, have), 4, n_gpus=a.gpus, devices_visible=have)
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")         # dmabuf IPC: what RCCL needs between the ranks on this pool
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    # the ranks run in their own session: if they hang past --launch-timeout the whole group (launcher + ranks) is ended,
    # not just the launcher - a rank left behind would keep the GPUs
    import signal
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, start_new_session=True)      # stderr passes through
    try:
        stdout, _ = proc.communicate(timeout=a.launch_timeout)
    except subprocess.TimeoutExpired:
        for sig in (signal.SIGTERM, signal.SIGKILL):
            try:
                os.killpg(proc.pid, sig)
            except ProcessLookupError:
                break
            try:
                proc.wait(timeout=20)
                break
            except subprocess.TimeoutExpired:
                continue
        fail_line("the %d ranks did not finish within %d s (--launch-timeout): process group ended" % (a.gpus, a.launch_timeout), 6,
                  n_gpus=a.gpus)
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    for ln in lines[-1:]:
        print(ln, flush=True)
    if proc.returncode == 0 and not lines:
        fail_line("the %d ranks ended without a bench line" % a.gpus, 5, n_gpus=a.gpus)
    sys.exit(proc.returncode)


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a, sys.argv[1:])                          # does not return
    import torch
    import torch.distributed as dist
    import neilpy_amd
    from neilpy_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if rank == 0:
            fail_line("--gpus %d but the launcher started %d ranks (WORLD_SIZE)" % (a.gpus, world), 4, n_gpus=a.gpus)
        sys.exit(4)
    if not torch.cuda.is_available():
        if rank == 0:
            fail_line("bench.py needs an MI355X: neilpy_amd has no CPU fallback", 4, n_gpus=a.gpus)
        sys.exit(4)
    if not a.share_gpu and local_rank >= torch.cuda.device_count():
        fail_line("rank %d has no device: %d visible for %d ranks (use --share-gpu --backend gloo to rehearse on one)"
                  % (rank, torch.cuda.device_count(), world), 4, n_gpus=a.gpus)
    if a.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")
        # a multi-GPU line measured over anything but RCCL is not the metric: refuse, loudly, on every rank
        if dist.get_backend() != "nccl" and a.backend != "gloo":
            if rank == 0:
                print(json.dumps({"error": "process group backend is %r: the N > 1 bench line is defined over nccl (= RCCL); "
                                           "pass --backend gloo only to rehearse (halos staged through the host)"
                                           % dist.get_backend()}), file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)

    n = a.n
    np_dtype = np.float32 if a.dtype == "f32" else np.float64
    windows = np.arange(1, a.windows + 1)
    cellsize, slope = 1, .15
    thresholds = slope * (windows * cellsize)
    b0, b1 = sharded.band_rows(n, world, rank)
    Z_host = neilpy_amd.synth_dem(n, seed=20240, dtype=np_dtype, row_range=(b0, b1))
    Z = torch.from_numpy(Z_host).to(dev)
    crop = None
    if rank == 0 and world == 1 and not a.no_cpu and a.cpu_crop > 0:   # the CPU figure belongs to the N = 1 line only
        c = min(a.cpu_crop, n, b1 - b0)
        crop = np.ascontiguousarray(Z_host[:c, :c])
    del Z_host

    state = {"profile": True}

    def step():
        if world == 1:
            # the drop-in entry itself (CUDA tensor in -> CUDA bool mask out), not an internal core
            return neilpy_amd.progressive_filter(Z, windows, cellsize, slope)
        return sharded.progressive_filter_sharded(Z, n, windows, thresholds, rank=rank, world_size=world, state=state)[0]

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(a.warmup):
        mask = step()
    barrier()
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
    xev = []                                              # N > 1: the exchange events of every timed step
    t0 = time.perf_counter()
    evs[0].record()
    for k in range(a.steps):
        mask = step()
        evs[k + 1].record()
        if world > 1:
            xev.append((state.get("exchange_events", []), state.get("exchange_bytes", [])))
    barrier()
    dt = time.perf_counter() - t0
    step_ms = np.array([evs[k].elapsed_time(evs[k + 1]) for k in range(a.steps)], dtype=np.float64)
    power = power_leg(step, barrier, dev) if world == 1 and not a.no_power else None
    n_obj = int(mask.sum().item())
    per_rank = None
    if world > 1:
        exch_ms = np.array([sum(e0.elapsed_time(e1) for e0, e1 in evl) for evl, _ in xev], dtype=np.float64)
        exch_bytes = float(sum(xev[-1][1])) if xev else 0.0
        cdev = dev if a.backend == "nccl" else "cpu"
        t = torch.tensor([dt, float(n_obj)], dtype=torch.float64, device=cdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        dt, n_obj = float(tmax[0]), int(t[1])
        mine = torch.tensor([float(np.median(step_ms)), float(np.median(exch_ms)), exch_bytes, float(len(xev[-1][0]))],
                            dtype=torch.float64, device=cdev)
        allr = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [{"rank": k, "step_ms": round(float(v[0]), 4), "exchange_ms": round(float(v[1]), 4),
                     "compute_ms": round(float(v[0] - v[1]), 4), "exchange_bytes_sent": int(v[2]), "exchanges": int(v[3])}
                    for k, v in enumerate(allr)]
        sm = torch.tensor(step_ms, dtype=torch.float64, device=cdev)
        dist.all_reduce(sm, op=dist.ReduceOp.MAX)         # a step is over when its slowest rank is
        step_ms = sm.cpu().numpy()

    # the workload's known answer (tests/test_gpu_fullsize.py pins it): a run, sharded or not, that flags other cells
    # is not a measurement.  Every rank holds the global count here, so every rank leaves with the same exit code.
    expected = {(16384, 50, "f32"): 51388194}.get((n, a.windows, a.dtype))
    if expected is not None and n_obj != expected:
        if rank == 0:
            print(json.dumps({"error": "object_cells %d != %d expected for this workload: the result is wrong, no bench "
                                       "line is printed" % (n_obj, expected), "n_gpus": world}), file=sys.stderr, flush=True)
        if world > 1:
            dist.destroy_process_group()
        sys.exit(2)

    # what a plain device copy of one plane reaches on this GPU (read + write), for context beside the 8 TB/s spec
    # (torch's copy_; what hand-written streaming kernels reach by access width, grid and plane size is
    # tools/ubench/stream_rate.hip -> profiles/r04_stream_rate_ubench.md)
    copy_gbps = None
    if rank == 0:
        tmp = torch.empty_like(Z)
        c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tmp.copy_(Z)
        c0.record()
        for _ in range(5):
            tmp.copy_(Z)
        c1.record()
        torch.cuda.synchronize()
        copy_gbps = 5 * 2 * Z.numel() * Z.element_size() / (c0.elapsed_time(c1) * 1e-3) / 1e9
        del tmp
    if rank == 0:
        cells = n * n
        elem = 4 if a.dtype == "f32" else 8
        peak = 8000.0
        ms_median = float(np.median(step_ms))
        ms_mean = dt / a.steps * 1e3
        # a "pass" is half a window (erosion, or dilation + flag): two ring launches per window, or one fused launch
        # that does both - the algorithmic bytes per window are SURVEY 8d's 5s + 2 either way
        launches = 2 * len(windows)
        alg_bytes_launch = cells / world * (5 * elem + 2) / 2.0  # per rank, per pass
        avg_launch_s = ms_median / 1e3 / launches
        achieved = alg_bytes_launch / avg_launch_s / 1e9
        classes, window_ms = (None, None)
        if world == 1:
            classes, window_ms = window_classes(Z, windows, thresholds, cells, elem, peak)
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_summary.json")
        live, valu = None, None
        if world == 1 and not a.no_pmc:
            # the timed region is over: release this process's planes first, the child runs need the same HBM
            del mask
            Z = None
            torch.cuda.empty_cache()
            live, valu = pmc_live(n, a.windows, a.dtype)
        if live is not None:
            traffic = live["hbm_bytes_per_launch"]
            traffic_source = ("live: two child runs of tools/pmc_traffic.py under rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                              "(--kernel-trace only), reads x%.3f (calibration kernel of known size), per pass"
                              % live["fetch_calibration"])
        elif os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("n") == n and rec.get("windows") == a.windows and rec.get("dtype") == a.dtype and world == 1:
                    traffic = rec.get("hbm_bytes_per_launch")
                    traffic_source = "profiles/pmc_summary.json (static: separate rocprofv3 --pmc passes, round %s)" \
                                     % rec.get("round", "1")
            except Exception:  # noqa: BLE001
                traffic = None
        out = {
            "metric": "Mcells/s through SMRF progressive_filter, %dx%d %s DEM" % (n, n, "fp32" if elem == 4 else "fp64"),
            "value": cells / (ms_median * 1e-3) / 1e6, "unit": "Mcells/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": ms_median, "ms_per_step_mean": ms_mean,
            "step_ms": [round(float(v), 3) for v in step_ms], "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": "%s %dx%d %s, windows 1..%d, cellsize 1, slope_threshold 0.15, synth_dem(seed=20240)"
                                   % ("neilpy_amd.progressive_filter (public drop-in call, CUDA tensor in, bool mask out)"
                                      if world == 1 else "sharded.progressive_filter_sharded (row bands)",
                                      n, n, "fp32" if elem == 4 else "fp64", a.windows),
                       "sharding": ({"bands": world, "rows_per_band": n // world, "backend": dist.get_backend(),
                                     "exchanges_per_step": state.get("exchanges", 0),
                                     "groups": state.get("groups"),
                                     "halo_rows_per_exchange": [sum(2 * r for r in g) for g in state.get("groups", [])],
                                     "message": "RCCL send/recv with rank +- 1, one per group of consecutive windows"}
                                    if world > 1 else "single device"),
                       "object_cells": n_obj, "timing": "value and ms_per_step from the median of the per-step device "
                                                        "times (max over ranks per step); ms_per_step_mean = wall / steps"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": peak, "unit": "GB/s", "frac": achieved / peak,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "frac_traffic": (traffic / avg_launch_s / 1e9 / peak) if traffic else None,
                         "kernel": "smrf::ring_kernel (two passes per window) and the fused small-disk launches (both passes of "
                                   "one or several windows in one launch); %d passes per step, achieved = algorithmic bytes per "
                                   "pass / median device time per pass" % launches,
                         "algorithmic_bytes_per_launch": alg_bytes_launch, "avg_launch_ms": avg_launch_s * 1e3,
                         "device_copy_gbps": copy_gbps, "frac_of_device_copy": achieved / copy_gbps,
                         "classes": classes, "window_ms": window_ms},
        }
        if power is not None:
            out["roofline"]["power"] = power
        if classes and valu:
            # The on-chip bound beside the HBM one (SURVEY 8d "on-chip secondary bound"): the VALU issue port.  Per class:
            # thread-level VALU instructions per cell and window (= wave instructions per 64-cell row) from SQ_INSTS_VALU, the
            # shader clock the counter pass ran at (GRBM_GUI_ACTIVE cycles / kernel time), and valu_frac = instructions x 4.1
            # cycles (what a min / max costs the issue port; these kernels' VALU work is min / max) / (1024 SIMDs x clock x the
            # class's time in the TIMED step): the fraction of the VALU issue bound the class runs at.
            for name, c in classes.items():
                v = valu.get(name)
                if not v:
                    continue
                clock_hz = v["cycles"] / (v["ns"] * 1e-9)
                c["valu_inst_per_cell"] = v["insts"] * 64.0 / (cells * c["windows"])
                c["shader_clock_ghz"] = clock_hz / 1e9
                c["valu_frac"] = v["insts"] * VALU_CYCLES_PER_MINMAX / (SIMDS * clock_hz * c["ms"] * 1e-3)
            tp = classes.get("two_pass", {})
            out["roofline"]["secondary_bound"] = {
                "bound": "valu_issue", "class": "two_pass", "frac": tp.get("valu_frac"), "hbm_frac": tp.get("frac"),
                "note": "the dominant class sits between its two bounds: frac of the VALU issue bound (SQ_INSTS_VALU x 4.1 cycles "
                        "per wave64 min / max over 1024 SIMDs at the measured shader clock) beside its fraction of 8 TB/s"}
        if classes:
            # the bytes the launches really move (a chain of k windows (8 + 2k) / k B per cell and window, a fused opening 10,
            # two ring passes 22 in fp32) over the time of the same timed call: the physical counterpart of `frac`, which
            # credits every window SURVEY 8d's 22 B whatever launch ran it
            moved = sum(c["bytes_per_cell_window"] * c["windows"] for c in classes.values()) * cells
            t_cls = sum(c["ms"] for c in classes.values()) * 1e-3
            out["roofline"]["achieved_moved"] = moved / t_cls / 1e9
            out["roofline"]["frac_moved"] = moved / t_cls / 1e9 / peak
            out["roofline"]["frac_convention"] = ("frac / achieved: SURVEY 8d's 5s + 2 B per cell and window for every window; "
                                                  "frac_moved: the bytes each launch class moves; frac_traffic: FETCH_SIZE + WRITE_SIZE")
        if per_rank is not None:
            out["per_rank"] = per_rank
        if world == 1 and not a.no_secondary and (n, a.windows, a.dtype) == (16384, 50, "f32"):
            try:
                Z = None
                torch.cuda.empty_cache()
                out["secondary"] = secondary(dev)
            except Exception as e:  # noqa: BLE001  (the headline stands on its own)
                out["secondary"] = {"error": repr(e)[:300]}
        if crop is not None:
            out["cpu_baseline"] = cpu_baseline(crop, windows, cellsize, slope)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
