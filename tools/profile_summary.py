#!/usr/bin/env python3
"""Turn rocprofv3 output of bench.py / pmc_traffic.py into the summaries kept under profiles/.

    python tools/profile_summary.py stats  <kernel_stats.csv> <size> <out.md>
    python tools/profile_summary.py pmc    <fetch_counter.csv> <write_counter.csv> <size> <windows> <out.json>
"""
import collections
import csv
import json
import re
import sys


def stats(path, n, out):
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    ring = [r for r in rows if "ring_kernel" in r["Name"]]
    fused = [r for r in rows if "fused_open_kernel" in r["Name"]]
    rt = sum(float(r["TotalDurationNs"]) for r in ring + fused)
    # a "pass" = half a window (erosion, or dilation + flag: 11 B/cell on average); a fused launch is two passes
    passes = sum(int(r["Calls"]) for r in ring) + 2 * sum(int(r["Calls"]) for r in fused)
    launches = sum(int(r["Calls"]) for r in ring + fused)
    cells = n * n
    lines = ["# rocprofv3 --kernel-trace --stats summary (bench.py, %dx%d fp32)" % (n, n), "",
             "total kernel time %.3f ms; smrf::ring_kernel + smrf::fused_open_kernel instances %.3f ms (%.1f %%), %d launches = "
             "%d passes (a fused launch is an erosion and a dilation + flag pass), average per pass %.4f ms"
             % (tot / 1e6, rt / 1e6, 100 * rt / tot, launches, passes, rt / passes / 1e6), "",
             "algorithmic bytes per pass: erosion 2*4 B/cell, dilation+flag 3*4+2 B/cell (mean 11 B/cell = %.3f GB), SURVEY 8d"
             % (cells * 11 / 1e9),
             "achieved over all passes: %.0f GB/s" % (cells * 11 / (rt / passes)), "",
             "| radius | kernel(s) | erosion avg ms | GB/s (8 B/cell) | dilation+flag avg ms | GB/s (14 B/cell) | window ms | GB/s (22 B/cell) | calls |",
             "|---|---|---|---|---|---|---|---|---|"]
    tab, ftab = {}, {}
    for r in ring:
        m = re.search(r"ring_kernel<\w+, (\d+), (true|false)", r["Name"])
        tab[(int(m.group(1)), m.group(2) == "true")] = (float(r["AverageNs"]), int(r["Calls"]))
    for r in fused:
        m = re.search(r"fused_open_kernel<\w+, (\d+)", r["Name"])
        ftab[int(m.group(1))] = (float(r["AverageNs"]), int(r["Calls"]))
    for R in sorted({k[0] for k in tab} | set(ftab)):
        if R in ftab:
            f = ftab[R]
            lines.append("| %d | fused | - | - | - | - | %.3f | %.0f | %d |" % (R, f[0] / 1e6, cells * 22 / f[0], f[1]))
            continue
        e, d = tab.get((R, False)), tab.get((R, True))
        lines.append("| %d | ring x2 | %.3f | %.0f | %.3f | %.0f | %.3f | %.0f | %d |" % (
            R, e[0] / 1e6, cells * 8 / e[0], d[0] / 1e6, cells * 14 / d[0], (e[0] + d[0]) / 1e6, cells * 22 / (e[0] + d[0]),
            e[1] + d[1]))
    lines += ["", "other kernels:", ""]
    for r in rows:
        if "ring_kernel" not in r["Name"] and "fused_open_kernel" not in r["Name"]:
            lines.append("- %s: %s calls, %.3f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))


ROUND = 2


def pmc(fetch_csv, write_csv, n, windows, out):
    def load(path, name):
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == name:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
        return agg
    f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
    cal = [v for k, v in f.items() if "count_nan" in k]
    known = 4.0 * n * n
    scale = known / (cal[0][0] * 1024.0) if cal else None   # FETCH_SIZE is in KiB; gfx950 under-reports reads
    hot = lambda k: "ring_kernel" in k or "fused_open_kernel" in k           # noqa: E731
    ring_f = sum(sum(v) for k, v in f.items() if hot(k)) * 1024.0
    ring_w = sum(sum(v) for k, v in w.items() if hot(k)) * 1024.0
    # per PASS (half a window, 11 B/cell algorithmic): a fused launch counts as two, as in bench.py's roofline
    launches = sum(len(v) * (2 if "fused_open_kernel" in k else 1) for k, v in f.items() if hot(k))
    rec = dict(n=n, windows=windows, dtype="f32", launches=launches, round=ROUND,
               fetch_bytes_raw=ring_f, write_bytes=ring_w, fetch_calibration=scale,
               calibration_note="count_nan reads 4*n*n bytes with one dword per lane; scale = known / FETCH_SIZE",
               fetch_bytes_corrected=ring_f * scale if scale else None,
               hbm_bytes_per_launch=((ring_f * scale if scale else ring_f) + ring_w) / max(launches, 1),
               algorithmic_bytes_per_launch=n * n * 11.0)
    if out:
        json.dump(rec, open(out, "w"), indent=1)
        print(json.dumps(rec, indent=1))
    return rec


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], int(sys.argv[3]), sys.argv[4])
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
