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


def stats3(path, n, out, elem=4):
    """Round 3 form: chain / fused / ring launches, every launch priced at the bytes it really moves per cell (a chain of
    k windows 2s + 2k, a fused window 2s + 2, a ring erosion 2s, a ring dilation + flag 3s + 2: no GB/s above the peak),
    beside SURVEY 8d's convention (5s + 2 per window, half of it per pass)."""
    rows = list(csv.DictReader(open(path)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    cells = n * n
    hot, other = [], []
    for r in rows:
        k = r["Name"]
        m = (re.search(r"(ring_kernel)<(\w+), (\d+), (true|false)", k) or re.search(r"(fused_open_kernel)<(\w+), (\d+)", k) or
             re.search(r"(chain_kernel)<(\w+), \d+, \d+, (\d+), (\d+), (\d+), (\d+)", k))
        if not m:
            other.append(r)
            continue
        g = m.groups()
        avg, calls = float(r["AverageNs"]), int(r["Calls"])
        if g[0] == "ring_kernel":
            dil = g[3] == "true"
            hot.append((int(g[2]), 1 if dil else 0, "ring dilation + flag" if dil else "ring erosion", 1, (3 * elem + 2) if dil else 2 * elem,
                        avg, calls))
        elif g[0] == "fused_open_kernel":
            hot.append((int(g[2]), 0, "fused opening + flag", 2, 2 * elem + 2, avg, calls))
        else:
            radii = [int(v) for v in g[2:] if int(v)]
            name = "chain %s" % ", ".join(str(v) for v in radii) if len(radii) > 1 else "table-free opening + flag"
            hot.append((radii[0], 0, name, 2 * len(radii), 2 * elem + 2 * len(radii), avg, calls))
    hot.sort()
    ht = sum(h[5] * h[6] for h in hot)
    passes = sum(h[3] * h[6] for h in hot)
    moved = sum(h[4] * cells * h[6] for h in hot)
    lines = ["# rocprofv3 --kernel-trace --stats summary (bench.py, %dx%d %s)" % (n, n, "fp32" if elem == 4 else "fp64"), "",
             "total kernel time %.3f ms; progressive_filter kernels (chain / fused / ring) %.3f ms (%.1f %%), %d launches = %d passes "
             "(a pass = half a window; a fused launch is 2, a chain of k windows 2k), average per pass %.4f ms"
             % (tot / 1e6, ht / 1e6, 100 * ht / tot, sum(h[6] for h in hot), passes, ht / passes / 1e6), "",
             "SURVEY 8d convention: %d B/cell per pass = %.3f GB -> %.0f GB/s over all passes (frac %.3f of 8 TB/s)"
             % ((5 * elem + 2) // 2, cells * (5 * elem + 2) / 2 / 1e9, cells * (5 * elem + 2) / 2 / (ht / passes), cells * (5 * elem + 2) / 2 / (ht / passes) / 8000),
             "bytes the launches really move (model below): %.1f GB per step -> %.0f GB/s (frac %.3f)"
             % (moved / max(1, min(h[6] for h in hot)) / 1e9, moved / ht, moved / ht / 8000), "",
             "| first radius | launch | windows' passes | B/cell moved | avg ms | GB/s moved | frac of 8 TB/s | calls |",
             "|---|---|---|---|---|---|---|---|"]
    for h in hot:
        lines.append("| %d | %s | %d | %d | %.3f | %.0f | %.2f | %d |" % (h[0], h[2], h[3], h[4], h[5] / 1e6, cells * h[4] / h[5],
                                                                          cells * h[4] / h[5] / 8000, h[6]))
    # per window (sum of a window's launches; chains as a whole)
    lines += ["", "per window or chain (ms): " + ", ".join(
        "%s %.3f" % (k, v / 1e6) for k, v in _windows(hot)), "", "other kernels:", ""]
    for r in other:
        lines.append("- %s: %s calls, %.3f ms" % (r["Name"][:90], r["Calls"], float(r["TotalDurationNs"]) / 1e6))
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:8]))


def _windows(hot):
    agg = collections.OrderedDict()
    for h in hot:
        key = ("R=%d" % h[0]) if h[2].startswith(("ring", "fused", "table")) else h[2]
        agg[key] = agg.get(key, 0.0) + h[5]
    return agg.items()


ROUND = 5


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
    hot = lambda k: "ring_kernel" in k or "fused_open_kernel" in k or "chain_kernel" in k           # noqa: E731
    ring_f = sum(sum(v) for k, v in f.items() if hot(k)) * 1024.0
    ring_w = sum(sum(v) for k, v in w.items() if hot(k)) * 1024.0
    # per PASS (half a window, 11 B/cell algorithmic): a fused launch counts as two, as in bench.py's roofline
    def passes_of(k):                                       # a fused launch is two passes, a chain of k windows 2k
        if "fused_open_kernel" in k:
            return 2
        m = re.search(r"chain_kernel<\w+, \d+, \d+, (\d+), (\d+), (\d+), (\d+)", k)
        return 2 * sum(1 for v in m.groups() if int(v)) if m else 1
    launches = sum(len(v) * passes_of(k) for k, v in f.items() if hot(k))
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
    elif sys.argv[1] == "stats3":
        stats3(sys.argv[2], int(sys.argv[3]), sys.argv[4], int(sys.argv[5]) if len(sys.argv) > 5 else 4)
    else:
        pmc(sys.argv[2], sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), sys.argv[6])
