#!/usr/bin/env python3
"""Reduce tools/pmc_f64.sh's two counter passes: HBM bytes per window of the fp64 progressive_filter, per launch kind (markdown)."""
import argparse
import collections
import csv
import glob
import os
import re
import sys

ap = argparse.ArgumentParser()
ap.add_argument("base")
ap.add_argument("--size", type=int, default=8192)
ap.add_argument("--windows", type=int, default=18)
a = ap.parse_args()
cells = a.size * a.size
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(a.base, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[r["Kernel_Name"]].append(float(r["Counter_Value"]) * 1024.0)
    val[c] = agg
cal = [v for k, v in val["FETCH_SIZE"].items() if "count_nan" in k][0][0]
scale = 8.0 * cells / cal
print("fp64 progressive_filter, %d x %d, windows 1..%d, one call.  FETCH_SIZE / WRITE_SIZE in separate passes; reads x %.3f by the "
      "calibration kernel (count_nan reads 8 n^2 bytes, one float64 per lane)." % (a.size, a.size, a.windows, scale))
print()
print("| launch | radius | read B / cell | written B / cell | sum | the planes' own (two passes 42, fused / single 18, chain 1, 2 22 for both) |")
print("|---|---|---|---|---|---|")
tot = 0.0
rows = []
for k in val["FETCH_SIZE"]:
    m = (re.search(r"(ring_kernel)<double, (\d+), (true|false)", k) or re.search(r"(fused_open_kernel)<double, (\d+)", k) or
         re.search(r"(chain_kernel)<double, \d+, \d+, (\d+), (\d+)", k))
    if not m:
        continue
    fr = sum(val["FETCH_SIZE"][k]) * scale / cells
    wr = sum(val["WRITE_SIZE"].get(k, [0.0])) / cells
    g = m.groups()
    if g[0] == "ring_kernel":
        name, own = "ring " + ("dilation + flag" if g[2] == "true" else "erosion"), (26 if g[2] == "true" else 16)
    elif g[0] == "fused_open_kernel":
        name, own = "fused opening + flag", 18
    else:
        name, own = ("chain %s, %s" % (g[1], g[2]) if int(g[2]) else "table-free single"), (22 if int(g[2]) else 18)
    rows.append((int(g[1]), name, fr, wr, own))
    tot += fr + wr
for r, name, fr, wr, own in sorted(rows):
    print("| %s | %d | %.1f | %.1f | %.1f | %d |" % (name, r, fr, wr, fr + wr, own))
print()
print("Whole call: %.2f GB = %.1f B per cell and window on average (SURVEY 8d's convention: 42)." % (tot * cells / 1e9, tot / a.windows))
