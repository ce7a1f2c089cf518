#!/usr/bin/env python3
"""Reduce tools/pmc_lsqr.sh's two counter passes to bytes per real LSQR iteration and per cell (markdown)."""
import collections
import csv
import glob
import os
import re
import sys

base = sys.argv[1]
info = re.search(r"LSQR n=(\d+) cells=(\d+) istop=(\d+) itn=(\d+) unknowns=(\d+) (?:event_)?ms=([\d.]+)", open(os.path.join(base, "FETCH_SIZE.log")).read())
n, cells, istop, itn, nunk = (int(info.group(i)) for i in range(1, 6))
val = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(base, c, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c:
                agg[re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))].append(float(r["Counter_Value"]) * 1024.0)
    val[c] = agg
cal_r = [v for k, v in val["FETCH_SIZE"].items() if "negate" in k][0][0]
cal_w = [v for k, v in val["WRITE_SIZE"].items() if "negate" in k][0][0]
known = 8.0 * cells
sr, sw = known / cal_r, known / cal_w
print("Spring-inpaint LSQR on %d x %d float64, %d unknowns (%.0f %% of the cells), istop %d after %d iterations." % (n, n, nunk, 100.0 * nunk / cells, istop, itn))
print("Counters: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (KiB); calibration on smrf_negate_f64 over one plane "
      "(8 B per lane, known %d B each way): reads x %.3f, writes x %.3f." % (int(known), sr, sw))
print()
print("| kernel | launches | with traffic | read GB / real launch | written GB | B / cell read | B / cell written |")
print("|---|---|---|---|---|---|---|")
tot = 0.0
for k in sorted(val["FETCH_SIZE"]):
    if not re.search(r"atu_kernel|atuxw_kernel|av2_kernel|setup_kernel|xwav_kernel|av_kernel|reduce_scalar|rhs_kernel|mask_kernel|scatter|w_init", k):
        continue
    fr = val["FETCH_SIZE"][k]
    wr = val["WRITE_SIZE"].get(k, [0.0] * len(fr))
    live = [i for i, v in enumerate(fr) if v * sr > 0.01 * known]        # the launches past the stop return at once
    if not live:
        continue
    r = sum(fr[i] for i in live) * sr / len(live)
    w = sum(wr[i] for i in live if i < len(wr)) * sw / len(live)
    print("| `%s` | %d | %d | %.3f | %.3f | %.1f | %.1f |" % (k[:60], len(fr), len(live), r / 1e9, w / 1e9, r / cells, w / cells))
    if re.search(r"atu_kernel<false>|xwav_kernel", k) and not any("atuxw_kernel" in q for q in val["FETCH_SIZE"]):
        tot += r + w                                        # round 4's iteration
    if re.search(r"atuxw_kernel|av2_kernel", k):
        tot += r + w                                        # round 5's: atuxw (average of the passes with and without the x step) + av2
print()
print("Per iteration (atu + xwav, or since round 5 atuxw + av2): %.2f GB = %.1f B per raster cell (model of DESIGN 4.3: 99 B since round 5, "
      "106 B for round 4's iteration)." % (tot / 1e9, tot / cells))
