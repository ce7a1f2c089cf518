#!/usr/bin/env python3
"""One line per kernel instance from the three rocprofv3 --pmc passes of tools/pmc_kernels.sh (markdown table).

VALU busy = SQ_INSTS_VALU x 4.14 cycles (the min / max issue rate, profiles/r02_issue_rate_ubench.md) / 1024 SIMDs / kernel
cycles; LDS busy = SQ_LDS_IDX_ACTIVE / 256 CUs / kernel cycles; waves per SIMD = SQ_WAVE_CYCLES x 4 / 1024 / kernel cycles;
kernel cycles = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs).
"""
import collections
import csv
import glob
import os
import re
import sys

base = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for d in ("pmc1", "pmc2", "pmc3"):
    for f in glob.glob(os.path.join(base, d, "**", "*_counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            m = (re.search(r"(ring_kernel)<(float|double), (\d+), (true|false), \d+, (\d+)", k) or
                 re.search(r"(fused_open_kernel)<(float|double), (\d+), \d+, (\d+)", k) or
                 re.search(r"(chain_kernel)<(float|double), (\d+), \d+, (\d+), (\d+), (\d+), (\d+)", k))
            if not m:
                continue
            g = m.groups()
            if g[0] == "ring_kernel":
                key = (int(g[2]), "ring %s" % ("dilate+flag" if g[3] == "true" else "erode"), "NP=%s" % g[4])
            elif g[0] == "fused_open_kernel":
                key = (int(g[2]), "fused open+flag", "NP=%s" % g[3])
            else:
                radii = [int(v) for v in g[3:] if int(v)]
                key = (radii[0], "chain %s" % ",".join(str(v) for v in radii), "NP=%s" % g[2])
            res[key][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] in ("SQ_WAVES", "SQ_INSTS_SALU", "SQ_WAIT_ANY"):
                calls[(key, r["Counter_Name"])] += 1
print("| R | kernel | | ms | waves/SIMD | VALU busy | LDS busy | waves waiting | VALU inst/row | LDS inst/row | SALU inst/row |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
n_cells = None
for key, c in sorted(res.items()):
    if "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_SALU" not in c or "SQ_WAIT_ANY" not in c:
        continue
    nc = max(1, calls[(key, "SQ_WAVES")])
    cyc = c["GRBM_GUI_ACTIVE"] / 8 / nc
    wc = c["SQ_WAVE_CYCLES"] / nc
    rows64 = float(os.environ.get("PMC_CELLS", 16384 * 16384)) / 64.0      # 64-cell row pieces of one launch
    print("| %d | %s | %s | %.3f | %.2f | %.2f | %.2f | %.2f | %.1f | %.1f | %.1f |" % (
        key[0], key[1], key[2], cyc / 2.1e6, wc * 4 / 1024 / cyc, c["SQ_INSTS_VALU"] / nc * 4.14 / 1024 / cyc,
        c["SQ_LDS_IDX_ACTIVE"] / max(1, calls[(key, "SQ_INSTS_SALU")]) / 256 / cyc,
        c["SQ_WAIT_ANY"] / max(1, calls[(key, "SQ_WAIT_ANY")]) / wc,
        c["SQ_INSTS_VALU"] / nc / rows64, c["SQ_INSTS_LDS"] / nc / rows64,
        c["SQ_INSTS_SALU"] / max(1, calls[(key, "SQ_INSTS_SALU")]) / rows64))
