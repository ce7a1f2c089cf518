#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc runs of tools/ring_probe.py (developer tool)."""
import collections
import csv
import glob
import os
import re
import sys

base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
res = collections.defaultdict(dict)
for d in ("pmc1", "pmc2", "pmc3"):
    fs = sorted(glob.glob(os.path.join(base, d, "*", "*_counter_collection.csv")), key=lambda f: -os.path.getmtime(f))
    if not fs:
        continue
    for r in csv.DictReader(open(fs[0])):
        m = re.search(r"ring_kernel<float, (\d+)", r["Kernel_Name"])
        if m:
            res[int(m.group(1))][r["Counter_Name"]] = float(r["Counter_Value"])
for R, c in sorted(res.items()):
    wc = c["SQ_WAVE_CYCLES"]
    cyc = c["GRBM_GUI_ACTIVE"] / 8
    print("R=%d waves %d  kernel %.3g Mcycles  wait_any %.2f  wait_inst %.2f  active %.2f" %
          (R, c["SQ_WAVES"], cyc / 1e6, c["SQ_WAIT_ANY"] / wc, c["SQ_WAIT_INST_ANY"] / wc, c["SQ_ACTIVE_INST_ANY"] / wc))
    print("   VALU/SIMD/cycle %.3f  LDS-pipe busy %.3f  avg waves/SIMD %.2f" %
          (c["SQ_INSTS_VALU"] / 1024 / cyc, c["SQ_LDS_IDX_ACTIVE"] / 256 / cyc, wc * 4 / cyc / 1024))
    print("   insts VALU %.3g LDS %.3g SALU %.3g VMEM rd %.3g wr %.3g  LDS cycles/inst %.2f  bank conflicts %g" %
          (c["SQ_INSTS_VALU"], c["SQ_INSTS_LDS"], c["SQ_INSTS_SALU"], c["SQ_INSTS_VMEM_RD"], c["SQ_INSTS_VMEM_WR"],
           c["SQ_LDS_IDX_ACTIVE"] / c["SQ_INSTS_LDS"], c["SQ_LDS_BANK_CONFLICT"]))
