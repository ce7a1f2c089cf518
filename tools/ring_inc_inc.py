#!/usr/bin/env python3
"""Per-radius switch of the ring kernels' incremental window widths (morph_ring.h, RingCfg::INC) ->
neilpy_amd/csrc/ring_inc.inc.

Input: logs of ``tools/ring_probe.py --libs <build with "-DSMRF_RING_INC(T,R)=1">`` run with a library built with
``"-DSMRF_RING_INC(T,R)=0"`` (or the tables all zero) as the current one - erosion and ``--flag`` runs per dtype:

    python tools/ring_inc_inc.py --f32 <dir>/inc_all_erode.log <dir>/inc_all_flag.log \\
                                 --f64 <dir>/inc_f64all_erode.log <dir>/inc_f64all_flag.log

A radius is switched on when erosion + dilation/flag together are at least 0.7 % faster with it.
The logs the committed table was written from are kept in profiles/tuning/r02_ring_table_inputs.tar.gz
(inc_all_*.log / inc_f64all_*.log; unpack into <dir>).
"""
import argparse
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(paths):
    cur, var = {}, {}
    for path in paths:
        for line in open(path):
            m = re.match(r"r=\s*(\d+) (\S+)\s+([\d.]+) ms", line)
            if m:
                d = cur if m.group(2) == "cur" else var
                d[int(m.group(1))] = d.get(int(m.group(1)), 0.0) + float(m.group(3))
    return cur, var


def table(paths):
    cur, var = load(paths)
    on = [0] * 65
    for r in range(1, 65):
        if r in cur and r in var and var[r] < cur[r] * 0.993:
            on[r] = 1
    rs = [r for r in cur if r >= 1]
    return on, sum(var[r] if on[r] else cur[r] for r in rs), sum(cur[r] for r in rs)


def fmt(a):
    return ",\n    ".join(", ".join(str(v) for v in a[i:i + 17]) for i in range(0, len(a), 17))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f32", nargs="+", required=True)
    ap.add_argument("--f64", nargs="+", required=True)
    a = ap.parse_args()
    t32, t64 = table(a.f32), table(a.f64)
    out = """// Per-radius switch of the incremental window widths of the ring kernels (morph_ring.h, RingCfg::INC), measured on
// MI355X (tools/ring_probe.py, erosion + dilation/flag, both builds interleaved in one process) and written by
// tools/ring_inc_inc.py.  Index = radius (0 unused).  fp32, all radii summed: %.1f -> %.1f ms; fp64 (8192^2): %.1f -> %.1f ms.
// "-DSMRF_RING_INC(T,R)=0" / "=1" overrides it in tuning builds.
inline constexpr unsigned char kRingIncF32[65] = {
    %s};
inline constexpr unsigned char kRingIncF64[65] = {
    %s};
template <typename T> constexpr bool ring_tuned_inc(int r) {
  return r <= 64 && (sizeof(T) == 4 ? kRingIncF32[r] : kRingIncF64[r]) != 0;
}
""" % (t32[2], t32[1], t64[2], t64[1], fmt(t32[0]), fmt(t64[0]))
    open(os.path.join(ROOT, "neilpy_amd", "csrc", "ring_inc.inc"), "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
