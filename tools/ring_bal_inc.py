#!/usr/bin/env python3
"""Per-radius switch of the ring kernels' balanced halo build (morph_ring.h, HaloCfg) -> neilpy_amd/csrc/ring_bal.inc.

Input: logs of ``tools/ring_probe.py --libs <a build with -DSMRF_RING_BAL=0>`` run with the library built with
``-DSMRF_RING_BAL_ALL=1`` as the current one - erosion and ``--flag`` runs, fp32 and fp64, any split of the radii:

    python tools/ring_bal_inc.py --f32 <dir>/bal_all_erode.log <dir>/bal_all_flag.log ... \\
                                 --f64 <dir>/bal_f64all_erode.log <dir>/bal_f64all_flag.log

A radius is switched on when erosion + dilation/flag together are at least 0.7 % faster with the balanced build
(run-to-run noise of the interleaved medians is about 1 %); radii from 59 up are left off (no consistent sign).
The logs the committed table was written from are kept in profiles/tuning/r02_ring_table_inputs.tar.gz
(bal_all_*.log / bal_more_*.log / bal_f64all_*.log; unpack into <dir>).
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
    for r in range(1, 59):
        if r in cur and r in var and cur[r] < var[r] * 0.993:
            on[r] = 1
    return on, sum(cur[r] if on[r] else var[r] for r in var), sum(var.values())


def fmt(a):
    return ",\n    ".join(", ".join(str(v) for v in a[i:i + 17]) for i in range(0, len(a), 17))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--f32", nargs="+", required=True)
    ap.add_argument("--f64", nargs="+", required=True)
    a = ap.parse_args()
    t32, t64 = table(a.f32), table(a.f64)
    out = """// Per-radius switch of the balanced halo build of the ring kernels (morph_ring.h, HaloCfg), measured on MI355X
// (tools/ring_probe.py, erosion + dilation/flag, balanced against plain build interleaved in one process) and written
// by tools/ring_bal_inc.py.  Index = radius (0 unused).  fp32: all radii summed %.1f -> %.1f ms; fp64 (8192^2): %.1f -> %.1f ms.
// -DSMRF_RING_BAL=0 switches it off everywhere, -DSMRF_RING_BAL_ALL=1 on everywhere (tuning builds).
inline constexpr unsigned char kRingBalF32[65] = {
    %s};
inline constexpr unsigned char kRingBalF64[65] = {
    %s};
template <typename T> constexpr bool ring_tuned_bal(int r) {
#if defined(SMRF_RING_BAL_ALL) && SMRF_RING_BAL_ALL
  return true;
#else
  return r <= 64 && (sizeof(T) == 4 ? kRingBalF32[r] : kRingBalF64[r]) != 0;
#endif
}
""" % (t32[2], t32[1], t64[2], t64[1], fmt(t32[0]), fmt(t64[0]))
    open(os.path.join(ROOT, "neilpy_amd", "csrc", "ring_bal.inc"), "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
