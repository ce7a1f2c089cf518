#!/usr/bin/env python3
"""Turn the tables measured by tools/ring_tune.py into neilpy_amd/csrc/ring_tune.inc.

    python tools/ring_tune_inc.py <dir>/ring_tune_f32.json <dir>/ring_tune_f64.json

Per radius the simplest build within 1% of the fastest wins: the default (n2d0) first, then the
builds at the estimated occupancy, then the rest.
The logs the committed table was written from are kept in profiles/tuning/r02_ring_table_inputs.tar.gz
(ring_tune*_f32.json / ring_tune*_f64.json; unpack into <dir>).
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pick(ms):
    best = min(ms.values())
    order = sorted(ms, key=lambda v: (v != "n2d0", "t" in v, v[3] != "0", ms[v]))
    for v in order:
        if ms[v] <= best * 1.01:
            return v
    raise AssertionError


def rows(path):
    t = json.load(open(path))
    npm, drop, top3 = [0], [0], [0]
    for r in range(1, 65):
        if str(r) not in t:                  # a table measured for radii 1..50 only: the defaults above it
            npm.append(2); drop.append(0); top3.append(0)
            continue
        v = pick(t[str(r)]["ms"])
        m = re.match(r"n(\d)d(\d)(?:t(\d+))?", v)
        npm.append(int(m.group(1)))
        drop.append(int(m.group(2)))
        # radii the fused opening covers keep the full table in both kernels (the fused kernel shares RingCfg and was
        # tuned on its own)
        top3.append(int(m.group(3) or 0) if r > 8 else 0)
    return npm, drop, top3


def fmt(a):
    return ",\n    ".join(", ".join(str(v) for v in a[i:i + 17]) for i in range(0, len(a), 17))


def main():
    f32, f64 = rows(sys.argv[1]), rows(sys.argv[2])
    out = """// Per-radius tuning of the ring kernels, measured on MI355X by tools/ring_tune.py (one
// progressive-filter window per radius on the 16384^2 benchmark DEM, all builds interleaved in
// one process) and written by tools/ring_tune_inc.py.  Index = radius (0 unused).
//   np_max:   most row pairs per batch the chooser ring_np() may take (fp64 takes half of it)
//   occ_drop: steps below the estimated waves/SIMD the kernel is built for (more registers,
//             fewer resident workgroups)
//   top3:     how many of the disk's widest widths may be looked up with three reads of the level below
//             instead of building the top table level for them (RingCfg::DROP_TOP); 0 = never
// -DSMRF_RING_NO_TUNE ignores the tables (tuning builds).  (Included inside namespace smrf.)
#ifndef SMRF_RING_TOP3_ALL
#define SMRF_RING_TOP3_ALL 0
#endif
#ifdef SMRF_RING_NO_TUNE
template <typename T> constexpr int ring_tuned_np_max(int) { return SMRF_RING_NP_MAX; }
template <typename T> constexpr int ring_tuned_occ_drop(int) { return SMRF_RING_OCC_DROP; }
template <typename T> constexpr int ring_tuned_top3(int) { return SMRF_RING_TOP3_ALL; }
#else
inline constexpr unsigned char kRingNpMaxF32[65] = {
    %s};
inline constexpr unsigned char kRingOccDropF32[65] = {
    %s};
inline constexpr unsigned char kRingTop3F32[65] = {
    %s};
inline constexpr unsigned char kRingNpMaxF64[65] = {
    %s};
inline constexpr unsigned char kRingOccDropF64[65] = {
    %s};
inline constexpr unsigned char kRingTop3F64[65] = {
    %s};
template <typename T> constexpr int ring_tuned_np_max(int r) {
  return r > 64 ? 2 : sizeof(T) == 4 ? kRingNpMaxF32[r] : kRingNpMaxF64[r];
}
template <typename T> constexpr int ring_tuned_occ_drop(int r) {
  return r > 64 ? 0 : sizeof(T) == 4 ? kRingOccDropF32[r] : kRingOccDropF64[r];
}
template <typename T> constexpr int ring_tuned_top3(int r) {
  return r > 64 ? 0 : sizeof(T) == 4 ? kRingTop3F32[r] : kRingTop3F64[r];
}
#endif
""" % (fmt(f32[0]), fmt(f32[1]), fmt(f32[2]), fmt(f64[0]), fmt(f64[1]), fmt(f64[2]))
    open(os.path.join(ROOT, "neilpy_amd", "csrc", "ring_tune.inc"), "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
