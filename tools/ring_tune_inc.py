#!/usr/bin/env python3
"""Turn the tables measured by tools/ring_tune.py into neilpy_amd/csrc/ring_tune.inc.

    python tools/ring_tune_inc.py gpurun_out/ring_tune_f32.json gpurun_out/ring_tune_f64.json

Per radius the simplest build within 1% of the fastest wins: the default (n2d0) first, then the
builds at the estimated occupancy, then the rest.
"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def pick(ms):
    best = min(ms.values())
    order = sorted(ms, key=lambda v: (v != "n2d0", v[-1] != "0", ms[v]))
    for v in order:
        if ms[v] <= best * 1.01:
            return v
    raise AssertionError


def rows(path):
    t = json.load(open(path))
    npm, drop = [0], [0]
    for r in range(1, 65):
        v = pick(t[str(r)]["ms"])
        m = re.match(r"n(\d)d(\d)", v)
        npm.append(int(m.group(1)))
        drop.append(int(m.group(2)))
    return npm, drop


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
// -DSMRF_RING_NO_TUNE ignores the tables (tuning builds).  (Included inside namespace smrf.)
#ifdef SMRF_RING_NO_TUNE
template <typename T> constexpr int ring_tuned_np_max(int) { return SMRF_RING_NP_MAX; }
template <typename T> constexpr int ring_tuned_occ_drop(int) { return SMRF_RING_OCC_DROP; }
#else
inline constexpr unsigned char kRingNpMaxF32[65] = {
    %s};
inline constexpr unsigned char kRingOccDropF32[65] = {
    %s};
inline constexpr unsigned char kRingNpMaxF64[65] = {
    %s};
inline constexpr unsigned char kRingOccDropF64[65] = {
    %s};
template <typename T> constexpr int ring_tuned_np_max(int r) {
  return r > 64 ? 2 : sizeof(T) == 4 ? kRingNpMaxF32[r] : kRingNpMaxF64[r];
}
template <typename T> constexpr int ring_tuned_occ_drop(int r) {
  return r > 64 ? 0 : sizeof(T) == 4 ? kRingOccDropF32[r] : kRingOccDropF64[r];
}
#endif
""" % (fmt(f32[0]), fmt(f32[1]), fmt(f64[0]), fmt(f64[1]))
    open(os.path.join(ROOT, "neilpy_amd", "csrc", "ring_tune.inc"), "w").write(out)
    print(out)


if __name__ == "__main__":
    main()
