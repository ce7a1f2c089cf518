// Shared host/device helpers of libsmrf_hip (gfx950 only; no other backend is supported).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "smrf_hip.h"

#define SMRF_HIDDEN __attribute__((visibility("hidden")))

SMRF_HIDDEN int smrf_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define SMRF_HIP_CHECK(expr)                                                                  \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return smrf_fail(SMRF_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                       __FILE__, __LINE__);                                                   \
  } while (0)

#define SMRF_LAUNCH_CHECK() SMRF_HIP_CHECK(hipGetLastError())

// scipy.ndimage mode='reflect' (d c b a | a b c d | d c b a): period 2n fold, any i, n >= 1
__host__ __device__ inline int smrf_fold(int i, int n) {
  const int p2 = 2 * n;
  int p = i % p2;
  if (p < 0) p += p2;
  return p < n ? p : p2 - 1 - p;
}

// The SMRF_* environment switches (developer A/B runs; the parity tests force every launch variant through them), read
// once by core.hip - smrf_switches_reload() of the C ABI reads them again.  -1 / 0 = "by the library's own rule".
struct SmrfSwitches {
  int fused;          // SMRF_FUSED: 0 never, 1 by rule, 2 every radius that has a fused / chained kernel whatever the raster size
  int chain;          // SMRF_CHAIN: 0 = no chained / table-free launches
  int nan_ride;       // SMRF_NAN_RIDE: 0 = always a separate NaN count pass
  int nt;             // SMRF_NT: -1 by plane size, 0 / 1 forced
  int ring_seg;       // SMRF_RING_SEG: output rows per workgroup (0 = from the occupancy)
  int ring_dual;      // SMRF_RING_DUAL: -1 by segment length, 0 shifting ring, 1 in-place ring
  int ring_rounds, fused_rounds, chain_rounds;   // workgroups per resident slot
  int seg_rule;       // SMRF_SEG_RULE: how a launch is cut into row segments (smrf_pick_nseg): 0 = by the cost model (default);
                      // 2 = one full round, rounds x resident x 256 / strips rounded down, segments >= 4R; 1 = rounded to nearest
                      // (rounds 1-4's rule)
  int xcd_remap;      // SMRF_XCD_REMAP=0: no XCD-aware tile placement in the ring kernels (A/B runs)
  int ring_slope;     // SMRF_RING_SLOPE: permille of segment length per residency class (-1 = the library's rule, 0 = equal segments)
  int ring_debug;     // SMRF_RING_DEBUG: print each instance's geometry once
};
SMRF_HIDDEN const SmrfSwitches& smrf_sw();

// integer floor(sqrt(v)), v >= 0
__host__ __device__ constexpr int smrf_isqrt(int v) {
  int r = 0;
  while ((long long)(r + 1) * (r + 1) <= v) ++r;
  return r;
}

// largest float <= t.  The flag step compares the raster dtype's difference with the float64 threshold in float64
// (neilpy.py:1671 under NumPy 2).  For a float `diff`: diff > t implies diff > thr_lo (thr_lo <= t); and diff > thr_lo
// implies diff >= the next float above thr_lo, which lies above t by the choice of thr_lo.  So `diff > thr_lo` is the same
// predicate, in one fp32 instruction instead of a conversion and a float64 compare (NaN: false either way).
inline float smrf_float_below(double t) {
  float f = (float)t;
  if ((double)f > t) f = __builtin_nextafterf(f, -__builtin_inff());
  return f;
}

#include "seg_rule.h"   // smrf_pick_nseg: how a launch is cut into row segments (plain C++: tests/test_host_logic.py compiles it)

// arguments of one disk erosion / dilation pass over a row band (see smrf_hip.h)
template <typename T>
struct DiskArgs {
  const T* in;        // first row held = global row in_row0
  T* out;             // first row = global row out_row0
  const T* last;      // flag step only (NULL otherwise), first row = out_row0
  uint8_t* mask;      // flag step only
  uint8_t* when;      // flag step only, may be NULL
  double thr;
  float thr_lo;       // largest float <= thr: for fp32 rasters `diff > thr_lo` decides exactly as `(double)diff > thr`
  int widx;
  int img_rows, cols;
  long long ld;
  int in_row0, in_rows, out_row0, out_rows;
  int radius;
  int nan_aware;
  int seg;            // output rows per workgroup (ring kernels)
  int nt;             // output cells as non-temporal (streaming) stores: planes far larger than the caches
  int dense;          // flag step writes EVERY mask / when byte (0 included): the planes need no clearing first
  // segments of unequal length (ring kernels, round 5): seg_cls = number of classes (0: every segment is `seg` rows); class c
  // holds the segments seg_first[c] .. seg_first[c + 1] - 1, each seg_len[c] rows, the first at out_row0 + seg_row0[c].
  // See ring_launch_np.
  int seg_cls;
  int seg_first[8], seg_row0[8], seg_len[8];
  int plain_tiles;    // 1: workgroup (x, y) takes tile (x, y) (SMRF_XCD_REMAP=0, A/B runs); 0: the XCD-aware placement of the kernel
};

// ring-kernel dispatchers, one per (dtype, radius % SMRF_RING_PARTS); defined in ring_part.hip.
// mode: erosion, dilation (+ flag step when mask != NULL), or the fused opening + flag of morph_fused.h (in = last)
enum { SMRF_RING_ERODE = 0, SMRF_RING_DILATE = 1, SMRF_RING_FUSED_OPEN = 2 };
// radii whose progressive_filter window runs as ONE fused opening + flag launch (morph_fused.h), per dtype: measured
// against the two ring passes per radius on the 16384^2 benchmark DEM (gpurun_out/r02/fused3_per_radius_f32.log,
// fused2_per_radius_f64.log, fused_hi_f32.log).  fp32: 1..8 and 10..14 (9 loses by 3 %, 15 and up by 20 % and more);
// fp64, whose tables are twice as large: 1..6.
#ifndef SMRF_FUSED_MAX_RADIUS
#define SMRF_FUSED_MAX_RADIUS 14
#endif
constexpr bool smrf_fused_radius(int elem_size, int r) {
  if (r < 1 || r > SMRF_FUSED_MAX_RADIUS) return false;
  return elem_size == 4 ? (r != 9) : (r <= 6);
}
#define SMRF_RING_PARTS 8
#define SMRF_RING_DECL(P)                                                                     \
  SMRF_HIDDEN int smrf_ring_f32_p##P(const DiskArgs<float>&, int mode, hipStream_t);          \
  SMRF_HIDDEN int smrf_ring_f64_p##P(const DiskArgs<double>&, int mode, hipStream_t);
SMRF_RING_DECL(0) SMRF_RING_DECL(1) SMRF_RING_DECL(2) SMRF_RING_DECL(3)
SMRF_RING_DECL(4) SMRF_RING_DECL(5) SMRF_RING_DECL(6) SMRF_RING_DECL(7)
#undef SMRF_RING_DECL
