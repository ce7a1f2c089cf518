// Instantiates the register-ring kernels for the radii r with r % SMRF_RING_PARTS == PART
// (compiled once per PART and per dtype so the 64 x 2 x 2 kernels build in parallel).
//   hipcc ... -DPART=k -DSMRF_F64=0|1 -c ring_part.hip
#if defined(SMRF_RING_TW) && SMRF_F64 && SMRF_RING_TW > 256
#undef SMRF_RING_TW            // tuning builds with 512-column strips: fp32 only (LDS offsets of the fp64 table exceed 16 bits)
#define SMRF_RING_TW 256
#endif
#include "morph_fused.h"

#ifndef PART
#error "compile with -DPART=0..7"
#endif
#if SMRF_F64
using elem_t = double;
#define SMRF_RING_FN2(P) smrf_ring_f64_p##P
#else
using elem_t = float;
#define SMRF_RING_FN2(P) smrf_ring_f32_p##P
#endif
#define SMRF_RING_FN1(P) SMRF_RING_FN2(P)
#define SMRF_RING_FN SMRF_RING_FN1(PART)

namespace {
template <int R>
int launch_r(const DiskArgs<elem_t>& a, int mode, hipStream_t s) {
  if (mode == SMRF_RING_FUSED_OPEN) {
    if constexpr (smrf_fused_radius((int)sizeof(elem_t), R)) return smrf::fused_launch<elem_t, R>(a, s);
    else return smrf_fail(SMRF_E_UNSUPPORTED, "no fused opening kernel for radius %d at this dtype", R);
  }
  return mode == SMRF_RING_DILATE ? smrf::ring_launch<elem_t, R, true>(a, s) : smrf::ring_launch<elem_t, R, false>(a, s);
}
}  // namespace

int SMRF_RING_FN(const DiskArgs<elem_t>& a, int mode, hipStream_t s) {
  constexpr int P = (PART == 0) ? SMRF_RING_PARTS : PART;   // radii P, P+8, ..., P+56
  switch (a.radius) {
    case P: return launch_r<P>(a, mode, s);
    case P + 8: return launch_r<P + 8>(a, mode, s);
    case P + 16: return launch_r<P + 16>(a, mode, s);
    case P + 24: return launch_r<P + 24>(a, mode, s);
    case P + 32: return launch_r<P + 32>(a, mode, s);
    case P + 40: return launch_r<P + 40>(a, mode, s);
    case P + 48: return launch_r<P + 48>(a, mode, s);
    case P + 56: return launch_r<P + 56>(a, mode, s);
    default: return smrf_fail(SMRF_E_ARG, "ring dispatcher %d got radius %d", PART, a.radius);
  }
}
