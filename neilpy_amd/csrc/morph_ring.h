// Register-ring disk erosion / dilation for gfx950 (MI355X), one kernel per radius.
//
// What it computes (scipy.ndimage.grey_erosion/grey_dilation with skimage's disk(R) and
// mode='reflect', as reached from neilpy.py:1670):
//     E[y, x] = min over dy in [-R, R] of  min over |dx| <= w(dy) of  Z[fold(y+dy), fold(x+dx)],
//     w(dy) = isqrt(R*R - dy*dy)
//
// How (details and measurements: DESIGN.md 4.1): a workgroup owns a strip of TW = 256 columns (one
// column per lane) and marches down the rows of its segment, NP row PAIRS per batch.
//   - The two rows of a pair are interleaved per cell in LDS ({rowA[x], rowB[x]}), so one
//     ds_read_b64 / ds_write_b64 serves both rows.  A batch is prefetched into registers one batch
//     ahead with coalesced global loads; level 0 of the table is double buffered.
//   - Per row a table of power-of-two window minima (level j = min over 2^j cells from the cell on).
//     Where ring_inc.inc switches RingCfg::INC on (fp32: R = 8 and up) only levels 0..3 exist, built
//     in ONE phase of 7 independent reads per cell, and the K distinct window minima of disk(R)
//     (K ~ 0.59 R + 1) are taken in ascending order, each GROWN from the one before it with one read
//     per side of the smallest level that spans the step: 2 barriers per batch.  Elsewhere the full
//     sparse table (levels up to log2(2R+1), base level + higher levels in two build phases, 3
//     barriers) and two reads of one level per width.  Either way two ds_read_b64 and one v_min3 /
//     v_min per width and row, in software-pipelined groups of G widths; all table reads are inline
//     asm ds_read_b64 with counted s_waitcnt: hipcc would fuse them into half-rate ds_read2_b64.
//   - The 2R halo cells of a staged row are shared out over the four waves row pair by row pair
//     (HaloCfg, ring_bal.inc) so that the waves reach the phase barriers together.
//   - The 2R output rows a pair contributes to are 2R accumulators in REGISTERS: slot s belongs to
//     output row y_in - R + s.  After a pair every slot moves down by two for free, because
//         acc[s] = min3(acc[s+2], RA[k(R-s-2)], RB[k(R-s-1)])
//     writes a different register than it reads; with R a template parameter every slot index and
//     window width is a compile-time constant, so the update is straight-line v_min3_f32 (v_max3
//     for dilation) on fixed registers.  Completed rows are stored one batch late (coalesced).
//   - The dilation instance fuses progressive_filter's flag step (neilpy.py:1671-1674).
// No MFMA: there is no contraction in this computation; the kernel is bound by VALU min/max issue
// and LDS reads, HBM traffic is ~2 plane passes per launch.
#pragma once
#include <algorithm>
#include <utility>

#include "smrf_common.h"

// lookup groups in flight: a third buffer (+16 VGPRs in fp32) only where the kernel sits at 2 waves/SIMD
// anyway (demand above 168 VGPRs) and the extra registers do not cost a wave
// lookup groups in flight.  Measured in both rounds: a third group in flight gains nothing at any radius and costs a
// wave at some (R = 28: 0.873 vs 0.838 ms, gpurun_out/r02/probe_exp2.log) - lookup latency is not what the consume
// phase waits for.
#ifndef SMRF_RING_DEPTH
#define SMRF_RING_DEPTH(need) 2
#endif
#ifndef SMRF_RING_BUILD_PRIO
#define SMRF_RING_BUILD_PRIO 3
#endif
#ifndef SMRF_RING_LOOKUP_PRIO
#define SMRF_RING_LOOKUP_PRIO 1
#endif
// columns (= lanes) per workgroup: 256 (shared table, barriers); tuning builds may fix 64 (wave-private table) or 512
// (fp32 only) with -DSMRF_RING_TW=n
#ifdef SMRF_RING_TW
#define SMRF_RING_TW_OF(T, R) SMRF_RING_TW
#else
#define SMRF_RING_TW_OF(T, R) ring_tuned_tw<T>(R)
#endif
// lookups per pipelined group: 4 for the small disks; 2 or 3 for the large ones, whose ring leaves
// few registers for lookups in flight (measured per radius: tools/ring_tune.py --variants cur,g2,g3,g6)
// fp64: per radius where ring_tune.inc says so (kRingGF64, round 5: the register rule below was written for fp32)
#ifndef SMRF_RING_G
#define SMRF_RING_G(ringregs) ((ringregs) > 132 ? 3 : (ringregs) > 100 ? 2 : 4)
#define SMRF_RING_G_OF(T, R, ringregs) (ring_tuned_g<T>(R) > 0 ? ring_tuned_g<T>(R) : SMRF_RING_G(ringregs))
#else
#define SMRF_RING_G_OF(T, R, ringregs) SMRF_RING_G(ringregs)
#endif
// widths that may be looked up with three reads of the level below instead of building the top table level for them
// (RingCfg::DROP_TOP); per radius, measured (gpurun_out/r02/ring_top3.log)
#ifndef SMRF_RING_TOP3
#define SMRF_RING_TOP3(T, R) ring_tuned_top3<T>(R)
#endif
// tuning builds (-DSMRF_RING_NO_TUNE): most row pairs per batch the chooser ring_np() may pick
// (fp32; fp64 cells are twice as large: half) for every radius; product builds read ring_tune.inc
#ifndef SMRF_RING_NP_MAX
#define SMRF_RING_NP_MAX 2
#endif
#ifndef SMRF_RING_XCD_REMAP
#define SMRF_RING_XCD_REMAP 1   // XCD-aware tile placement (see ring_kernel)
#endif
#ifndef SMRF_RING_SLOPE_DEFAULT
#define SMRF_RING_SLOPE_DEFAULT 60  // permille of segment length per residency class (ring_launch_np); SMRF_RING_SLOPE overrides
#endif
#ifndef SMRF_RING_OCC_DROP
#define SMRF_RING_OCC_DROP 0   // tuning builds: run every radius one occupancy step below the estimate
#endif
// incremental window widths (RingCfg::INC): per radius from ring_inc.inc, or everywhere / nowhere in tuning builds
#ifndef SMRF_RING_INC
#define SMRF_RING_INC(T, R) ring_tuned_inc<T>(R)
#endif
// the 2R halo cells of a staged row (beyond the 256 under the lanes) shared out over all waves of the workgroup,
// row pair by row pair, instead of all of them falling to the first waves (ring_kernel, HaloCfg)
#ifndef SMRF_RING_BAL
#define SMRF_RING_BAL 1
#endif
// highest table level of the incremental widths (RingCfg::INC): 3 (eight cells per read), or 2 where no width step of the
// disk is longer than 8 cells (two reads of level 2 per side cover it): one level less to build and to hold in LDS
#ifndef SMRF_RING_JCAP
#define SMRF_RING_JCAP(T, R) ring_tuned_jcap<T>(R)
#endif
// second half of the ring updated in place (RingCfg::INPLACE): per radius from ring_inpl.inc, or everywhere / nowhere in
// tuning builds; SLACK: registers added to the kernel's demand estimate (tuning the occupancy step it is built for)
#ifndef SMRF_RING_INPLACE
#define SMRF_RING_INPLACE(T, R) ring_tuned_inplace<T>(R)
#endif
#ifndef SMRF_RING_BASE_PB
#define SMRF_RING_BASE_PB(T, R) 0   // 0: the default of RingCfg::BASE_PB
#endif
#ifndef SMRF_RING_INPLACE_SLACK
#define SMRF_RING_INPLACE_SLACK 0
#endif
// in-place instances from R = 39 up (the 3-wave ones whose register budget is the point; R = 15..17 and 36 measured
// +-0 ... +2.6 percent with either and keep round 3's first form): smrf_rare, and the turn-back of the ring as asm moves
// in-place instances: lookups per group / groups in flight (0: as the shifting ring, SMRF_RING_G / SMRF_RING_DEPTH)
#ifndef SMRF_RING_INPLACE_G
#define SMRF_RING_INPLACE_G(T, R) ring_tuned_inplace_g<T>(R)
#endif
#ifndef SMRF_RING_INPLACE_D
#define SMRF_RING_INPLACE_D(T, R) 0
#endif
#ifndef SMRF_RING_RARE_OPAQUE
#define SMRF_RING_RARE_OPAQUE(T, R) ((R) >= 39 || ring_tuned_inplace_occ<T>(R) == 4)
#endif
#ifndef SMRF_RING_TURN_ASM
#define SMRF_RING_TURN_ASM(T, R) ((R) >= 39 || ring_tuned_inplace_occ<T>(R) == 4)
#endif
// in-place kernels: waves per SIMD they are built for and most row pairs per batch (per radius from ring_inpl.inc)
#ifndef SMRF_RING_INPLACE_OCC
#define SMRF_RING_INPLACE_OCC(T, R) ring_tuned_inplace_occ<T>(R)
#endif
#ifndef SMRF_RING_INPLACE_NP
#define SMRF_RING_INPLACE_NP(T, R) ring_tuned_inplace_np<T>(R)
#endif
// per-radius overrides of ring_tune.inc's two knobs in single-knob tuning builds (product builds read the tables)
#ifndef SMRF_RING_OCC_DROP_OF
#define SMRF_RING_OCC_DROP_OF(T, R) ring_tuned_occ_drop<T>(R)
#endif
#ifndef SMRF_RING_NP_MAX_OF
#define SMRF_RING_NP_MAX_OF(T, R) ring_tuned_np_max<T>(R)
#endif
#ifndef SMRF_FORCE_OCC
#define SMRF_OCC_OVERRIDE(...) __VA_ARGS__
#else
#define SMRF_OCC_OVERRIDE(...) SMRF_FORCE_OCC
#endif

// tools/isa_budget.py builds single instances with -DSMRF_ISA_MARK: comment lines in the assembly that name the phase
// the instructions after them belong to (the min / max and LDS instructions are volatile asm and keep their order against
// the marks; address arithmetic may float by a few instructions).  Nothing in a product build.
#ifdef SMRF_ISA_MARK
#define SMRF_MARK(s) asm volatile("; SMRF_MARK " s)
#else
#define SMRF_MARK(s) do { } while (0)
#endif

namespace smrf {

// one step down the occupancy ladder the kernels are built for (waves per SIMD)
constexpr int ring_occ_drop(int occ, int steps) {
  for (; steps > 0; --steps) occ = occ > 5 ? 5 : occ > 1 ? occ - 1 : 1;
  return occ;
}

#include "ring_tune.inc"
#include "ring_bal.inc"
#include "ring_inc.inc"
#include "ring_inpl.inc"
#include "ring_buf.inc"
#include "ring_fuse.inc"
// Highest table level of the incremental widths per radius (RingCfg::INC, SMRF_RING_JCAP).  Level 3 (eight cells per
// read) is only ever read for the FIRST width step of a disk (isqrt(2R - 1) cells: 5..11 for R = 15..64); with the cap at 2
// that step takes two (three from R = 41) reads of level 2 per side and one or two more min / max, and level 3 is neither
// built (4 of the 7 reads and 4 of the 8 instructions of the build per cell and row pair) nor held in LDS.  fp32, two-pass
// windows, 16384^2 / 4096^2 (profiles/r03_logs/jcap2_all_radii.log): -1...-6 % at most radii from 15 to 58; the radii left
// at 3 lose with it (29, 31, 50; 59 and up spill); the fused kernels (R <= 14) and fp64 keep 3 (R = 14 fused: +24 %).
// Round 4, builds that differ in these two instances only (profiles/r04_logs/jcap2_r46_r50_ab.log): R = 46 -2.9 % at 2
// (2.206 -> 2.143 ms per window, taken), R = 50 +8.7 % (stays at 3).
template <typename T> constexpr int ring_tuned_jcap(int r) {
  if (sizeof(T) != 4 || r < 15 || r > 58) return 3;
  return (r == 29 || r == 31 || r == 50) ? 3 : 2;
}

// columns per workgroup per radius: 256 everywhere (512-column workgroups, one per CU, measured within +-1 % of 256 at
// R >= 39 and 20-25 % slower below: gpurun_out/r02/probe_tw512.log)
template <typename T> constexpr int ring_tuned_tw(int) { return 256; }

constexpr int clog2(int v) {  // floor(log2(v)), v >= 1
  int l = 0;
  while ((2 << l) <= v) ++l;
  return l;
}

// Compile-time description of skimage's disk(R): half-width per row, the ascending list of
// distinct half-widths and, for every |dy|, the index of its half-width in that list.
template <int R>
struct DiskTables {
  int halfw[R + 1];   // halfw[dy] = isqrt(R*R - dy*dy), dy in [0, R]
  int kidx[R + 1];    // index of halfw[dy] among the distinct half-widths (ascending, 0 for dy = R)
  int wk[R + 1];      // distinct half-widths, ascending (first K entries valid)
  int K;
  constexpr DiskTables() : halfw{}, kidx{}, wk{}, K(0) {
    for (int d = 0; d <= R; ++d) halfw[d] = smrf_isqrt(R * R - d * d);
    int k = 0;
    kidx[R] = 0;
    wk[0] = halfw[R];
    for (int d = R - 1; d >= 0; --d) {
      if (halfw[d] != halfw[d + 1]) { ++k; wk[k] = halfw[d]; }
      kidx[d] = k;
    }
    K = k + 1;
  }
};

template <int R>
struct DiskShape {
  static constexpr DiskTables<R> tab{};
  static constexpr int kidx(int dy) { return tab.kidx[dy]; }   // dy in [0, R]
  static constexpr int K = tab.K;                             // number of distinct half-widths
  static constexpr int wk(int k) { return tab.wk[k]; }        // k-th distinct half-width
  static constexpr int J = clog2(2 * R + 1);                  // highest table level
};

template <bool DIL>
__device__ __forceinline__ float op2(float a, float b) {
  float r;
  // single instruction, no canonicalising v_max in front (IEEE minNum/maxNum: a NaN operand loses)
  if constexpr (DIL) asm volatile("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm volatile("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool DIL>
__device__ __forceinline__ double op2(double a, double b) {
  double r;
  if constexpr (DIL) asm volatile("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm volatile("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <typename T> __device__ __forceinline__ T ident(bool dil);
template <> __device__ __forceinline__ float ident<float>(bool dil) { return dil ? -INFINITY : INFINITY; }
template <> __device__ __forceinline__ double ident<double>(bool dil) { return dil ? -(double)INFINITY : (double)INFINITY; }

template <typename T> __device__ __forceinline__ T qnan();
template <> __device__ __forceinline__ float qnan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double qnan<double>() { return __builtin_nan(""); }

template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct Vec2<double> { typedef double type __attribute__((ext_vector_type(2))); };

template <bool DIL>
__device__ __forceinline__ float op3(float a, float b, float c) {
  float r;
  if constexpr (DIL) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  else asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
template <bool DIL>
__device__ __forceinline__ double op3(double a, double b, double c) { return op2<DIL>(op2<DIL>(a, b), c); }

// acc = op(acc, b, c) with the accumulator TIED to one register (the in-place half of the ring: left to itself the
// register allocator writes every result to a fresh register and keeps the ring sliding through the spare ones)
template <bool DIL>
__device__ __forceinline__ void op3_acc(float& acc, float b, float c) {
  if constexpr (DIL) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(b), "v"(c));
  else asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(acc) : "v"(b), "v"(c));
}
template <bool DIL>
__device__ __forceinline__ void op3_acc(double& acc, double b, double c) { acc = op2<DIL>(op2<DIL>(acc, b), c); }

// The window steps of one lookup group as ONE asm statement (fp32, one read per side for every width of the group): width
// k's two minima grown from width k-1's, N widths chained,
//     x[i] = op3(x[i-1], tx[i], u[i]),  y[i] = op3(y[i-1], ty[i], w[i])      (x[-1] = a0, y[-1] = b0)
// ({tx[i], ty[i]} = the ds_read_b64 of width i's left end, rows A and B; {u[i], w[i]} the one of its right end; results in
// early-clobber outputs).  Written one op3 per asm statement, every step that reads the one before it
// costs an `s_nop 0`: hipcc counts no wait states for asm statements (they have no known length) and assumes gfx950's
// dst_sel forwarding hazard between an asm that writes a register and the next asm that reads it - two to three nops per
// group beside 12-14 min / max (15 per 64-cell row at R = 50).  Inside one statement nothing is inserted, and the
// hardware interlocks ordinary VALU results by itself.  Bit-identical (same instructions, same operand order).
// Per radius and instance kind (ring_fuse.inc, written by tools/ring_fuse_inc.py from the compiled instances' register
// counts and same-process timings): bit 0 = the fused window steps, bit 1 = the ring slots a group releases are updated one
// group LATE, after the next group's window steps - they then read registers written before that group's (real)
// s_waitcnt and need no nop either, at the price of one more group's widths staying live.
#ifndef SMRF_RING_FUSE_MODE
#define SMRF_RING_FUSE_MODE(T, R, INPL) ring_tuned_fuse<T>(R, INPL)
#endif
// (A first form tied each result to the register its left read arrived in - "+v" on a copy of the ds_read's half - and
// failed on the GPU at a dozen radii: hipcc materialised some of those copies as real v_mov instructions and scheduled
// them above the s_waitcnt, reading a register the LDS was still filling.  Nothing but asm statements may read a ds_read's
// destination, and none through a tied operand: the results are early-clobber outputs, the reads plain inputs.)
#define SMRF_RED_STEP(OP, x, y, px, py, tx, ty, u, w) OP " %" #x ", %" #px ", %" #tx ", %" #u "\n\t" OP " %" #y ", %" #py ", %" #ty ", %" #w
template <bool DIL>
__device__ __forceinline__ void red_chain(float a0, float b0, float& x1, float& y1, float tx1, float ty1, float u1, float w1, float& x2,
                                          float& y2, float tx2, float ty2, float u2, float w2) {
  if constexpr (DIL)
    asm volatile(SMRF_RED_STEP("v_max3_f32", 0, 1, 4, 5, 6, 7, 8, 9) "\n\t" SMRF_RED_STEP("v_max3_f32", 2, 3, 0, 1, 10, 11, 12, 13)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2));
  else
    asm volatile(SMRF_RED_STEP("v_min3_f32", 0, 1, 4, 5, 6, 7, 8, 9) "\n\t" SMRF_RED_STEP("v_min3_f32", 2, 3, 0, 1, 10, 11, 12, 13)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2));
}
template <bool DIL>
__device__ __forceinline__ void red_chain(float a0, float b0, float& x1, float& y1, float tx1, float ty1, float u1, float w1, float& x2,
                                          float& y2, float tx2, float ty2, float u2, float w2, float& x3, float& y3, float tx3, float ty3,
                                          float u3, float w3) {
  if constexpr (DIL)
    asm volatile(SMRF_RED_STEP("v_max3_f32", 0, 1, 6, 7, 8, 9, 10, 11) "\n\t" SMRF_RED_STEP("v_max3_f32", 2, 3, 0, 1, 12, 13, 14, 15) "\n\t"
                 SMRF_RED_STEP("v_max3_f32", 4, 5, 2, 3, 16, 17, 18, 19)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2), "=&v"(x3), "=&v"(y3)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2), "v"(tx3), "v"(ty3),
                   "v"(u3), "v"(w3));
  else
    asm volatile(SMRF_RED_STEP("v_min3_f32", 0, 1, 6, 7, 8, 9, 10, 11) "\n\t" SMRF_RED_STEP("v_min3_f32", 2, 3, 0, 1, 12, 13, 14, 15) "\n\t"
                 SMRF_RED_STEP("v_min3_f32", 4, 5, 2, 3, 16, 17, 18, 19)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2), "=&v"(x3), "=&v"(y3)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2), "v"(tx3), "v"(ty3),
                   "v"(u3), "v"(w3));
}
template <bool DIL>
__device__ __forceinline__ void red_chain(float a0, float b0, float& x1, float& y1, float tx1, float ty1, float u1, float w1, float& x2,
                                          float& y2, float tx2, float ty2, float u2, float w2, float& x3, float& y3, float tx3, float ty3,
                                          float u3, float w3, float& x4, float& y4, float tx4, float ty4, float u4, float w4) {
  if constexpr (DIL)
    asm volatile(SMRF_RED_STEP("v_max3_f32", 0, 1, 8, 9, 10, 11, 12, 13) "\n\t" SMRF_RED_STEP("v_max3_f32", 2, 3, 0, 1, 14, 15, 16, 17) "\n\t"
                 SMRF_RED_STEP("v_max3_f32", 4, 5, 2, 3, 18, 19, 20, 21) "\n\t" SMRF_RED_STEP("v_max3_f32", 6, 7, 4, 5, 22, 23, 24, 25)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2), "=&v"(x3), "=&v"(y3), "=&v"(x4), "=&v"(y4)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2), "v"(tx3), "v"(ty3),
                   "v"(u3), "v"(w3), "v"(tx4), "v"(ty4), "v"(u4), "v"(w4));
  else
    asm volatile(SMRF_RED_STEP("v_min3_f32", 0, 1, 8, 9, 10, 11, 12, 13) "\n\t" SMRF_RED_STEP("v_min3_f32", 2, 3, 0, 1, 14, 15, 16, 17) "\n\t"
                 SMRF_RED_STEP("v_min3_f32", 4, 5, 2, 3, 18, 19, 20, 21) "\n\t" SMRF_RED_STEP("v_min3_f32", 6, 7, 4, 5, 22, 23, 24, 25)
                 : "=&v"(x1), "=&v"(y1), "=&v"(x2), "=&v"(y2), "=&v"(x3), "=&v"(y3), "=&v"(x4), "=&v"(y4)
                 : "v"(a0), "v"(b0), "v"(tx1), "v"(ty1), "v"(u1), "v"(w1), "v"(tx2), "v"(ty2), "v"(u2), "v"(w2), "v"(tx3), "v"(ty3),
                   "v"(u3), "v"(w3), "v"(tx4), "v"(ty4), "v"(u4), "v"(w4));
}

// One LDS read of a {row A, row B} cell at byte address `addr + OFF`.  Written as asm so that
// hipcc cannot fuse two of them into ds_read2_b64, which moves half the bytes per LDS cycle of
// ds_read_b64 on gfx950 (MI355X_MICROARCH: 128 vs 256 B/clk).  The caller owns the wait
// (lds_wait<N>) - the compiler does not count asm loads.
template <int OFF>
__device__ __forceinline__ Vec2<float>::type lds_read2(unsigned addr, float) {
  Vec2<float>::type v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int OFF>
__device__ __forceinline__ Vec2<double>::type lds_read2(unsigned addr, double) {
  Vec2<double>::type v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
// One LDS store of a {row A, row B} cell, as asm for the same reason: two stores of a batch's row pairs at one lane
// address and different offsets are otherwise fused into ds_write2st64_b64 (13 cycles against 6 each, MI355X_MICROARCH
// LDS table) - whether hipcc does so flipped with an unrelated refactoring of this file and cost 2-4 % per launch.
// The value is in registers when the instruction issues (the compiler waits for the loads that produce it); the
// caller owns the wait for the store itself (lds_wait<0>() before the barrier that publishes it).
__device__ __forceinline__ void lds_write2(unsigned addr, Vec2<float>::type v) {
  asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_write2(unsigned addr, Vec2<double>::type v) {
  asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
}
// The counted wait.  BI = false: an asm statement, as every round so far.  BI = true (round 5; the consume phase of the
// instances ring_fuse.inc switches on): the compiler's own s_waitcnt.  hipcc's hazard recognizer gives an asm statement no
// length: between an asm that writes a register and a later asm that reads it, it counts only the REAL instructions in
// between, and with none it inserts `s_nop 0` (gfx950's dst_sel forwarding hazard, assumed for any asm) - with the waits as
// asm, the first min / max after every wait got one.  The builtin is a real instruction the recognizer counts; the
// waitcnt insertion pass never weakens or drops it, but it MERGES its own vmcnt waits into it, i.e. moves them earlier:
// the fused openings R = 12..14 measured 2 % slower with builtin waits everywhere, hence the switch per use.
template <int N, bool BI = false>
__device__ __forceinline__ void lds_wait() {   // at most N LDS operations still outstanding
  if constexpr (BI) {
    // s_waitcnt simm16 on gfx9: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt[15:14]; vmcnt = 63, expcnt = 7: not waited for
    __builtin_amdgcn_s_waitcnt(0xC07F | ((N > 15 ? 15 : N) << 8));   // 4-bit counter field
    asm volatile("" ::: "memory");             // and no LDS access of the compiler's own moves across it
  } else {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N > 15 ? 15 : N) : "memory");
  }
}

// Geometry of one kernel instance.  Input rows are handled in PAIRS (A = y, B = y+1): the two
// rows are interleaved cell by cell in LDS ({A, B} per cell), so one ds_read_b64 serves a window
// lookup of both rows and one v_min3 folds both rows into a ring slot.
template <typename T, int R, int TW, int NP>
struct RingCfg {
  using S = DiskShape<R>;
  static constexpr int E = sizeof(T) / 4;
  // Highest table level.  The widest widths of a disk whose 2R + 1 just passes a power of two are the only users of
  // level S::J (R = 32: one width of 30); when there are at most SMRF_RING_TOP3 of them, that level is not built and
  // they are looked up with THREE reads of the level below (min3: the same instruction count per row) - one table
  // level less in LDS and 2^(J-JB-1) fewer reads per cell in the higher-level build.
  static constexpr int n_top() {
    int n = 0;
    for (int k = 1; k < S::K; ++k)
      if (clog2(2 * S::wk(k) + 1) == S::J) ++n;
    return n;
  }
  // INC: the disk's widths taken in ascending order, each window grown from the one before it:
  //     H[w_k] = min3(H[w_k-1], T_j[x - w_k], T_j[x + w_k - 2^j + 1]),  2^j >= w_k - w_k-1  (H[w_0 = 0] = the cell)
  // - one read per side of the smallest table level that spans the step (a read may reach back into the window it
  // extends: min / max is idempotent), two per side where the step is longer than 8 cells (only the first width of
  // a disk with R >= 41).  The same two reads and one instruction per width as a lookup in a full sparse table, but
  // no level above 3 is ever read: the second build phase with its reads and its barrier is gone, a level costs
  // 8 cells of padding instead of 64, and the table of a large disk holds 5 levels instead of 6.
  static constexpr bool INC = SMRF_RING_INC(T, R);
  static constexpr int inc_lev(int k) {                  // table level of the reads for width index k >= 1
    const int d = S::wk(k) - S::wk(k - 1);
    int j = 0;
    while ((1 << j) < d && j < SMRF_RING_JCAP(T, R)) ++j;
    return j;
  }
  static constexpr int inc_n(int k) {                    // reads per side
    const int d = S::wk(k) - S::wk(k - 1), j = inc_lev(k);
    return (d + (1 << j) - 1) >> j;
  }
  static constexpr int inc_jmax() {
    int j = 1;
    for (int k = 1; k < S::K; ++k)
      if (inc_lev(k) > j) j = inc_lev(k);
    return j;
  }
  static constexpr bool DROP_TOP = !INC && S::J >= 3 && n_top() <= SMRF_RING_TOP3(T, R) && 2 * R + 1 <= 3 * (1 << (S::J - 1));
  static constexpr int J = INC ? inc_jmax() : DROP_TOP ? S::J - 1 : S::J;
  static constexpr int lev(int w) { const int j = clog2(2 * w + 1); return j > J ? J : j; }   // level a width is read at
  static constexpr int nreads(int w) { return clog2(2 * w + 1) > J ? 3 : 2; }
  static constexpr int W = TW + 2 * R;                   // staged cells per row
  static constexpr int PAD = 1 << J;                     // furthest build read is < 2^J cells ahead
  static constexpr int WP = ((W + PAD + 3) / 4) * 4;     // pitch of one table level (cells)
  static constexpr int NPOS = (W + TW - 1) / TW;         // staged cells per lane and row
  static constexpr int ROWS = 2 * NP;                    // input rows per batch
  // table levels: level j holds the min/max over 2^j cells starting at the cell.  A lookup of
  // half-width w reads level floor(log2(2w+1)).  Built in two stages of INDEPENDENT reads (one
  // LDS round trip and one barrier each): the base level JB straight from the staged row
  // (2^JB - 1 reads), then every higher level from the base level (stride 2^JB).
  static constexpr bool used(int j) {
    if (j == 0) return true;
    for (int k = 1; k < S::K; ++k)
      if ((INC ? inc_lev(k) : lev(S::wk(k))) == j) return true;
    return false;
  }
  static constexpr int jmin() {                          // lowest level >= 1 that a lookup reads
    for (int j = 1; j <= J; ++j)
      if (used(j)) return j;
    return J;
  }
  static constexpr int JB = INC ? J : jmin() < 3 ? jmin() : 3;   // base level (INC: the only build phase makes every level)
  static constexpr bool stored(int j) {
    if (INC) return j == 0 || (j >= 1 && j <= J && (j == J || used(j)));
    return j == 0 || (j >= JB && j <= J && (j == JB || used(j)));
  }
  // storage index of a stored level; level 0 is double buffered (indices 0 and 1, alternating per
  // batch) so that staging the next batch never races with a slower wave still reading its cells
  static constexpr int slot_of(int j) {
    if (j == 0) return 0;
    int n = 2;
    for (int c = 1; c < j; ++c)
      if (stored(c)) ++n;
    return n;
  }
  static constexpr int NLEV = slot_of(J) + 1;
  static constexpr size_t LDS_BYTES = ((size_t)NP * NLEV * WP + PAD) * 2 * sizeof(T);
  // INPLACE (round 3): the ring's second half (slots R-1 .. 2R-1, output rows below the input row) is updated IN PLACE,
  // the slot <-> register mapping rotating by two per row pair (compile-time inside a batch) and one register rotation
  // per batch, so that BOTH halves consume the window widths in the ascending order the lookups produce them and
  // only the last few widths stay live (about 2K - 2G registers less than the shifting second half, which walks the
  // widths downwards and so keeps all K of both rows until the end of the pair).  See ring_consume_inplace.
  // Where ring_inpl.inc marks a radius DUAL both forms exist - the in-place instance is the one with the table's row
  // pairs per batch, any other NP is the shifting ring - and ring_launch takes the in-place one for long segments only.
  static constexpr bool INPLACE = INC && SMRF_RING_INPLACE(T, R) &&
                                  (!ring_tuned_inplace_dual<T>(R) || NP == SMRF_RING_INPLACE_NP(T, R));
  static constexpr int G = INPLACE && SMRF_RING_INPLACE_G(T, R) > 0 ? SMRF_RING_INPLACE_G(T, R)
                                                                     : SMRF_RING_G_OF(T, R, E * (2 * R + 2 * S::K));   // window lookups per pipelined group
  static constexpr int NG = (S::K - 1 + G - 1) / G;      // groups for k = 1..K-1
  static constexpr int gsize(int g) { int n = S::K - 1 - g * G; return n < 0 ? 0 : (n > G ? G : n); }
  static constexpr int BASE_PB = SMRF_RING_BASE_PB(T, R) > 0 ? SMRF_RING_BASE_PB(T, R) : INPLACE ? 1 : 8;   // row pairs per round trip of the base-level build (in-place kernels are built for registers)
  static constexpr int NEED_BASE = INPLACE ? E * (2 * R + 2 * (G + 2) + 3 + 20 + 4 * G) + 16 + SMRF_RING_INPLACE_SLACK
                                           : E * (2 * R + 2 * S::K + 20 + 4 * G) + 16;   // measured VGPR demand at D = 2
  static constexpr int D = INPLACE && SMRF_RING_INPLACE_D(T, R) > 0 ? SMRF_RING_INPLACE_D(T, R) : SMRF_RING_DEPTH(NEED_BASE);   // lookup groups kept in flight
  static constexpr int greads(int g) {                   // LDS reads of lookup group g
    int n = 0;
    for (int k = 1 + g * G; k < 1 + (g + 1) * G && k < S::K; ++k) n += INC ? 2 * inc_n(k) : nreads(S::wk(k));
    return n;
  }
  static constexpr int inflight_after(int g) {           // reads of groups g+1 .. g+D-1
    int n = 0;
    for (int i = 1; i < D; ++i) n += greads(g + i);
    return n;
  }
  // lookup group g's window steps as one asm statement (red_chain): fp32, incremental widths, 2..4 widths, one read per side each
  static constexpr int FUSE_MODE = INC && sizeof(T) == 4 ? SMRF_RING_FUSE_MODE(T, R, INPLACE) : 0;
  static constexpr bool fused_red(int g) {
    if (!(FUSE_MODE & 1)) return false;
    const int n = gsize(g);
    if (n < 2 || n > 4) return false;
    for (int k = 1 + g * G; k < 1 + g * G + n; ++k)
      if (inc_n(k) != 1) return false;
    return true;
  }
  static constexpr bool SLOT_DELAY = (FUSE_MODE & 2) != 0;   // a group's ring slots updated after the NEXT group's window steps
  static constexpr int NEED = NEED_BASE + (D - 2) * 4 * G * E;
  static constexpr int OCC_EST = NEED <= 64 ? 8 : NEED <= 96 ? 5 : NEED <= 128 ? 4 : NEED <= 168 ? 3 : NEED <= 264 ? 2 : 1;
  static constexpr int OCC_REG = INPLACE && SMRF_RING_INPLACE_OCC(T, R) > 0 ? SMRF_RING_INPLACE_OCC(T, R)
                                                                          : ring_occ_drop(OCC_EST, SMRF_RING_OCC_DROP_OF(T, R));
  static constexpr int WAVES = TW / 64;                  // waves per workgroup
  static constexpr int WG_LDS = (int)(160 * 1024 / LDS_BYTES) < 1 ? 1 : (int)(160 * 1024 / LDS_BYTES);
  static constexpr int OCC_LDS = WG_LDS * WAVES / 4 < 1 ? 1 : WG_LDS * WAVES / 4;
  static constexpr int OCC = OCC_REG < OCC_LDS ? OCC_REG : OCC_LDS;   // waves per SIMD the kernel is built for
  // ring slot s (0 .. 2R-3) after a pair takes min3(acc[s+2], RA[kA(s)], RB[kB(s)])
  static constexpr int kA(int s) { int d = R - s - 2; return S::kidx(d < 0 ? -d : d); }
  static constexpr int kB(int s) { int d = R - s - 1; return S::kidx(d < 0 ? -d : d); }
  // lookup group after which slot s can be updated (first half: as soon as its widths are there;
  // second half: after the last group).  Group g covers k in [1 + g*G, 1 + (g+1)*G).
  static constexpr int slot_group(int s) {
    if (s > R - 2) return NG;                            // second half (and the middle): at the end
    const int k = kA(s);                                 // kA >= kB in the first half
    return k == 0 ? 0 : (k - 1) / G;
  }
};

// Row pairs per batch: the most (up to SMRF_RING_NP_MAX) whose tables still leave room in the
// CU's 160 KB of LDS for every workgroup the register budget allows.  More pairs per batch spread
// the three barriers and the table build's latency over more output rows.
template <typename T, int R, int TW>
constexpr int ring_np() {
  using C1 = RingCfg<T, R, TW, 1>;
  constexpr int mx = C1::INPLACE && SMRF_RING_INPLACE_NP(T, R) > 0 ? SMRF_RING_INPLACE_NP(T, R)
                     : sizeof(T) == 4 ? SMRF_RING_NP_MAX_OF(T, R) : (ring_tuned_np_max<T>(R) / 2 < 1 ? 1 : ring_tuned_np_max<T>(R) / 2);
  constexpr int want = C1::OCC_REG * 4 / C1::WAVES < 1 ? 1 : C1::OCC_REG * 4 / C1::WAVES;   // workgroups per CU
  if constexpr (mx >= 4) if (RingCfg<T, R, TW, 4>::WG_LDS >= want) return 4;
  if constexpr (mx >= 3) if (RingCfg<T, R, TW, 3>::WG_LDS >= want) return 3;
  if constexpr (mx >= 2) if (RingCfg<T, R, TW, 2>::WG_LDS >= want) return 2;
  return 1;
}
#define SMRF_RING_NP(T, R) ring_np<T, R, SMRF_RING_TW_OF(T, R)>()


// Buffer addressing (SMRF_RING_BUF): the common-case loads and stores of the ring kernel as buffer instructions - a
// resource descriptor in 4 SGPRs per plane (based at the first row the workgroup touches), the row as a scalar byte
// offset, the lane's column as one 32-bit VGPR offset computed once - instead of a 64-bit address per lane and access
// (v_lshl_add_u64, 4.1 cycles each, ~40 per batch in round 2's kernels, and a VGPR pair each while in flight).
#ifndef SMRF_RING_BUF
#define SMRF_RING_BUF(T, R) ring_tuned_buf<T>(R)
#endif
// ... but not in the in-place instances of R = 47, 49 (3 waves per SIMD): there the descriptors' offsets cost the registers
// the instance was built to save (20 / 36 B of scratch with, 0 / 12 B without)
#ifndef SMRF_RING_INPLACE_NOBUF
#define SMRF_RING_INPLACE_NOBUF(T, R) ((R) > 45)
#endif
template <typename T, int R, int TW, int NP>
constexpr bool ring_buf_on() { return SMRF_RING_BUF(T, R) && !(RingCfg<T, R, TW, NP>::INPLACE && SMRF_RING_INPLACE_NOBUF(T, R)); }
// a lane's column index made opaque on the rare paths (reflected / clamped rows, the NaN rule, segment ends) of the in-place
// instances: without it hipcc hoists their loop-invariant 64-bit addresses (plane + column) out of the row loop and keeps a
// VGPR pair per plane alive through the consume phase for code that runs on a few batches per segment - registers the
// 3-wave instances do not have (R = 47, 50: 44 / 20 B of scratch without, none with).  The shifting instances keep the
// hoisted form (R = 26..34 measured 1.5-3 percent slower with the opaque one, profiles/r03_logs/turn2_ab.log).
template <bool ON>
__device__ __forceinline__ int smrf_rare(int v) {
  if constexpr (ON) asm volatile("" : "+v"(v));
  return v;
}
using smrf_rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ smrf_rsrc_t smrf_make_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}
template <bool NT = false>
__device__ __forceinline__ float smrf_buf_load(smrf_rsrc_t r, unsigned voff, unsigned soff, float) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, NT ? 2 : 0));
}
template <bool NT = false>
__device__ __forceinline__ double smrf_buf_load(smrf_rsrc_t r, unsigned voff, unsigned soff, double) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, NT ? 2 : 0));
}
// `last` of the flag step in the ring dilation: read once per launch, never again - a streaming load (SMRF_NT_LAST)
#ifndef SMRF_NT_LAST
#define SMRF_NT_LAST 0
#endif
template <bool NT>
__device__ __forceinline__ void smrf_buf_store(smrf_rsrc_t r, unsigned voff, unsigned soff, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)voff, (int)soff, NT ? 2 : 0);
}
template <bool NT>
__device__ __forceinline__ void smrf_buf_store(smrf_rsrc_t r, unsigned voff, unsigned soff, double v) {
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, v), r, (int)voff, (int)soff, NT ? 2 : 0);
}
__device__ __forceinline__ void smrf_buf_store_u8(smrf_rsrc_t r, unsigned voff, unsigned soff, uint8_t v) {
  __builtin_amdgcn_raw_buffer_store_b8(v, r, (int)voff, (int)soff, 0);
}

// A completed output cell.  nt: as a non-temporal (streaming) store - the output plane is not read again before the
// next launch, and kept out of the L2 it leaves the lines the kernels DO re-read there (the ring kernels share 2R halo
// columns between neighbouring strips).  Measured on the 16384^2 benchmark: -1.7 % on the whole call, -3...-8 % on the
// memory-bound windows; on a 4096^2 raster (64 MB planes, which the 256 MB infinity cache hands to the next launch)
// +2 %, so the host sets DiskArgs::nt by plane size (morph.hip).
template <typename T>
__device__ __forceinline__ void smrf_store_out(T* p, T v, int nt) {
  if (nt) __builtin_nontemporal_store(v, p);
  else *p = v;
}
// the flag step of one output cell (neilpy.py:1671-1674): sparse (only flagged cells are written, the planes were
// cleared) or dense (every byte written: the first window of a call, whose planes then need no clearing)
template <typename T>
__device__ __forceinline__ void smrf_flag_cell(const DiskArgs<T>& a, long long off, T lastval, T val) {
  const T diff = lastval - val;                            // raster dtype
  bool hit;                                                // float64 comparison (NumPy 2); fp32: smrf_float_below
  if constexpr (sizeof(T) == 4) hit = diff > a.thr_lo;
  else hit = (double)diff > a.thr;
  if (a.dense) {
    a.mask[off] = hit ? 1 : 0;
    if (a.when != nullptr) a.when[off] = hit ? (uint8_t)a.widx : (uint8_t)0;
  } else if (hit) {
    a.mask[off] = 1;
    if (a.when != nullptr) a.when[off] = (uint8_t)a.widx;
  }
}

// reflect-folded local row index of consecutive global rows without a division per row
struct RowFold {
  int n, p;                                              // p = y mod 2n in [0, 2n)
  __device__ __forceinline__ RowFold(int y, int n_) : n(n_) {
    p = y % (2 * n);
    if (p < 0) p += 2 * n;
  }
  __device__ __forceinline__ int at(int i) const {       // fold(y + i), i >= 0
    int q = p + i;
    while (q >= 2 * n) q -= 2 * n;
    return q < n ? q : 2 * n - 1 - q;
  }
  __device__ __forceinline__ void advance(int d) {
    p += d;
    while (p >= 2 * n) p -= 2 * n;
  }
};

// One batch of a ring stage (NP row pairs whose level-0 cells are staged and visible to the whole workgroup) in three
// phases, a workgroup barrier between them: ring_base (base table level), ring_upper (higher levels), ring_consume
// (per pair the window lookups and the ring update).  Shared by the single-pass kernels and by both stages of the
// fused opening (morph_fused.h), which runs its two stages' phases side by side under the same three barriers.
//   v[p][i]  the lane's own staged cells of pair p (cell OFF + tid + i * TW); NPB of them per pair are built
//   acc      the ring (2R partial output rows), outv <- the 2 * NP rows this batch completes
//   OFF = 0, NPB = NPOS: every staged cell (TW + 2R per row).  OFF = R, NPB = 1: only the TW cells under the lanes
//   (the fused opening's second stage, whose level 0 holds the first stage's TW eroded columns)
template <typename T, int R, bool DIL, int TW, int NP, int NPB, int OFF>
__device__ __forceinline__ void ring_base(typename Vec2<T>::type* const L, const int par, const int tid, const bool has_last,
                                          typename Vec2<T>::type (&v)[NP][RingCfg<T, R, TW, NP>::NPOS]) {
  using C = RingCfg<T, R, TW, NP>;
  using T2 = typename Vec2<T>::type;
  constexpr int WP = C::WP, NLEV = C::NLEV;
  constexpr int JB = C::JB, SB = C::slot_of(JB);
  // (2) base level JB from level 0: 2^JB - 1 independent reads per cell.  All reads of the
  //     batch are issued first (asm ds_read_b64: hipcc would fuse them into half-rate
  //     ds_read2_b64), then one wait, then the min/max and the writes.
  constexpr int NA = (1 << JB) - 1;
  const unsigned lds_l = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + tid + OFF);
  // PB row pairs per round trip: all NP of them (one wait per cell position) unless the kernel is built for registers
  // (RingCfg::BASE_PB: the reads of a round trip are 2 * NA registers per pair)
  constexpr int PB = C::BASE_PB < NP ? C::BASE_PB : NP;
#pragma unroll
  for (int i = 0; i < NPB; ++i) {
    if (i < NPB - 1 || has_last) {
     const int pos = tid + OFF + i * TW;
#pragma unroll
     for (int p0 = 0; p0 < NP; p0 += PB) {
      T2 na[NP][NA];
#pragma unroll
      for (int p = p0; p < p0 + PB && p < NP; ++p) {
        const unsigned ad = lds_l + ((p * NLEV + par) * WP + i * TW) * (unsigned)sizeof(T2);
        [&]<int... Kk>(std::integer_sequence<int, Kk...>) {
          ((na[p][Kk] = lds_read2<(Kk + 1) * (int)sizeof(T2)>(ad, T())), ...);
        }(std::make_integer_sequence<int, NA>{});
      }
      lds_wait<0>();
#pragma unroll
      for (int p = p0; p < p0 + PB && p < NP; ++p) {
        const T2* n = na[p];
        T2 m = v[p][i];
        if constexpr (C::INC) {
          // every level a width step reads, on the way up to level JB (the same instruction count)
          m.x = op2<DIL>(m.x, n[0].x); m.y = op2<DIL>(m.y, n[0].y);
          if constexpr (JB >= 2) {
            if constexpr (C::stored(1)) { constexpr int s1 = C::slot_of(1); L[(p * NLEV + s1) * WP + pos] = m; }
            m.x = op3<DIL>(m.x, n[1].x, n[2].x); m.y = op3<DIL>(m.y, n[1].y, n[2].y);
          }
          if constexpr (JB == 3) {
            if constexpr (C::stored(2)) { constexpr int s2 = C::slot_of(2); L[(p * NLEV + s2) * WP + pos] = m; }
            m.x = op3<DIL>(m.x, n[3].x, n[4].x); m.y = op3<DIL>(m.y, n[3].y, n[4].y);
            m.x = op3<DIL>(m.x, n[5].x, n[6].x); m.y = op3<DIL>(m.y, n[5].y, n[6].y);
          }
        } else {
        if constexpr (JB == 1) { m.x = op2<DIL>(m.x, n[0].x); m.y = op2<DIL>(m.y, n[0].y); }
        if constexpr (JB >= 2) {
          m.x = op3<DIL>(m.x, n[0].x, n[1].x); m.y = op3<DIL>(m.y, n[0].y, n[1].y);
          m.x = op2<DIL>(m.x, n[2].x); m.y = op2<DIL>(m.y, n[2].y);
        }
        if constexpr (JB == 3) {
          m.x = op3<DIL>(m.x, n[3].x, n[4].x); m.y = op3<DIL>(m.y, n[3].y, n[4].y);
          m.x = op3<DIL>(m.x, n[5].x, n[6].x); m.y = op3<DIL>(m.y, n[5].y, n[6].y);
        }
        }
        v[p][i] = m;
        L[(p * NLEV + SB) * WP + pos] = m;
      }
      __builtin_amdgcn_sched_barrier(0);
     }
    }
  }
}

template <typename T, int R, bool DIL, int TW, int NP, int NPB, int OFF>
__device__ __forceinline__ void ring_upper(typename Vec2<T>::type* const L, const int tid, const bool has_last,
                                           typename Vec2<T>::type (&v)[NP][RingCfg<T, R, TW, NP>::NPOS]) {
  using C = RingCfg<T, R, TW, NP>;
  using S = typename C::S;
  using T2 = typename Vec2<T>::type;
  constexpr int J = C::J, WP = C::WP, NLEV = C::NLEV;
  constexpr int JB = C::JB, SB = C::slot_of(JB);
  const unsigned lds_l = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + tid + OFF);
  // (3) levels JB+1 .. J from the base level: cells pos + k * 2^JB, k < 2^(J-JB).  One job per (cell position,
  //     row pair): NB independent reads, then the min/max chain and the stores.  Jobs are software-pipelined two
  //     deep (the next job's reads are issued before this job's wait), within the 15 LDS operations a wave may have
  //     outstanding; round 1 drained every job on its own and spent 14 % of the R = 50 wave time here.
  if constexpr (J > JB) {
    constexpr int NB = (1 << (J - JB)) - 1;
    constexpr int NBUF = (2 * NB <= 15 && NP > 1) ? 2 : 1;
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      if (i < NPB - 1 || has_last) {
        const int pos = tid + OFF + i * TW;
        T2 nb[NBUF][NB];
        auto issue_job = [&]<int P>(std::integral_constant<int, P>) {
          const unsigned ad = lds_l + ((P * NLEV + SB) * WP + i * TW) * (unsigned)sizeof(T2);
          [&]<int... Kk>(std::integer_sequence<int, Kk...>) {
            ((nb[P % NBUF][Kk] = lds_read2<((Kk + 1) << JB) * (int)sizeof(T2)>(ad, T())), ...);
          }(std::make_integer_sequence<int, NB>{});
        };
        auto finish_job = [&]<int P>(std::integral_constant<int, P>) {
          const T2* n = nb[P % NBUF];
          T2 m = v[P][i];
          [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
            (([&] {
               constexpr int j = JB + 1 + JJ;            // level being completed
               constexpr int k0 = 1 << (j - 1 - JB);     // new cells k0 .. 2*k0 - 1
               if constexpr (k0 == 1) { m.x = op2<DIL>(m.x, n[0].x); m.y = op2<DIL>(m.y, n[0].y); }
               else {
#pragma unroll
                 for (int k = k0; k < 2 * k0; k += 2) {
                   m.x = op3<DIL>(m.x, n[k - 1].x, n[k].x); m.y = op3<DIL>(m.y, n[k - 1].y, n[k].y);
                 }
               }
               if constexpr (C::stored(j)) {
                 constexpr int sj = C::slot_of(j);
                 L[(P * NLEV + sj) * WP + pos] = m;
               }
             }()), ...);
          }(std::make_integer_sequence<int, J - JB>{});
        };
        if constexpr (NBUF == 2) issue_job(std::integral_constant<int, 0>{});
        [&]<int... P>(std::integer_sequence<int, P...>) {
          (([&] {
             if constexpr (NBUF == 2) {
               // the stores of job P-1 were issued before these reads: in-order completion covers them too
               if constexpr (P + 1 < NP) issue_job(std::integral_constant<int, P + 1>{});
               lds_wait<(P + 1 < NP ? NB : 0)>();
             } else {
               issue_job(std::integral_constant<int, P>{});
               lds_wait<0>();
             }
             finish_job(std::integral_constant<int, P>{});
             __builtin_amdgcn_sched_barrier(0);
           }()), ...);
        }(std::make_integer_sequence<int, NP>{});
      }
    }
  }
}

// ---- the halo cells of a batch, shared out over the waves (ring_kernel) -------------------------------------------
// A staged row is TW + 2R cells wide.  With one cell per lane plus "cell TW + tid if there is one", the 2R halo cells of
// EVERY row pair fall to the first one or two waves, which then build twice the table cells of the others and keep the
// whole workgroup at each phase barrier.  Here the halo of a batch is cut into wave-jobs (row pair p, 64-cell part) and
// wave w takes jobs w, w + WAVES, ...: with 4 row pairs and 2R <= 64 one job per wave, with 2R > 64 two.  A lane's halo
// cell is the same in all its jobs (WAVES is a multiple of the parts per row), only the row pair changes.
template <typename T, int R, int TW, int NP>
struct HaloCfg {
  using C = RingCfg<T, R, TW, NP>;
  static constexpr int HW = C::W - TW;                   // halo cells per row
  static constexpr int WAVES = TW / 64;
  static constexpr int NH = (HW + 63) / 64;              // 64-cell parts per row
  static constexpr int NQ = NP * NH;                     // wave-jobs per batch
  static constexpr int NJ = (NQ + WAVES - 1) / WAVES;    // most jobs one wave takes
  static constexpr bool OK = SMRF_RING_BAL && ring_tuned_bal<T>(R) && C::NPOS == 2 && TW % 64 == 0 && WAVES >= 2 && WAVES % NH == 0;
};
struct HaloLane {
  int wave;      // wave index in the workgroup (uniform)
  bool act;      // this lane has a halo cell
  int pos;       // its staged cell (TW + h), or a cell that is safe to read when it has none
};

template <typename T, int R, bool DIL, int TW, int NP>
__device__ __forceinline__ void ring_base_halo(typename Vec2<T>::type* const L, const int par, const HaloLane hl,
                                               typename Vec2<T>::type (&vh)[HaloCfg<T, R, TW, NP>::NJ]) {
  using C = RingCfg<T, R, TW, NP>;
  using H = HaloCfg<T, R, TW, NP>;
  using T2 = typename Vec2<T>::type;
  constexpr int WP = C::WP, NLEV = C::NLEV, JB = C::JB, SB = C::slot_of(JB);
  constexpr int NA = (1 << JB) - 1;
  const unsigned lds_h = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + hl.pos);
  if (hl.act) {
    T2 na[H::NJ][NA];
#pragma unroll
    for (int j = 0; j < H::NJ; ++j) {
      const int q = hl.wave + H::WAVES * j;
      if (q < H::NQ) {
        const unsigned ad = lds_h + (unsigned)(((q / H::NH) * NLEV + par) * WP) * (unsigned)sizeof(T2);
        [&]<int... Kk>(std::integer_sequence<int, Kk...>) {
          ((na[j][Kk] = lds_read2<(Kk + 1) * (int)sizeof(T2)>(ad, T())), ...);
        }(std::make_integer_sequence<int, NA>{});
      }
    }
    lds_wait<0>();
#pragma unroll
    for (int j = 0; j < H::NJ; ++j) {
      const int q = hl.wave + H::WAVES * j;
      if (q < H::NQ) {
        const T2* n = na[j];
        T2 m = vh[j];
        const int pq = q / H::NH;
        if constexpr (C::INC) {
          m.x = op2<DIL>(m.x, n[0].x); m.y = op2<DIL>(m.y, n[0].y);
          if constexpr (JB >= 2) {
            if constexpr (C::stored(1)) { constexpr int s1 = C::slot_of(1); L[(pq * NLEV + s1) * WP + hl.pos] = m; }
            m.x = op3<DIL>(m.x, n[1].x, n[2].x); m.y = op3<DIL>(m.y, n[1].y, n[2].y);
          }
          if constexpr (JB == 3) {
            if constexpr (C::stored(2)) { constexpr int s2 = C::slot_of(2); L[(pq * NLEV + s2) * WP + hl.pos] = m; }
            m.x = op3<DIL>(m.x, n[3].x, n[4].x); m.y = op3<DIL>(m.y, n[3].y, n[4].y);
            m.x = op3<DIL>(m.x, n[5].x, n[6].x); m.y = op3<DIL>(m.y, n[5].y, n[6].y);
          }
        } else {
        if constexpr (JB == 1) { m.x = op2<DIL>(m.x, n[0].x); m.y = op2<DIL>(m.y, n[0].y); }
        if constexpr (JB >= 2) {
          m.x = op3<DIL>(m.x, n[0].x, n[1].x); m.y = op3<DIL>(m.y, n[0].y, n[1].y);
          m.x = op2<DIL>(m.x, n[2].x); m.y = op2<DIL>(m.y, n[2].y);
        }
        if constexpr (JB == 3) {
          m.x = op3<DIL>(m.x, n[3].x, n[4].x); m.y = op3<DIL>(m.y, n[3].y, n[4].y);
          m.x = op3<DIL>(m.x, n[5].x, n[6].x); m.y = op3<DIL>(m.y, n[5].y, n[6].y);
        }
        }
        vh[j] = m;
        L[(pq * NLEV + SB) * WP + hl.pos] = m;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}

template <typename T, int R, bool DIL, int TW, int NP>
__device__ __forceinline__ void ring_upper_halo(typename Vec2<T>::type* const L, const HaloLane hl,
                                                typename Vec2<T>::type (&vh)[HaloCfg<T, R, TW, NP>::NJ]) {
  using C = RingCfg<T, R, TW, NP>;
  using H = HaloCfg<T, R, TW, NP>;
  using T2 = typename Vec2<T>::type;
  constexpr int J = C::J, WP = C::WP, NLEV = C::NLEV, JB = C::JB, SB = C::slot_of(JB);
  if constexpr (J > JB) {
    constexpr int NB = (1 << (J - JB)) - 1;
    const unsigned lds_h = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + hl.pos);
    if (hl.act) {
#pragma unroll
      for (int j = 0; j < H::NJ; ++j) {
        const int q = hl.wave + H::WAVES * j;
        if (q < H::NQ) {
          const int pq = q / H::NH;
          const unsigned ad = lds_h + (unsigned)((pq * NLEV + SB) * WP) * (unsigned)sizeof(T2);
          T2 nb[NB];
          [&]<int... Kk>(std::integer_sequence<int, Kk...>) {
            ((nb[Kk] = lds_read2<((Kk + 1) << JB) * (int)sizeof(T2)>(ad, T())), ...);
          }(std::make_integer_sequence<int, NB>{});
          lds_wait<0>();
          T2 m = vh[j];
          [&]<int... JJ>(std::integer_sequence<int, JJ...>) {
            (([&] {
               constexpr int lv = JB + 1 + JJ;           // level being completed
               constexpr int k0 = 1 << (lv - 1 - JB);    // new cells k0 .. 2*k0 - 1
               if constexpr (k0 == 1) { m.x = op2<DIL>(m.x, nb[0].x); m.y = op2<DIL>(m.y, nb[0].y); }
               else {
#pragma unroll
                 for (int k = k0; k < 2 * k0; k += 2) {
                   m.x = op3<DIL>(m.x, nb[k - 1].x, nb[k].x); m.y = op3<DIL>(m.y, nb[k - 1].y, nb[k].y);
                 }
               }
               if constexpr (C::stored(lv)) {
                 constexpr int sj = C::slot_of(lv);
                 L[(pq * NLEV + sj) * WP + hl.pos] = m;
               }
             }()), ...);
          }(std::make_integer_sequence<int, J - JB>{});
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int R, bool DIL, int TW, int NP>
__device__ __forceinline__ void ring_consume_inplace(typename Vec2<T>::type* const L, const int par, const int tid,
                                                     T (&acc)[2 * R], T (&outv)[2 * NP]);

template <typename T, int R, bool DIL, int TW, int NP>
__device__ __forceinline__ void ring_consume(typename Vec2<T>::type* const L, const int par, const int tid, T (&acc)[2 * R],
                                             T (&outv)[2 * NP]) {
  using C = RingCfg<T, R, TW, NP>;
  using S = typename C::S;
  using T2 = typename Vec2<T>::type;
  if constexpr (C::INPLACE) {
    ring_consume_inplace<T, R, DIL, TW, NP>(L, par, tid, acc, outv);
    return;
  }
  constexpr int K = S::K, WP = C::WP, G = C::G, NG = C::NG, NLEV = C::NLEV, D = C::D;
  constexpr int KR1 = S::kidx(R - 1);                   // width index of dy = +-(R-1)
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + tid + R);
  // (4) consume: window lookups + ring update, pair by pair
  {
  SMRF_MARK("consume");
  T2 own[NP];                                            // the lane's own cells (level 0)
#pragma unroll
  for (int p = 0; p < NP; ++p)
    own[p] = lds_read2<0>(lds_q + (p * NLEV + par) * WP * (unsigned)sizeof(T2), T());
  lds_wait<0>();
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned q = lds_q + p * NLEV * WP * (unsigned)sizeof(T2);   // this lane's cell, level 0
    // the lookup half of a pair (LDS reads + the first-half ring slots they release) above the VALU-only second
    // half: of the SIMD's two or three waves, the one that can put reads in flight goes first and the other fills the
    // gaps with its min3 stretch.  Measured -4...-8 % at every radius from 20 up (gpurun_out/r02/probe_prio2.log);
    // levels 1 and 2 are equal, 3 (= the build phases' level) gives the gain back.
    __builtin_amdgcn_s_setprio(SMRF_RING_LOOKUP_PRIO);
    T ra[K], rb[K];                                      // window results of row A / row B per width
    T2 ta[D][G], tb[D][G], tc[D][G], td[D][G];          // D lookup groups in flight (tc: third read of the widest widths; INC: tc, td second read per side)
    T2 te[D][G], tf[D][G];                               // INC: third read per side (a first step of 9..11 cells at level 2)
    // INC reads level 0 as well: this batch's copy of it
    // (addressed from the row's first staged cell: ds_read offsets are unsigned)
    const unsigned q0 = q + (unsigned)par * (unsigned)(WP * sizeof(T2)) - (unsigned)(R * sizeof(T2));
    auto issue = [&]<int GI>(std::integral_constant<int, GI>) {
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = 1 + GI * G + I;
           if constexpr (k < K && C::INC) {
             constexpr int w = S::wk(k);
             constexpr int j = C::inc_lev(k);
             static_assert(C::stored(j), "step level not built");
             static_assert(C::inc_n(k) <= 3, "width step longer than three table entries");
             constexpr int base = j == 0 ? R : C::slot_of(j) * WP;
             const unsigned qq = j == 0 ? q0 : q;
             ta[GI % D][I] = lds_read2<(base - w) * (int)sizeof(T2)>(qq, T());
             tb[GI % D][I] = lds_read2<(base + w - (1 << j) + 1) * (int)sizeof(T2)>(qq, T());
             if constexpr (C::inc_n(k) >= 2) {
               tc[GI % D][I] = lds_read2<(base - w + (1 << j)) * (int)sizeof(T2)>(qq, T());
               td[GI % D][I] = lds_read2<(base + w - (2 << j) + 1) * (int)sizeof(T2)>(qq, T());
             }
             if constexpr (C::inc_n(k) == 3) {
               te[GI % D][I] = lds_read2<(base - w + (2 << j)) * (int)sizeof(T2)>(qq, T());
               tf[GI % D][I] = lds_read2<(base + w - (3 << j) + 1) * (int)sizeof(T2)>(qq, T());
             }
           } else if constexpr (k < K) {
             constexpr int w = S::wk(k);
             constexpr int j = C::lev(w);
             constexpr int base = C::slot_of(j) * WP;
             static_assert(C::stored(j), "lookup level not built");
             ta[GI % D][I] = lds_read2<(base - w) * (int)sizeof(T2)>(q, T());
             tb[GI % D][I] = lds_read2<(base + w - (1 << j) + 1) * (int)sizeof(T2)>(q, T());
             if constexpr (C::nreads(w) == 3)              // window longer than two entries of the top level: one in between
               tc[GI % D][I] = lds_read2<(base - w + (1 << j)) * (int)sizeof(T2)>(q, T());
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    auto reduce = [&]<int GI>(std::integral_constant<int, GI>) {
      if constexpr (C::fused_red(GI)) {
        constexpr int k0 = 1 + GI * G, n = C::gsize(GI), b = GI % D;
        constexpr int i2 = G > 2 ? 2 : 0, i3 = G > 3 ? 3 : 0;
        float x0, y0, x1, y1;
        if constexpr (n == 2) {
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y);
        } else if constexpr (n == 3) {
          float x2, y2;
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y, x2, y2, ta[b][i2].x, ta[b][i2].y, tb[b][i2].x, tb[b][i2].y);
          ra[k0 + 2] = x2; rb[k0 + 2] = y2;
        } else {
          float x2, y2, x3, y3;
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y, x2, y2, ta[b][i2].x, ta[b][i2].y, tb[b][i2].x, tb[b][i2].y, x3, y3, ta[b][i3].x,
                         ta[b][i3].y, tb[b][i3].x, tb[b][i3].y);
          ra[k0 + 2] = x2; rb[k0 + 2] = y2; ra[k0 + 3] = x3; rb[k0 + 3] = y3;
        }
        ra[k0] = x0; rb[k0] = y0; ra[k0 + 1] = x1; rb[k0 + 1] = y1;
      } else
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = 1 + GI * G + I;
           if constexpr (k < K && C::INC) {
             T a = op3<DIL>(ra[k - 1], ta[GI % D][I].x, tb[GI % D][I].x);
             T b = op3<DIL>(rb[k - 1], ta[GI % D][I].y, tb[GI % D][I].y);
             if constexpr (C::inc_n(k) >= 2) {
               a = op3<DIL>(a, tc[GI % D][I].x, td[GI % D][I].x);
               b = op3<DIL>(b, tc[GI % D][I].y, td[GI % D][I].y);
             }
             if constexpr (C::inc_n(k) == 3) {
               a = op3<DIL>(a, te[GI % D][I].x, tf[GI % D][I].x);
               b = op3<DIL>(b, te[GI % D][I].y, tf[GI % D][I].y);
             }
             ra[k] = a;
             rb[k] = b;
           } else if constexpr (k < K) {
             if constexpr (C::nreads(S::wk(k)) == 3) {
               ra[k] = op3<DIL>(ta[GI % D][I].x, tc[GI % D][I].x, tb[GI % D][I].x);
               rb[k] = op3<DIL>(ta[GI % D][I].y, tc[GI % D][I].y, tb[GI % D][I].y);
             } else {
               ra[k] = op2<DIL>(ta[GI % D][I].x, tb[GI % D][I].x);
               rb[k] = op2<DIL>(ta[GI % D][I].y, tb[GI % D][I].y);
             }
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    auto slots = [&]<int GI>(std::integral_constant<int, GI>) {   // ring slots released by group GI
      [&]<int... Sl>(std::integer_sequence<int, Sl...>) {
        (([&] {
           if constexpr (C::slot_group(Sl) == GI) {
             constexpr int ka = C::kA(Sl), kb = C::kB(Sl);   // forced compile-time: static registers
             acc[Sl] = op3<DIL>(acc[Sl + 2], ra[ka], rb[kb]);
           }
         }()), ...);
      }(std::make_integer_sequence<int, (2 * R - 2 > 0 ? 2 * R - 2 : 0)>{});
    };

    [&]<int... GI>(std::integer_sequence<int, GI...>) {   // prologue: the first D-1 groups
      (([&] { if constexpr (GI < NG) issue(std::integral_constant<int, GI>{}); }()), ...);
    }(std::make_integer_sequence<int, D - 1>{});
    ra[0] = own[p].x;
    rb[0] = own[p].y;
    // the two rows this pair completes (before their slots are overwritten)
    outv[2 * p] = op2<DIL>(acc[0], ra[0]);
    constexpr bool SD = C::SLOT_DELAY && NG >= 2;          // slots one group late (RingCfg::SLOT_DELAY)
    [&]<int... GI>(std::integer_sequence<int, GI...>) {
      (([&] {
         if constexpr (GI + D - 1 < NG) issue(std::integral_constant<int, GI + D - 1>{});
         lds_wait<C::inflight_after(GI), (C::FUSE_MODE != 0)>();   // reads of the groups issued after group GI
         reduce(std::integral_constant<int, GI>{});
         constexpr int GS = SD ? GI - 1 : GI;             // the group whose slots are updated here
         if constexpr (GS == 0) {
           if constexpr (R >= 2) outv[2 * p + 1] = op3<DIL>(acc[1], ra[KR1], rb[0]);
         }
         if constexpr (GS >= 0) slots(std::integral_constant<int, GS>{});
       }()), ...);
    }(std::make_integer_sequence<int, NG>{});
    if constexpr (R == 1) outv[2 * p + 1] = op3<DIL>(acc[1], ra[KR1], rb[0]);
    __builtin_amdgcn_s_setprio(0);
    if constexpr (SD) slots(std::integral_constant<int, NG - 1>{});
    slots(std::integral_constant<int, NG>{});           // second half, all widths are in registers
    acc[2 * R - 2] = op2<DIL>(ra[0], rb[KR1]);
    acc[2 * R - 1] = rb[0];
    __builtin_amdgcn_sched_barrier(0);
  }
  }
}

// ring_consume with the second half of the ring updated in place (RingCfg::INPLACE).
//   acc[0 .. R-2]      first half F, shifting as in ring_consume: F[s] = op3(F[s+2], RA[kA(s)], RB[kB(s)]), s ascending,
//                      i.e. widths ascending
//   acc[R-1 .. 2R-1]   second half B as a register ring of M = R + 1: logical slot j = s - (R - 1) of pair P (P-th pair of
//                      the batch) lives in acc[R - 1 + (j + 2P) % M].  The pair's update B'[j] = op3(B[j+2], RA[kidx(j+1)],
//                      RB[kidx(j)]) overwrites the register of B[j+2] - which IS the register of B'[j] under the next
//                      pair's mapping - so the slots can be taken in any order: j descending = widths ascending, each as
//                      soon as its lookup group is reduced.  B[0], B[1] are only read (by F[R-3], F[R-2], last) and then
//                      take the pair's two new-born slots.  After the batch's NP pairs the mapping is turned back by
//                      2 NP registers (M v_mov per batch, 2.1 cycles each against 4.1 for the min/max they sit beside).
template <typename T, int R, bool DIL, int TW, int NP>
__device__ __forceinline__ void ring_consume_inplace(typename Vec2<T>::type* const L, const int par, const int tid,
                                                     T (&acc)[2 * R], T (&outv)[2 * NP]) {
  using C = RingCfg<T, R, TW, NP>;
  using S = typename C::S;
  using T2 = typename Vec2<T>::type;
  static_assert(C::INC && R >= 4, "in-place ring: incremental widths, R >= 4");
  constexpr int K = S::K, WP = C::WP, G = C::G, NG = C::NG, NLEV = C::NLEV, D = C::D;
  constexpr int KR1 = S::kidx(R - 1);
  constexpr int M = R + 1;
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + tid + R);
  SMRF_MARK("consume");
  T2 own[NP];
#pragma unroll
  for (int p = 0; p < NP; ++p)
    own[p] = lds_read2<0>(lds_q + (p * NLEV + par) * WP * (unsigned)sizeof(T2), T());
  lds_wait<0>();
  auto pair_body = [&]<int P>(std::integral_constant<int, P>) {
    const unsigned q = lds_q + P * NLEV * WP * (unsigned)sizeof(T2);
    __builtin_amdgcn_s_setprio(SMRF_RING_LOOKUP_PRIO);
    T ra[K], rb[K];
    T2 ta[D][G], tb[D][G], tc[D][G], td[D][G], te[D][G], tf[D][G];
    const unsigned q0 = q + (unsigned)par * (unsigned)(WP * sizeof(T2)) - (unsigned)(R * sizeof(T2));
    auto issue = [&]<int GI>(std::integral_constant<int, GI>) {
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = 1 + GI * G + I;
           if constexpr (k < K) {
             constexpr int w = S::wk(k);
             constexpr int j = C::inc_lev(k);
             static_assert(C::stored(j), "step level not built");
             static_assert(C::inc_n(k) <= 3, "width step longer than three table entries");
             constexpr int base = j == 0 ? R : C::slot_of(j) * WP;
             const unsigned qq = j == 0 ? q0 : q;
             ta[GI % D][I] = lds_read2<(base - w) * (int)sizeof(T2)>(qq, T());
             tb[GI % D][I] = lds_read2<(base + w - (1 << j) + 1) * (int)sizeof(T2)>(qq, T());
             if constexpr (C::inc_n(k) >= 2) {
               tc[GI % D][I] = lds_read2<(base - w + (1 << j)) * (int)sizeof(T2)>(qq, T());
               td[GI % D][I] = lds_read2<(base + w - (2 << j) + 1) * (int)sizeof(T2)>(qq, T());
             }
             if constexpr (C::inc_n(k) == 3) {
               te[GI % D][I] = lds_read2<(base - w + (2 << j)) * (int)sizeof(T2)>(qq, T());
               tf[GI % D][I] = lds_read2<(base + w - (3 << j) + 1) * (int)sizeof(T2)>(qq, T());
             }
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    auto reduce = [&]<int GI>(std::integral_constant<int, GI>) {
      if constexpr (C::fused_red(GI)) {
        constexpr int k0 = 1 + GI * G, n = C::gsize(GI), b = GI % D;
        constexpr int i2 = G > 2 ? 2 : 0, i3 = G > 3 ? 3 : 0;
        float x0, y0, x1, y1;
        if constexpr (n == 2) {
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y);
        } else if constexpr (n == 3) {
          float x2, y2;
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y, x2, y2, ta[b][i2].x, ta[b][i2].y, tb[b][i2].x, tb[b][i2].y);
          ra[k0 + 2] = x2; rb[k0 + 2] = y2;
        } else {
          float x2, y2, x3, y3;
          red_chain<DIL>(ra[k0 - 1], rb[k0 - 1], x0, y0, ta[b][0].x, ta[b][0].y, tb[b][0].x, tb[b][0].y, x1, y1, ta[b][1].x, ta[b][1].y,
                         tb[b][1].x, tb[b][1].y, x2, y2, ta[b][i2].x, ta[b][i2].y, tb[b][i2].x, tb[b][i2].y, x3, y3, ta[b][i3].x,
                         ta[b][i3].y, tb[b][i3].x, tb[b][i3].y);
          ra[k0 + 2] = x2; rb[k0 + 2] = y2; ra[k0 + 3] = x3; rb[k0 + 3] = y3;
        }
        ra[k0] = x0; rb[k0] = y0; ra[k0 + 1] = x1; rb[k0 + 1] = y1;
      } else
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = 1 + GI * G + I;
           if constexpr (k < K) {
             T a = op3<DIL>(ra[k - 1], ta[GI % D][I].x, tb[GI % D][I].x);
             T b = op3<DIL>(rb[k - 1], ta[GI % D][I].y, tb[GI % D][I].y);
             if constexpr (C::inc_n(k) >= 2) {
               a = op3<DIL>(a, tc[GI % D][I].x, td[GI % D][I].x);
               b = op3<DIL>(b, tc[GI % D][I].y, td[GI % D][I].y);
             }
             if constexpr (C::inc_n(k) == 3) {
               a = op3<DIL>(a, te[GI % D][I].x, tf[GI % D][I].x);
               b = op3<DIL>(b, te[GI % D][I].y, tf[GI % D][I].y);
             }
             ra[k] = a;
             rb[k] = b;
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    // (k - 1) / G: the lookup group that completes width index k >= 1
    auto slots = [&]<int GI>(std::integral_constant<int, GI>) {
      // first half, s = 0 .. R-2, ascending (F[R-3], F[R-2] take their shifted-in value from B[0], B[1])
      [&]<int... Sl>(std::integer_sequence<int, Sl...>) {
        (([&] {
           constexpr int ka = C::kA(Sl), kb = C::kB(Sl);       // ka >= kb >= 0, ka >= 1
           if constexpr ((ka - 1) / G == GI) {
             constexpr int src = Sl + 2 <= R - 2 ? Sl + 2 : R - 1 + ((Sl + 2 - (R - 1)) + 2 * P) % M;
             acc[Sl] = op3<DIL>(acc[src], ra[ka], rb[kb]);
           }
         }()), ...);
      }(std::make_integer_sequence<int, R - 1>{});
      // second half in place, logical j = R-2 .. 0 (widths ascending): B'[j] = op3(B[j+2], RA[kidx(j+1)], RB[kidx(j)])
      [&]<int... Jr>(std::integer_sequence<int, Jr...>) {
        (([&] {
           constexpr int j = R - 2 - Jr;
           constexpr int ka = S::kidx(j + 1), kb = S::kidx(j);   // kb >= ka
           if constexpr ((kb - 1) / G == GI) {
             constexpr int reg = R - 1 + (j + 2 + 2 * P) % M;
             op3_acc<DIL>(acc[reg], ra[ka], rb[kb]);
           }
         }()), ...);
      }(std::make_integer_sequence<int, R - 1>{});
    };
    [&]<int... GI>(std::integer_sequence<int, GI...>) {   // prologue: the first D-1 groups
      (([&] { if constexpr (GI < NG) issue(std::integral_constant<int, GI>{}); }()), ...);
    }(std::make_integer_sequence<int, D - 1>{});
    ra[0] = own[P].x;
    rb[0] = own[P].y;
    outv[2 * P] = op2<DIL>(acc[0], ra[0]);
    constexpr bool SD = C::SLOT_DELAY && NG >= 2;          // slots one group late (RingCfg::SLOT_DELAY)
    [&]<int... GI>(std::integer_sequence<int, GI...>) {
      (([&] {
         if constexpr (GI + D - 1 < NG) issue(std::integral_constant<int, GI + D - 1>{});
         lds_wait<C::inflight_after(GI), (C::FUSE_MODE != 0)>();
         reduce(std::integral_constant<int, GI>{});
         constexpr int GS = SD ? GI - 1 : GI;             // the group whose slots are updated here
         if constexpr (GS == (KR1 - 1) / G) outv[2 * P + 1] = op3<DIL>(acc[1], ra[KR1], rb[0]);   // before F[1] is overwritten
         if constexpr (!SD && GI == NG - 1) __builtin_amdgcn_s_setprio(0);
         if constexpr (GS >= 0) slots(std::integral_constant<int, GS>{});
       }()), ...);
    }(std::make_integer_sequence<int, NG>{});
    if constexpr (SD) {
      __builtin_amdgcn_s_setprio(0);
      if constexpr (NG - 1 == (KR1 - 1) / G) outv[2 * P + 1] = op3<DIL>(acc[1], ra[KR1], rb[0]);
      slots(std::integral_constant<int, NG - 1>{});
    }
    // the pair's two new-born slots (logical R-1, R of the next pair's mapping) take the registers of B[0], B[1]
    acc[R - 1 + (2 * P) % M] = op2<DIL>(ra[0], rb[KR1]);
    acc[R - 1 + (1 + 2 * P) % M] = rb[0];
    __builtin_amdgcn_sched_barrier(0);
  };
  [&]<int... P>(std::integer_sequence<int, P...>) {
    (pair_body(std::integral_constant<int, P>{}), ...);
  }(std::make_integer_sequence<int, NP>{});
  // turn the mapping back: logical j sits in B[(j + 2 NP) % M]
  {
    constexpr int ROT = (2 * NP) % M;
    if constexpr (ROT != 0) {
      SMRF_MARK("turn");
      T t[M];
#pragma unroll
      for (int i = 0; i < M; ++i) t[i] = acc[R - 1 + (i + ROT) % M];
      // as asm moves: left to the compiler the copies are sunk to the next batch's tied updates and multiply there
      // (R = 46: 102 v_mov per batch, 12 B of scratch; as asm 70 and none - M = 47 of them are the turn itself)
      if constexpr (sizeof(T) == 4 && SMRF_RING_TURN_ASM(T, R)) {
#pragma unroll
        for (int i = 0; i < M; ++i) asm volatile("v_mov_b32 %0, %1" : "=v"(acc[R - 1 + i]) : "v"(t[i]));
      } else {
#pragma unroll
        for (int i = 0; i < M; ++i) acc[R - 1 + i] = t[i];
      }
    }
  }
}

// the three phases of one stage back to back (the single-pass kernels)
template <typename T, int R, bool DIL, int TW, int NP, int NPB, int OFF, typename Sync>
__device__ __forceinline__ void ring_build_consume(typename Vec2<T>::type* const L, const int par, const int tid,
                                                   const bool has_last,
                                                   typename Vec2<T>::type (&v)[NP][RingCfg<T, R, TW, NP>::NPOS],
                                                   T (&acc)[2 * R], T (&outv)[2 * NP], Sync&& phase_sync) {
  // the build phases end in barriers the whole workgroup waits at: let a wave in them win the issue arbitration
  // against the SIMD's other wave (which is usually consuming); measured -3...-5.5 % at every radius >= 8
  __builtin_amdgcn_s_setprio(SMRF_RING_BUILD_PRIO);
#ifndef SMRF_RING_DBG_NOBUILD   // (timing experiment only, wrong results: what the table build costs)
  ring_base<T, R, DIL, TW, NP, NPB, OFF>(L, par, tid, has_last, v);
  phase_sync();
  if constexpr (RingCfg<T, R, TW, NP>::J > RingCfg<T, R, TW, NP>::JB) {
    ring_upper<T, R, DIL, TW, NP, NPB, OFF>(L, tid, has_last, v);
    phase_sync();
  }
#endif
  __builtin_amdgcn_s_setprio(0);
  ring_consume<T, R, DIL, TW, NP>(L, par, tid, acc, outv);
}

template <typename T, int R, bool DIL, int TW, int NP>
__global__ __launch_bounds__(TW, (SMRF_OCC_OVERRIDE(TW > 256 ? (RingCfg<T, R, TW, NP>::OCC * 256 / TW < 1 ? 1 : RingCfg<T, R, TW, NP>::OCC * 256 / TW) : RingCfg<T, R, TW, NP>::OCC)))
void ring_kernel(const DiskArgs<T> a) {
  using C = RingCfg<T, R, TW, NP>;
  using T2 = typename Vec2<T>::type;
  constexpr int WP = C::WP, ROWS = C::ROWS, NLEV = C::NLEV;
  // fp64 with three row pairs per batch is NOT a validated configuration: a variant build that forced it gave wrong cells at
  // R = 57, 58, 62, 63, 64 (512 registers + 240-290 B of scratch per lane; tools/ab_equal.py, profiles/r05_f64_np.md) while
  // every shipped instance (<= 2 pairs) equals the oracle on every radius.  ring_tune.inc's fp64 caps keep it out.
  static_assert(sizeof(T) == 4 || NP <= 2, "fp64 ring instances are validated for at most two row pairs per batch");
  extern __shared__ __attribute__((aligned(16))) unsigned char smrf_lds[];
  T2* const L = reinterpret_cast<T2*>(smrf_lds);         // [NP][NLEV][WP] of {row A, row B}

  const int tid = threadIdx.x;
  // Workgroups are dealt round-robin over the 8 XCDs in dispatch order (x fastest), each XCD with
  // its own L2.  Remap the tile so that an XCD owns a contiguous range of strips (all their
  // segments): neighbouring strips share 2R halo columns, which then hit the same L2.  Placement
  // only, any mapping is correct (MI355X_MICROARCH: workgroup dispatch, XCD placement).
  int bx = blockIdx.x, by = blockIdx.y;
#if SMRF_RING_XCD_REMAP == 2
  {   // tuning build: an XCD owns a contiguous range of tiles in segment-major order (all strips of the same rows together)
    const int total = gridDim.x * gridDim.y;
    if ((total & 7) == 0) {
      const int id = blockIdx.y * gridDim.x + blockIdx.x;
      const int t = (id & 7) * (total >> 3) + (id >> 3);
      bx = t % gridDim.x;
      by = t / gridDim.x;
    }
  }
#elif SMRF_RING_XCD_REMAP
  if (a.plain_tiles) {
  } else if ((gridDim.x & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.x >> 3;
    const int xcd = id & 7, slot = id >> 3;
    bx = xcd * per + slot % per;
    by = slot / per;
  } else if (gridDim.x > 8 && a.seg_cls == 0) {
    // any other strip count (round 5): an XCD owns a contiguous range of the tiles taken strip by strip (a strip's segments
    // together), i.e. strips / 8 neighbouring strips and parts of the two at its ends.  Without it the halo columns of a
    // raster of arbitrary width came from HBM again: 8193 columns fetched 5.6-7.3 B per cell and erosion pass against
    // 4.7-5.3 at 8192 (profiles/r05_segment_balance.md section 5).  XCD x gets ceil((total - x) / 8) of the workgroups.
    const int total = gridDim.x * gridDim.y, id = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = id & 7, slot = id >> 3, q = total >> 3, rem = total & 7;
    const int t = xcd * q + (xcd < rem ? xcd : rem) + slot;
    bx = t / (int)gridDim.y;
    by = t % (int)gridDim.y;
  }
#endif
#ifdef SMRF_RING_DBG_CLOCK   // timing experiment only: the shader clock this workgroup ran at, left in the output's first two cells
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memtime(), dbg_q0 = __builtin_amdgcn_s_memrealtime();
#endif
#ifdef SMRF_RING_DBG_TS
  const unsigned long long dbg_ts0 = __builtin_amdgcn_s_memrealtime();
#endif
  const int x0 = bx * TW;
  const int x = x0 + tid;
  int ys, ye;                                            // global output rows [ys, ye)
  if (a.seg_cls > 0) {                                   // segments of unequal length (ring_launch_np: residency classes)
    int cls = 0;
    for (int c = 1; c < a.seg_cls; ++c) cls += by >= a.seg_first[c] ? 1 : 0;
    ys = a.out_row0 + a.seg_row0[cls] + (by - a.seg_first[cls]) * a.seg_len[cls];
    ye = min(a.out_row0 + a.out_rows, ys + a.seg_len[cls]);
    if (ys >= ye) return;                                // (lengths are rounded up: a trailing segment may be empty)
  } else {
    ys = a.out_row0 + by * a.seg;
    ye = min(a.out_row0 + a.out_rows, ys + a.seg);
  }
  constexpr int NPOS = C::NPOS, W = C::W;
  // lane-owned staged cells: positions tid + i*TW of the TW+2R wide row; only the last can be absent
  const bool has_last = tid + (NPOS - 1) * TW < W;
  int cpos[NPOS];
#pragma unroll
  for (int i = 0; i < NPOS; ++i) cpos[i] = smrf_fold(x0 - R + tid + (tid + i * TW < W ? i * TW : 0), a.cols);
  const int last_in = a.in_rows - 1;
  // halo cells shared out over the waves (HaloCfg): this lane's halo cell and the column it is loaded from
  using H = HaloCfg<T, R, TW, NP>;
  constexpr bool BAL = H::OK;
  HaloLane hl;
  hl.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const int h = (hl.wave % H::NH) * 64 + (tid & 63);
    hl.act = BAL && h < H::HW;
    hl.pos = hl.act ? TW + h : TW;
  }
  const int hcol = smrf_fold(x0 - R + hl.pos, a.cols);
  // workgroup-wide when the table is shared by several waves; a single-wave workgroup owns its
  // table and only has to keep the compiler from moving LDS accesses across the phase boundary
  auto phase_sync = [&]() {
#ifdef SMRF_RING_DBG_NOSYNC   // timing experiment only (wrong results): what the workgroup barriers cost
    asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); return;
#endif
    if constexpr (TW > 64) __syncthreads();
    else { asm volatile("" ::: "memory"); __builtin_amdgcn_wave_barrier(); }
  };
  const bool flag = a.mask != nullptr;
  const int xc = x < a.cols ? x : a.cols - 1;

  // ring: between pairs, slot s holds the partial result of output row (next input row) - R + s
  T acc[2 * R];
#pragma unroll
  for (int i = 0; i < 2 * R; ++i) acc[i] = ident<T>(DIL);

  T2 pf[NP][NPOS];                                       // next batch, this lane's staged cells
  T2 pfh[H::NJ];                                         // BAL: next batch, this lane's halo cell of its wave's jobs
  T outv[ROWS], lastv[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) { outv[i] = T(0); lastv[i] = T(0); }

  // The row loop starts DELTA rows early so that the ROWS outputs a batch completes are either all
  // at or above ys or all below it (rows before ys - R only feed outputs that are never written).
  constexpr int DELTA = (ROWS - (2 * R) % ROWS) % ROWS;
  const int ystart = ys - R - DELTA;
  RowFold rf(ystart, a.img_rows);                        // tracks the NEXT batch to prefetch
  // buffer addressing of the common cases (SMRF_RING_BUF): descriptors based at the first row of each plane this
  // workgroup touches there, byte offsets of the lane's columns; the host keeps a segment's span below 2 GiB (ring_launch_np)
  constexpr bool BUF = ring_buf_on<T, R, TW, NP>();
  constexpr bool RARE = C::INPLACE && SMRF_RING_RARE_OPAQUE(T, R);
  const int bi = max(0, ystart - a.in_row0);                       // first band row of `in` a fast-path batch can start at
  const int bl = max(0, ys - 2 * R - DELTA - a.out_row0);          // ... of `last` (first batch: outputs of rows ystart - R ...)
  const int bo = ys - a.out_row0;                                  // ... of `out`, `mask`, `when`
  const unsigned rowb = (unsigned)a.ld * (unsigned)sizeof(T);     // bytes per row
  // (descriptors of planes a launch does not have are built from null pointers and never used)
  const smrf_rsrc_t rs_in = smrf_make_rsrc(a.in + (long long)bi * a.ld);
  const smrf_rsrc_t rs_out = smrf_make_rsrc(a.out + (long long)bo * a.ld);
  const smrf_rsrc_t rs_last = smrf_make_rsrc(flag ? a.last + (long long)bl * a.ld : nullptr);
  const smrf_rsrc_t rs_mask = smrf_make_rsrc(flag ? a.mask + (long long)bo * a.ld : nullptr);
  const smrf_rsrc_t rs_when = smrf_make_rsrc(flag && a.when != nullptr ? a.when + (long long)bo * a.ld : nullptr);
  unsigned cposb[NPOS], hcolb = 0, xb = 0, xcb = 0;
  if constexpr (BUF) {
#pragma unroll
    for (int i = 0; i < NPOS; ++i) cposb[i] = (unsigned)cpos[i] * (unsigned)sizeof(T);
    hcolb = (unsigned)hcol * (unsigned)sizeof(T);
    xb = (unsigned)x * (unsigned)sizeof(T);
    xcb = (unsigned)xc * (unsigned)sizeof(T);
  }
  auto prefetch = [&]() {
#ifdef SMRF_RING_DBG_NOLOAD   // timing experiment only (wrong results): the kernel without its loads from HBM
    {
      for (int p = 0; p < NP; ++p)
        for (int i = 0; i < NPOS; ++i) { pf[p][i].x = (T)(tid + p); pf[p][i].y = (T)(tid - i); }
      for (int j = 0; j < H::NJ; ++j) { pfh[j].x = (T)tid; pfh[j].y = (T)j; }
      rf.advance(ROWS);
      return;
    }
#endif
    const int l0 = rf.p - a.in_row0;
    if (rf.p + ROWS <= rf.n && l0 >= 0 && l0 + ROWS - 1 <= last_in) {
      // common case: ROWS consecutive rows inside the band, no reflection: one address, row strides
      SMRF_MARK("prefetch");
      if constexpr (BUF) {
        const unsigned s0 = (unsigned)(l0 - bi) * rowb;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
#pragma unroll
          for (int i = 0; i < (BAL ? 1 : NPOS); ++i) {
            pf[p][i].x = smrf_buf_load(rs_in, cposb[i], s0 + (unsigned)(2 * p) * rowb, T());
            pf[p][i].y = smrf_buf_load(rs_in, cposb[i], s0 + (unsigned)(2 * p + 1) * rowb, T());
          }
        }
        if constexpr (BAL) {
#pragma unroll
          for (int j = 0; j < H::NJ; ++j) {
            const int q = hl.wave + H::WAVES * j;
            const int pq = q < H::NQ ? q / H::NH : 0;
            pfh[j].x = smrf_buf_load(rs_in, hcolb, s0 + (unsigned)(2 * pq) * rowb, T());
            pfh[j].y = smrf_buf_load(rs_in, hcolb, s0 + (unsigned)(2 * pq + 1) * rowb, T());
          }
        }
      } else {
      const T* r0 = a.in + (long long)l0 * a.ld;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < (BAL ? 1 : NPOS); ++i) {
          pf[p][i].x = r0[(long long)(2 * p) * a.ld + cpos[i]];
          pf[p][i].y = r0[(long long)(2 * p + 1) * a.ld + cpos[i]];
        }
      }
      if constexpr (BAL) {
#pragma unroll
        for (int j = 0; j < H::NJ; ++j) {
          const int q = hl.wave + H::WAVES * j;            // wave-job: row pair q / NH (inactive lanes load a valid cell)
          const int pq = q < H::NQ ? q / H::NH : 0;
          pfh[j].x = r0[(long long)(2 * pq) * a.ld + hcol];
          pfh[j].y = r0[(long long)(2 * pq + 1) * a.ld + hcol];
        }
      }
      }
    } else {
      SMRF_MARK("rare");
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        int la = rf.at(2 * p) - a.in_row0;
        int lb = rf.at(2 * p + 1) - a.in_row0;
        la = la < 0 ? 0 : (la > last_in ? last_in : la);   // only rows outside the segment's halo clamp
        lb = lb < 0 ? 0 : (lb > last_in ? last_in : lb);
        const T* ra = a.in + (long long)la * a.ld;
        const T* rb = a.in + (long long)lb * a.ld;
#pragma unroll
        for (int i = 0; i < (BAL ? 1 : NPOS); ++i) { const int c = smrf_rare<RARE>(cpos[i]); pf[p][i].x = ra[c]; pf[p][i].y = rb[c]; }
      }
      if constexpr (BAL) {
#pragma unroll
        for (int j = 0; j < H::NJ; ++j) {
          const int q = hl.wave + H::WAVES * j;
          const int pq = q < H::NQ ? q / H::NH : 0;
          int la = rf.at(2 * pq) - a.in_row0;
          int lb = rf.at(2 * pq + 1) - a.in_row0;
          la = la < 0 ? 0 : (la > last_in ? last_in : la);
          lb = lb < 0 ? 0 : (lb > last_in ? last_in : lb);
          const int hc = smrf_rare<RARE>(hcol);
          pfh[j].x = a.in[(long long)la * a.ld + hc];
          pfh[j].y = a.in[(long long)lb * a.ld + hc];
        }
      }
    }
    SMRF_MARK("prefetch");
    rf.advance(ROWS);
  };
  // one completed output cell, general form: NaN rule, store, flag step (sparse or dense)
  auto emit = [&](int yo, long long off, T val, T lastval) {
    if (a.nan_aware) {
      // scipy: the first visited footprint element (offset (-R, 0)) decides NaN-ness
      const int ly = smrf_fold(yo - R, a.img_rows) - a.in_row0;
      const T first = a.in[(long long)ly * a.ld + smrf_rare<RARE>(x)];
      if (first != first) val = qnan<T>();
    }
    smrf_store_out(&a.out[off], val, a.nt);
    if (flag) smrf_flag_cell(a, off, lastval, val);
  };
  // the common case of a batch (all ROWS rows inside the segment, no NaN rule, sparse flags) as straight-line code per
  // (store kind, flag step): the uniform tests are made once per batch, not once per cell - scalar branches between
  // the VALU instructions of a kernel that runs at 2-3 waves per SIMD are not free (profiles/r02_issue_rate_ubench.md)
  auto emit_rows = [&]<bool NT, bool FLAG>(std::bool_constant<NT>, std::bool_constant<FLAG>, long long off0, int ro0) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const long long off = off0 + (long long)i * a.ld;
      const unsigned so = (unsigned)(ro0 - bo + i) * rowb;   // BUF: row offset from the segment's first output row
      if constexpr (BUF) smrf_buf_store<NT>(rs_out, xb, so, outv[i]);
      else if constexpr (NT) __builtin_nontemporal_store(outv[i], &a.out[off]);
      else a.out[off] = outv[i];
      if constexpr (FLAG) {
        const T diff = lastv[i] - outv[i];                   // raster dtype
        bool hit;                                            // float64 comparison (NumPy 2); fp32: smrf_float_below
        if constexpr (sizeof(T) == 4) hit = diff > a.thr_lo;
        else hit = (double)diff > a.thr;
        if (hit) {
          if constexpr (BUF) {
            const unsigned sm = (unsigned)(ro0 - bo + i) * (unsigned)a.ld;
            smrf_buf_store_u8(rs_mask, (unsigned)x, sm, (uint8_t)1);
            if (a.when != nullptr) smrf_buf_store_u8(rs_when, (unsigned)x, sm, (uint8_t)a.widx);
          } else {
            a.mask[off] = 1;
            if (a.when != nullptr) a.when[off] = (uint8_t)a.widx;
          }
        }
      }
    }
  };
  // outputs of the batch whose first input row was yyb; written one iteration late so that no
  // store is younger than the prefetch loads the loop waits for
  auto epilogue = [&](int yyb) {
    const int yob = yyb - R;                               // first output row of the batch
    if (yob < ys || x >= a.cols) return;                   // (aligned: yob < ys means all rows are)
#ifdef SMRF_RING_DBG_NOSTORE   // timing experiment only (wrong results): the kernel without its stores (kept alive by a test no value passes)
    {
      bool any = false;
      for (int i = 0; i < ROWS; ++i) any |= (outv[i] == (T)-12345.678f) | (lastv[i] == (T)-12345.678f);
      if (!any) return;
    }
#endif
    const long long off0 = (long long)(yob - a.out_row0) * a.ld + x;
    if (yob + ROWS <= ye && !a.nan_aware && !a.dense) {
      const int ro0 = yob - a.out_row0;
      if (flag) {
        if (a.nt) { SMRF_MARK("epilogue:nt1:flag1"); emit_rows(std::true_type{}, std::true_type{}, off0, ro0); }
        else { SMRF_MARK("epilogue:nt0:flag1"); emit_rows(std::false_type{}, std::true_type{}, off0, ro0); }
      } else {
        if (a.nt) { SMRF_MARK("epilogue:nt1:flag0"); emit_rows(std::true_type{}, std::false_type{}, off0, ro0); }
        else { SMRF_MARK("epilogue:nt0:flag0"); emit_rows(std::false_type{}, std::false_type{}, off0, ro0); }
      }
      SMRF_MARK("epilogue");
    } else {
      SMRF_MARK("rare");
      const long long off0r = (long long)(yob - a.out_row0) * a.ld + smrf_rare<RARE>(x);
#pragma unroll
      for (int i = 0; i < ROWS; ++i)
        if (yob + i < ye) emit(yob + i, off0r + (long long)i * a.ld, outv[i], lastv[i]);
      SMRF_MARK("epilogue");
    }
  };
  auto load_last = [&](int yyb) {
    if (!flag) return;
#ifdef SMRF_RING_DBG_NOLOAD
    { for (int i = 0; i < ROWS; ++i) lastv[i] = (T)-1e30f; return; }   // (no cell is flagged)
#endif
    const int y0 = yyb - R - a.out_row0;
    if (y0 >= 0 && y0 + ROWS <= a.out_rows) {
      SMRF_MARK("last");
      if constexpr (BUF) {
        const unsigned s0 = (unsigned)(y0 - bl) * rowb;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) lastv[i] = smrf_buf_load<SMRF_NT_LAST != 0>(rs_last, xcb, s0 + (unsigned)i * rowb, T());
      } else {
      const T* l0 = a.last + (long long)y0 * a.ld + xc;
#pragma unroll
      for (int i = 0; i < ROWS; ++i)
        lastv[i] = SMRF_NT_LAST ? __builtin_nontemporal_load(&l0[(long long)i * a.ld]) : l0[(long long)i * a.ld];
      }
    } else {
      SMRF_MARK("rare");
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        int yo = y0 + i;
        yo = yo < 0 ? 0 : (yo >= a.out_rows ? a.out_rows - 1 : yo);
        lastv[i] = a.last[(long long)yo * a.ld + smrf_rare<RARE>(xc)];
      }
    }
    SMRF_MARK("last");
  };

  prefetch();
  int par = 0;                                           // which level-0 copy this batch uses
  for (int yy0 = ystart; yy0 < ye + R; yy0 += ROWS, par ^= 1) {
    // (1) stage the prefetched rows into this batch's level-0 copy.  The other copy may still be
    //     read by a slower wave (its own cells of the previous batch); the higher levels are only
    //     written after the barrier below, which every wave reaches after its previous consume.
    SMRF_MARK("stage");
    T2 v[NP][NPOS];
    T2 vh[H::NJ];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < (BAL ? 1 : NPOS); ++i) {
        v[p][i] = pf[p][i];
        if (i < NPOS - 1 || has_last)
          lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + (p * NLEV + par) * WP + tid + i * TW), v[p][i]);
      }
    }
    if constexpr (BAL) {
#pragma unroll
      for (int j = 0; j < H::NJ; ++j) {
        vh[j] = pfh[j];
        const int q = hl.wave + H::WAVES * j;
        if (hl.act && q < H::NQ)
          lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + ((q / H::NH) * NLEV + par) * WP + hl.pos), vh[j]);
      }
    }
    lds_wait<0>();                                         // the compiler does not count asm stores: complete them before the barrier
    phase_sync();
    SMRF_MARK("epilogue");
    if (yy0 > ystart) epilogue(yy0 - ROWS);
    SMRF_MARK("prefetch");
    if (yy0 + ROWS < ye + R) prefetch();
    SMRF_MARK("last");
    load_last(yy0);
    SMRF_MARK("build");

    if constexpr (BAL) {
      // ring_build_consume with the halo cells as wave-jobs: every wave builds its own 256 cells' worth of each row
      // pair plus its share of the halo, so the waves reach the phase barriers together
      __builtin_amdgcn_s_setprio(SMRF_RING_BUILD_PRIO);
#ifndef SMRF_RING_DBG_NOBUILD
      ring_base<T, R, DIL, TW, NP, 1, 0>(L, par, tid, true, v);
      SMRF_MARK("build:halo");
      ring_base_halo<T, R, DIL, TW, NP>(L, par, hl, vh);
      SMRF_MARK("build");
      phase_sync();
      if constexpr (C::J > C::JB) {
        ring_upper<T, R, DIL, TW, NP, 1, 0>(L, tid, true, v);
        ring_upper_halo<T, R, DIL, TW, NP>(L, hl, vh);
        phase_sync();
      }
#endif
      __builtin_amdgcn_s_setprio(0);
      ring_consume<T, R, DIL, TW, NP>(L, par, tid, acc, outv);
    } else {
      ring_build_consume<T, R, DIL, TW, NP, NPOS, 0>(L, par, tid, has_last, v, acc, outv, phase_sync);
    }
  }
  SMRF_MARK("tail");
  {
    const int nb = (ye + R - ystart + ROWS - 1) / ROWS;
    epilogue(ystart + (nb - 1) * ROWS);
  }
#ifdef SMRF_RING_DBG_TS   // timing experiment only (tools/experiments/ring_tails.py): when and where this workgroup ran, left in
  if (tid == 0 && x0 + 6 <= a.cols) {   // the first cells of its segment's first row (the RESULT IS WRONG there)
    __threadfence();
    unsigned xcc, hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    T* const o = a.out + (long long)(ys - a.out_row0) * a.ld + x0;
    o[0] = (T)(float)(dbg_ts0 & 0x7fffff);
    o[1] = (T)(float)(__builtin_amdgcn_s_memrealtime() & 0x7fffff);
    o[2] = (T)(float)(xcc & 0xf);
    o[3] = (T)(float)(hwid & 0xffff);
    o[4] = (T)(float)(blockIdx.y * gridDim.x + blockIdx.x);
    o[5] = (T)(float)(ye - ys);
  }
#endif
#ifdef SMRF_RING_DBG_CLOCK
  if (bx == gridDim.x / 2 && by == gridDim.y / 2 && tid == 0) {
    __threadfence();
    a.out[0] = (T)(float)(__builtin_amdgcn_s_memtime() - dbg_t0);
    a.out[1] = (T)(float)(__builtin_amdgcn_s_memrealtime() - dbg_q0);
  }
#endif
}

template <typename T, int R, bool DIL, int NP>
int ring_launch_np(const DiskArgs<T>& a_in, hipStream_t stream, bool probe_only, int* seg_if_launched) {
  constexpr int TW = SMRF_RING_TW_OF(T, R);
  using C = RingCfg<T, R, TW, NP>;
  auto kern = ring_kernel<T, R, DIL, TW, NP>;
  // workgroups one CU really holds (registers + LDS), per device: the attribute below is per device too
  static int resident_of[64] = {0};
  int dev = 0;
  SMRF_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return smrf_fail(SMRF_E_UNSUPPORTED, "device index %d out of range", dev);
  int resident = __atomic_load_n(&resident_of[dev], __ATOMIC_ACQUIRE);   // host threads may launch one radius at once
  if (resident == 0) {
    if (C::LDS_BYTES > 48 * 1024)
      SMRF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    int nb = 0;
    SMRF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), TW,
                                                              C::LDS_BYTES));
    resident = std::max(1, nb);
    __atomic_store_n(&resident_of[dev], resident, __ATOMIC_RELEASE);
    if (smrf_sw().ring_debug)
      fprintf(stderr, "smrf ring: R=%d %s%s NP=%d G=%d%s LDS=%zu built for %d waves/SIMD, %d workgroups/CU resident\n", R,
              sizeof(T) == 4 ? "f32" : "f64", DIL ? " dilate" : " erode", NP, C::G, C::INPLACE ? " in-place" : "", C::LDS_BYTES,
              C::OCC, resident);
  }
  DiskArgs<T> a = a_in;
  const int strips = (a.cols + TW - 1) / TW;
  if (a.seg <= 0) {
    // output rows per workgroup: one round (every workgroup resident at once, the longest segments, the fewest re-read
    // halo rows) is the fastest from radius 20 up and as fast as any below (tools/ring_tune.py cur@SMRF_RING_ROUNDS=n);
    // how many rows of segments that round holds - all slots on a large raster, the CUs k times over on a small one -
    // is smrf_pick_nseg's cost model (seg_rule.h)
    const int rounds = smrf_sw().ring_rounds;
    const int nseg = smrf_pick_nseg(a.out_rows, strips, resident, rounds, 2 * R, C::ROWS, std::max(32, 4 * R), smrf_sw().seg_rule);
    int seg = (a.out_rows + nseg - 1) / nseg;
    a.seg = seg;
  }
  a.seg = ((a.seg + C::ROWS - 1) / C::ROWS) * C::ROWS;
  const int seg_equal = a.seg;
  int grid_y = (a.out_rows + a.seg - 1) / a.seg;
  // Segments of unequal length.  All workgroups of a one-round launch start within ~1 us, but they do not run at one speed:
  // a CU's SIMDs issue oldest wave first, so the workgroup that reached a CU first finishes first - measured per workgroup
  // (tools/experiments/ring_tails.py, profiles/r05_segment_balance.md): the k-th workgroup of a CU takes 5-9 % longer than the
  // (k-1)-th at every radius, the launch lasts as long as the youngest, and the workgroups are resident for only 0.87-0.90 of
  // it on average.  Workgroups are dealt to the CUs in dispatch order, 256 at a time (8 XCDs x 32 CUs), and ring_kernel's tile
  // mapping makes `by` grow with the dispatch id: segment `by` is of residency class (by * strips + strips / 2) / 256 (where most
  // of its workgroups are), and the segments of class c get 1 + slope * ((classes - 1) / 2 - c) times the mean length.  Any
  // segmentation gives the same bits.
  a.seg_cls = 0;
  a.plain_tiles = smrf_sw().xcd_remap ? 0 : 1;
  {
    // Measured (profiles/r05_segment_balance.md): -1.4 ... -2.1 % of the 16384^2 step at 60 permille per class (40 ... 100 are
    // within 0.3 % of it), nothing for fp64 (its classes differ by 3 %), and -1 ... +1 % where the classes do not fall on
    // whole rows of segments (strips does not divide 256) - so the built-in slope is for fp32 rasters whose strips do;
    // SMRF_RING_SLOPE=n asks for n whatever the shape.
    const int slope_env = smrf_sw().ring_slope;
    const int slope = slope_env >= 0 ? slope_env : (sizeof(T) == 4 && 256 % strips == 0 ? SMRF_RING_SLOPE_DEFAULT : 0);
    const bool one_round = a_in.seg <= 0 && smrf_sw().ring_rounds == 1;
    const auto cls_of = [&](int by) { return std::min(7, (int)(((long long)by * strips + strips / 2) / 256)); };
    const int ncls = cls_of(grid_y - 1) + 1;
    if (slope > 0 && one_round && strips <= 256 && ncls >= 2) {
      int n[8] = {0};
      for (int by = 0; by < grid_y; ++by) n[cls_of(by)]++;
      double wsum = 0.0, w[8];
      for (int c = 0; c < ncls; ++c) {
        w[c] = 1.0 + 1e-3 * slope * (0.5 * (ncls - 1) - c);
        wsum += w[c] * n[c];
      }
      const double base = (double)a.out_rows / wsum;
      int first = 0, row0 = 0, longest = 0;
      bool ok = true;
      for (int c = 0; c < 8; ++c) {
        int len = C::ROWS;
        if (c < ncls) {
          len = std::max(1, (int)(base * w[c] / C::ROWS + 0.999)) * C::ROWS;      // up to a multiple of the batch
          ok = ok && n[c] > 0 && 2 * len >= std::max(32, 4 * R);
        }
        a.seg_first[c] = c < ncls ? first : grid_y;
        a.seg_row0[c] = row0;
        a.seg_len[c] = len;
        if (c < ncls) {
          first += n[c];
          row0 += n[c] * len;
          longest = std::max(longest, len);
        }
      }
      if (ok && row0 >= a.out_rows) {
        a.seg_cls = ncls;
        a.seg = longest;                                                    // what the span clamp below looks at
      }
    }
  }
  if constexpr (ring_buf_on<T, R, TW, NP>()) {
    // buffer addressing: a workgroup's row offsets are 32-bit (and not range-checked by the hardware): keep the span of
    // a segment (its rows + warm-up + one batch) below 2 GiB
    const long long rowb = (long long)a.ld * (long long)sizeof(T);
    const long long max_rows = ((1ll << 31) - 1) / rowb - (4 * R + 4 * C::ROWS);
    if (max_rows < C::ROWS) return smrf_fail(SMRF_E_UNSUPPORTED, "raster rows of %lld bytes are too long for this build", rowb);
    if (a.seg > max_rows) {
      a.seg_cls = 0;
      a.seg = std::min(seg_equal, (int)(max_rows / C::ROWS) * C::ROWS);
    }
  }
  if (a.seg_cls == 0) grid_y = (a.out_rows + a.seg - 1) / a.seg;
  if (seg_if_launched) *seg_if_launched = a.seg_cls ? seg_equal : a.seg;   // (the dual rule looks at the mean length)
  if (probe_only) return SMRF_OK;
  dim3 grid(strips, grid_y);
  hipLaunchKernelGGL(kern, grid, dim3(TW), C::LDS_BYTES, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

template <typename T, int R, bool DIL>
int ring_launch(const DiskArgs<T>& a_in, hipStream_t stream) {
  constexpr int TW = SMRF_RING_TW_OF(T, R);
  constexpr int NP = SMRF_RING_NP(T, R);                 // the shifting ring's row pairs per batch where a radius is dual
  if constexpr (ring_tuned_inplace_dual<T>(R) && SMRF_RING_INPLACE(T, R)) {
    // Both forms exist (ring_inpl.inc): the in-place instance runs one more workgroup per CU (3 or 4 waves per SIMD), worth
    // 3-7 % when its segments are long and a loss when they are not (more segments, each with 2R warm-up rows: +3...11 %
    // on a 4096^2 raster or a 2048-row band).  Taken when the segments it would march are >= 8 R rows (16 R until round 5:
    // measured on mid-size rasters, profiles/r05_logs/segments/dual.log - 10000 x 12000, 625-row segments: -7 ... -14 %
    // at R = 40..50; 8193^2, 357 rows: -6 ... -7 % at R = 39..44 (8-9 R), +-1.5 % at 45..50 (7-8 R); 132..250-row
    // segments of 5000^2, 4096^2 and a 2048-row band: +3 ... +10 % at R >= 45, i.e. 2.6-5 R).
    constexpr int NPI = SMRF_RING_INPLACE_NP(T, R);
    static_assert(NPI != NP, "a dual radius needs two different instances");
    static_assert(RingCfg<T, R, TW, NPI>::INPLACE && !RingCfg<T, R, TW, NP>::INPLACE, "dual instances mixed up");
    const int mode = smrf_sw().ring_dual;    // tests / A-B runs: 0 = shifting ring, 1 = in place, -1 = by rule
    int seg = a_in.seg;                      // (a forced segment length is judged like the library's own)
    if (mode != 0 && a_in.seg <= 0) {
      if (int rc = ring_launch_np<T, R, DIL, NPI>(a_in, stream, true, &seg)) return rc;
    }
    if (mode == 1 || (mode < 0 && seg >= 8 * R)) return ring_launch_np<T, R, DIL, NPI>(a_in, stream, false, nullptr);
  }
  return ring_launch_np<T, R, DIL, NP>(a_in, stream, false, nullptr);
}

}  // namespace smrf
