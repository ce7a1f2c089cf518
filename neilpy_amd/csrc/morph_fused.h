// Fused grey opening + progressive_filter flag step for the small disks (gfx950), one launch per window.
//
// progressive_filter (neilpy.py:1667-1676) opens `last` with disk(R) and flags `last - opened > thr`.  As two ring
// launches that is 22 B/cell of HBM traffic in fp32 (erosion 4 + 4, dilation + flag 4 + 4 + 4 + 2); for R <= ~14
// both launches run at or near the device's copy bandwidth, so the traffic is the time (which radii take this kernel:
// smrf_fused_radius() in smrf_common.h, and the raster-size rule in morph.hip).  This kernel chains the two ring
// stages of morph_ring.h inside one workgroup: the eroded rows never leave the CU (they become level 0 of the second
// stage's table in LDS) and `last` is read from HBM once - 10 B/cell.
//
//   - A workgroup of TW = 256 lanes erodes TW columns [xe0, xe0 + TW) and opens the TW - 2R inner ones (the dilation
//     of a column needs the eroded columns R either side of it): strips advance by TW - 2R columns.
//   - Rows: one march over the input rows; the erosion stage completes eroded row y - R when input row y arrives,
//     the dilation stage opened row y - 2R.  Rows and columns outside the raster are never special: the input is
//     taken at reflected coordinates, the reflect-extended raster is symmetric about every border, so is its
//     erosion by the symmetric disk, and an "eroded row -3" computed this way IS what scipy's reflect hands the
//     dilation there (eroded[fold(-3)]).  A segment therefore only starts 2R rows earlier than a plain ring pass.
//   - Both stages are ring_build_consume of morph_ring.h (tables, lookups, register ring), min for the first,
//     max for the second, one after the other: six barriers per batch instead of three.
// The NaN rule of scipy's filters is not implemented here: the host only takes this path for NaN-free rasters.
#pragma once
#include "morph_ring.h"

namespace smrf {

// Row pairs per batch of the fused kernel (both stages), per radius.  More pairs spread the six barriers of a batch
// over more rows, fewer pairs halve the two tables' LDS and let 8 instead of 4 (3 at R = 8) workgroups share a CU;
// measured per radius on the 16384^2 benchmark DEM (gpurun_out/r02/fused3_per_radius_f32.log).  Running the two
// stages' phases side by side under three barriers per batch (second stage one batch behind) was measured too:
// no faster at any radius, slower at most (fused2_per_radius_f32.log), so the stages simply follow each other.
#ifndef SMRF_FUSED_NP_ALL
#define SMRF_FUSED_NP_ALL 0     // tuning builds: this many row pairs per batch at every fp32 radius
#endif
template <typename T>
constexpr int fused_np(int r) {
  if (SMRF_FUSED_NP_ALL && sizeof(T) == 4) return SMRF_FUSED_NP_ALL;
  return sizeof(T) == 4 ? ((r == 2 || r == 3 || r == 8 || r == 10) ? 1 : 2) : 1;
}

// `last` for the flag step from a register queue instead of a second read from memory.  The flag step of opened row y
// needs last[y], which the workgroup staged 2R rows earlier as erosion input; read again from global memory it comes
// from HBM, not from the L2 (PMC, profiles/r03_fused_flag_traffic.md: +1.0 GiB of reads per window at every radius
// from 2 up, 0.10-0.18 ms of a 0.6-0.9 ms launch - eight workgroups per CU stream more rows through an XCD's 4 MB L2
// in 2R + 2 batches than it holds).  The queue keeps the lane's own column of the last ceil(2R / ROWS) batches in
// registers (2 * NP per batch, moved down once per batch: v_mov at 2.1 cycles).  Per radius, measured.
#ifndef SMRF_FUSED_LASTQ
#define SMRF_FUSED_LASTQ(T, R) fused_lastq<T>(R)
#endif
template <typename T>
constexpr bool fused_lastq(int r) {   // profiles/r03_logs/lastq_ab_f32.log: fp32 R = 2..7 -9...-17 %, R >= 8 lose a workgroup per CU to the queue's registers
  return sizeof(T) == 4 ? (r >= 2 && r <= 7) : (r == 1 || r == 2 || r == 4 || r == 5);
}

// workgroups per CU the kernel is built for: what the two tables' LDS allows, at most 4 (128 registers per lane)
#ifndef SMRF_FUSED_BLOCKS
#define SMRF_FUSED_BLOCKS(R) 4
#endif
template <typename T, int R, int TW, int NP>
constexpr int fused_min_blocks() {
  const int by_lds = (int)(160 * 1024 / (2 * RingCfg<T, R, TW, NP>::LDS_BYTES));
  const int by_regs = R >= 19 ? 2 : R >= 15 ? 3 : SMRF_FUSED_BLOCKS(R);   // two rings of 2R registers: 128 per lane no longer hold them from R = 15
  const int cap = by_lds < by_regs ? by_lds : by_regs;
  return cap < 1 ? 1 : cap;
}

template <typename T, int R, int TW, int NP>
__global__ __launch_bounds__(TW, (fused_min_blocks<T, R, TW, NP>()))
void fused_open_kernel(const DiskArgs<T> a) {
  using C = RingCfg<T, R, TW, NP>;
  using T2 = typename Vec2<T>::type;
  constexpr int WP = C::WP, ROWS = C::ROWS, NLEV = C::NLEV, NPOS = C::NPOS, W = C::W;
  constexpr int TWO = TW - 2 * R;                        // opened columns per workgroup
  constexpr int TABLE = NP * NLEV * WP + C::PAD;         // cells of one stage's tables
  static_assert(TWO > 0, "radius too large for the fused kernel");
  extern __shared__ __attribute__((aligned(16))) unsigned char smrf_lds[];
  T2* const LE = reinterpret_cast<T2*>(smrf_lds);        // erosion stage:  [NP][NLEV][WP] of {row A, row B}
  T2* const LD = LE + TABLE;                             // dilation stage: same geometry, cells R .. R + TW - 1 filled

  const int tid = threadIdx.x;
  int bx = blockIdx.x, by = blockIdx.y;                  // XCD-aware placement, as ring_kernel
#if SMRF_RING_XCD_REMAP
  if ((gridDim.x & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.x >> 3;
    const int xcd = id & 7, slot = id >> 3;
    bx = xcd * per + slot % per;
    by = slot / per;
  }
#endif
  const int xe0 = bx * TWO - R;                          // first eroded column of the workgroup
  const int x = xe0 + tid;                               // this lane's column (eroded; opened if it is an inner one)
  const bool writes = tid >= R && tid < TW - R && x < a.cols;
  const int ys = a.out_row0 + by * a.seg;                // global output rows [ys, ye)
  const int ye = min(a.out_row0 + a.out_rows, ys + a.seg);
  const bool has_last = tid + (NPOS - 1) * TW < W;
  int cpos[NPOS];                                        // input columns of the lane's staged cells (reflected)
#pragma unroll
  for (int i = 0; i < NPOS; ++i) cpos[i] = smrf_fold(xe0 - R + tid + (tid + i * TW < W ? i * TW : 0), a.cols);
  const int last_in = a.in_rows - 1;
  auto phase_sync = [&]() { __syncthreads(); };
  const bool flag = a.mask != nullptr;
  const int xc = x < 0 ? 0 : (x < a.cols ? x : a.cols - 1);

  T accE[2 * R], accD[2 * R];
#pragma unroll
  for (int i = 0; i < 2 * R; ++i) { accE[i] = ident<T>(false); accD[i] = ident<T>(true); }
  T2 pf[NP][NPOS];
  T outvE[ROWS], outvD[ROWS], lastv[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) { outvE[i] = T(0); outvD[i] = T(0); lastv[i] = T(0); }

  // input row yy completes eroded row yy - R and opened row yy - 2R; start so that a batch's opened rows are all
  // inside or all outside the segment
  constexpr int DELTA = (ROWS - (4 * R) % ROWS) % ROWS;
  const int ystart = ys - 2 * R - DELTA;
  RowFold rf(ystart, a.img_rows);
  auto prefetch = [&]() {
    const int l0 = rf.p - a.in_row0;
    if (rf.p + ROWS <= rf.n && l0 >= 0 && l0 + ROWS - 1 <= last_in) {
      const T* r0 = a.in + (long long)l0 * a.ld;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
#pragma unroll
        for (int i = 0; i < NPOS; ++i) {
          pf[p][i].x = r0[(long long)(2 * p) * a.ld + cpos[i]];
          pf[p][i].y = r0[(long long)(2 * p + 1) * a.ld + cpos[i]];
        }
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        int la = rf.at(2 * p) - a.in_row0;
        int lb = rf.at(2 * p + 1) - a.in_row0;
        la = la < 0 ? 0 : (la > last_in ? last_in : la);
        lb = lb < 0 ? 0 : (lb > last_in ? last_in : lb);
        const T* ra = a.in + (long long)la * a.ld;
        const T* rb = a.in + (long long)lb * a.ld;
#pragma unroll
        for (int i = 0; i < NPOS; ++i) { pf[p][i].x = ra[cpos[i]]; pf[p][i].y = rb[cpos[i]]; }
      }
    }
    rf.advance(ROWS);
  };
  auto emit = [&](long long off, T val, T lastval) {       // one opened cell, general form
    smrf_store_out(&a.out[off], val, a.nt);
    if (flag) smrf_flag_cell(a, off, lastval, val);
  };
  // a batch whose ROWS opened rows are all inside the segment, as straight-line code per (store kind, flag step: none,
  // sparse, dense): the uniform tests once per batch instead of once per cell
  auto emit_rows = [&]<bool NT, int MODE>(std::bool_constant<NT>, std::integral_constant<int, MODE>, long long off0)
                       __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      const long long off = off0 + (long long)i * a.ld;
      if constexpr (NT) __builtin_nontemporal_store(outvD[i], &a.out[off]);
      else a.out[off] = outvD[i];
      if constexpr (MODE != 0) {
        const T diff = lastv[i] - outvD[i];                  // raster dtype
        bool hit;                                            // float64 comparison (NumPy 2); fp32: smrf_float_below
        if constexpr (sizeof(T) == 4) hit = diff > a.thr_lo;
        else hit = (double)diff > a.thr;
        if constexpr (MODE == 2) {                           // first window of a call: every byte, the planes were not cleared
          a.mask[off] = hit ? 1 : 0;
          if (a.when != nullptr) a.when[off] = hit ? (uint8_t)a.widx : (uint8_t)0;
        } else if (hit) {
          a.mask[off] = 1;
          if (a.when != nullptr) a.when[off] = (uint8_t)a.widx;
        }
      }
    }
  };
  auto epilogue = [&](int yyb) {                           // opened rows of the batch whose first input row was yyb
    const int yob = yyb - 2 * R;
    if (yob < ys || !writes) return;
    const long long off0 = (long long)(yob - a.out_row0) * a.ld + x;
    if (yob + ROWS <= ye) {
      using I0 = std::integral_constant<int, 0>;
      using I1 = std::integral_constant<int, 1>;
      using I2 = std::integral_constant<int, 2>;
      if (!flag) {
        if (a.nt) emit_rows(std::true_type{}, I0{}, off0); else emit_rows(std::false_type{}, I0{}, off0);
      } else if (!a.dense) {
        if (a.nt) emit_rows(std::true_type{}, I1{}, off0); else emit_rows(std::false_type{}, I1{}, off0);
      } else {
        if (a.nt) emit_rows(std::true_type{}, I2{}, off0); else emit_rows(std::false_type{}, I2{}, off0);
      }
    } else {
#pragma unroll
      for (int i = 0; i < ROWS; ++i)
        if (yob + i < ye) emit(off0 + (long long)i * a.ld, outvD[i], lastv[i]);
    }
  };
  auto load_last = [&](int yyb) {                          // `last` at the rows this batch opens (L2: read 2R rows ago)
    if (!flag) return;
    const int y0 = yyb - 2 * R - a.out_row0;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
      int yo = y0 + i;
      yo = yo < 0 ? 0 : (yo >= a.out_rows ? a.out_rows - 1 : yo);
      lastv[i] = a.last[(long long)yo * a.ld + xc];
    }
  };

  constexpr bool LASTQ = SMRF_FUSED_LASTQ(T, R);
  constexpr int KQ = (2 * R + ROWS - 1) / ROWS;          // the batch that staged a row this batch opens: KQ (or KQ - 1) back
  T2 dq[LASTQ ? KQ : 1][NP];                             // the lane's own column of batches b - KQ .. b - 1
#pragma unroll
  for (int s = 0; s < (LASTQ ? KQ : 1); ++s)
#pragma unroll
    for (int p = 0; p < NP; ++p) { dq[s][p].x = T(0); dq[s][p].y = T(0); }

  prefetch();
  int par = 0;
  for (int yy0 = ystart; yy0 < ye + 2 * R; yy0 += ROWS, par ^= 1) {
    // ---- erosion stage: stage the prefetched input rows, build, consume -> eroded rows yy0 - R ...
    T2 v[NP][NPOS];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int i = 0; i < NPOS; ++i) {
        v[p][i] = pf[p][i];
        if (i < NPOS - 1 || has_last)
          lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(LE + (p * NLEV + par) * WP + tid + i * TW), v[p][i]);
      }
    }
    lds_wait<0>();                                         // asm stores: complete them before the barrier
    phase_sync();
    if (yy0 > ystart) epilogue(yy0 - ROWS);                // stores older than the loads issued next
    if (yy0 + ROWS < ye + 2 * R) prefetch();
    if constexpr (LASTQ) {
      // opened row yy0 - 2R + i is input row e = i - 2R + KQ * ROWS of the batches since b - KQ: slot e / ROWS (slot KQ =
      // this batch, whose own-column cells are staged at tid + R), row e % ROWS of it
      T2 own[NP];
#pragma unroll
      for (int p = 0; p < NP; ++p)
        own[p] = lds_read2<0>((unsigned)(size_t)(__attribute__((address_space(3))) void*)(LE + (p * NLEV + par) * WP + tid + R), T());
      lds_wait<0>();
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        const int e = i - 2 * R + KQ * ROWS, slot = e / ROWS, w = e % ROWS;
        const T2 src = slot < KQ ? dq[slot < KQ ? slot : 0][w / 2] : own[w / 2];
        lastv[i] = (w & 1) ? src.y : src.x;
      }
#pragma unroll
      for (int sl = 0; sl + 1 < KQ; ++sl)
#pragma unroll
        for (int p = 0; p < NP; ++p) dq[sl][p] = dq[sl + 1][p];
#pragma unroll
      for (int p = 0; p < NP; ++p) dq[KQ - 1][p] = own[p];
    } else {
      load_last(yy0);
    }
    ring_build_consume<T, R, false, TW, NP, NPOS, 0>(LE, par, tid, has_last, v, accE, outvE, phase_sync);
    // ---- dilation stage: the eroded rows are its level 0 (cell R + tid), build, consume -> opened rows yy0 - 2R ...
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      v[p][0].x = outvE[2 * p];
      v[p][0].y = outvE[2 * p + 1];
      lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(LD + (p * NLEV + par) * WP + tid + R), v[p][0]);
    }
    lds_wait<0>();
    phase_sync();
    ring_build_consume<T, R, true, TW, NP, 1, R>(LD, par, tid, true, v, accD, outvD, phase_sync);
  }
  {
    const int nb = (ye + 2 * R - ystart + ROWS - 1) / ROWS;
    epilogue(ystart + (nb - 1) * ROWS);
  }
}

template <typename T, int R>
int fused_launch(const DiskArgs<T>& a_in, hipStream_t stream) {
  constexpr int TW = 256;
  constexpr int NP = fused_np<T>(R);
  using C = RingCfg<T, R, TW, NP>;
  constexpr size_t LDS = 2 * C::LDS_BYTES;
  auto kern = fused_open_kernel<T, R, TW, NP>;
  static int resident_of[64] = {0};
  int dev = 0;
  SMRF_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return smrf_fail(SMRF_E_UNSUPPORTED, "device index %d out of range", dev);
  int resident = __atomic_load_n(&resident_of[dev], __ATOMIC_ACQUIRE);
  if (resident == 0) {
    if (LDS > 48 * 1024)
      SMRF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)LDS));
    int nb = 0;
    SMRF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), TW, LDS));
    resident = std::max(1, nb);
    __atomic_store_n(&resident_of[dev], resident, __ATOMIC_RELEASE);
    if (smrf_sw().ring_debug)
      fprintf(stderr, "smrf fused: R=%d %s NP=%d LDS=%zu, %d workgroups/CU resident\n", R, sizeof(T) == 4 ? "f32" : "f64", NP,
              LDS, resident);
  }
  DiskArgs<T> a = a_in;
  constexpr int TWO = TW - 2 * R;
  const int strips = (a.cols + TWO - 1) / TWO;
  if (a.seg <= 0) {
    const int rounds = smrf_sw().fused_rounds;
    // one round: every workgroup resident; a segment re-reads 4R warm-up rows (smrf_pick_nseg, seg_rule.h)
    const int nseg = smrf_pick_nseg(a.out_rows, strips, resident, rounds, 4 * R, C::ROWS, std::max(32, 8 * R), smrf_sw().seg_rule);
    int seg = (a.out_rows + nseg - 1) / nseg;
    a.seg = seg;
  }
  a.seg = ((a.seg + C::ROWS - 1) / C::ROWS) * C::ROWS;
  dim3 grid(strips, (a.out_rows + a.seg - 1) / a.seg);
  hipLaunchKernelGGL(kern, grid, dim3(TW), LDS, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // namespace smrf
