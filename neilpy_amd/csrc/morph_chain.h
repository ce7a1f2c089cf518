// Several consecutive small progressive_filter windows in ONE launch (gfx950): open(R1) -> flag -> open(R2) -> flag ...
//
// progressive_filter (neilpy.py:1667-1676) feeds every opening to the next window.  For the small disks a window is pure
// traffic: even the fused opening of morph_fused.h moves 10 B/cell/window (read `last`, write `opened`, flags) and runs at
// the device's copy rate.  Chaining k windows in one workgroup reads `last` once and writes only the k-th opened surface:
// (2s + 2k) / k B/cell/window, and the arithmetic of the small disks (about 4R + 2 min/max per cell and window here)
// fits beside that traffic.
//
// One workgroup = TW = 256 lanes = 256 columns, one column per lane, marching down the rows of its segment NP row PAIRS
// per batch ({row A, row B} interleaved per cell as in morph_ring.h).  Every window is two STAGES (erosion, dilation) of
// the same shape: the lanes write their column's NP cells into the stage's row buffer in LDS, one barrier, then every
// lane reads the R cells either side of its own (2R ds_read_b64 per pair, no table: the disk's K window minima are grown
// cell by cell from the lane's own value, which for R <= 9 costs less than building even one table level) and folds the
// pair into the stage's register ring (2R accumulators, the shifting ring of morph_ring.h).  The rows a stage completes
// are the next stage's input, in registers; nothing but the buffers' cells goes through LDS and nothing but the first
// input and the last output through HBM.
//   - A stage of radius R delays its rows by R and leaves R more garbage columns at either edge of the strip (its
//     buffer's pad cells are never written).  A chain's outputs are therefore valid in the strip's inner
//     TW - 2 M columns, M = sum(2 R_i): strips advance by that much; rows start S = M rows early.
//   - Rows and columns outside the raster are taken at reflected coordinates, and because the reflect-extended raster is
//     symmetric about every border so is every surface computed from it: what a stage sees beyond a border IS scipy's
//     reflect of the surface it filters (the argument of morph_fused.h, applied stage by stage).
//   - Flag step of window i (last_i - opened_i > thr_i, neilpy.py:1671-1674): last_i is the surface entering the window,
//     whose rows passed through the lane 2 R_i rows earlier; they wait in a register queue.  Every window's flags and
//     the last window's opened rows are written one batch late (stores older than the loads the loop waits for).
//     A cell's flags are all written by the workgroup that owns its row and column, by the same lane in window order,
//     so when_dropped keeps the LAST window that flagged it, as in the reference.
// The NaN rule of scipy's filters is not implemented: the host takes this path for NaN-free rasters only.
#pragma once
#include "morph_ring.h"

// arguments of one chain launch over a row band (the opening of window i uses radii[i], thr[i], widx[i])
template <typename T>
struct ChainArgs {
  const T* in;        // surface entering the first window; first row held = global row in_row0
  T* out;             // opened surface of the LAST window; first row = global row out_row0
  uint8_t* mask;      // first row = out_row0 (may be NULL: no flag step)
  uint8_t* when;      // may be NULL
  double thr[4];
  float thr_lo[4];    // largest float <= thr[i] (fp32 rasters: decides `diff > thr` exactly, one instruction; set by chain.hip)
  int widx[4];
  int img_rows, cols;
  long long ld;
  int in_row0, in_rows, out_row0, out_rows;
  int seg;            // output rows per workgroup (0: the launcher decides)
  int nt;             // output cells as non-temporal stores
  int dense0;         // the first window writes EVERY mask / when byte (the call's first window: planes not cleared)
  unsigned* nan_flag; // if not NULL: set to 1 when any input cell this launch loads is NaN (it loads every cell of the band)
};

// chain dispatch (chain.hip): the pattern a window list starts with (-1: none), its length and halo rows, the launch
SMRF_HIDDEN int smrf_chain_match(int elem_size, const int32_t* windows, int n, long long cells);
SMRF_HIDDEN int smrf_chain_length(int pattern);
SMRF_HIDDEN int smrf_chain_halo(int pattern);
SMRF_HIDDEN int smrf_chain_f32(int pattern, const ChainArgs<float>& a, hipStream_t s);
SMRF_HIDDEN int smrf_chain_f64(int pattern, const ChainArgs<double>& a, hipStream_t s);

// row pairs per batch of a chain kernel, per dtype and pattern index of chain.hip (tuning builds: -DSMRF_CHAIN_NP_ALL=n)
#ifndef SMRF_CHAIN_NP_ALL
#define SMRF_CHAIN_NP_ALL 0
#endif
#ifndef SMRF_CHAIN_NP
// (round 4: the single R = 8, pattern 8, takes 4 - 0.668 -> 0.642 ms on 16384^2; 4 pairs lose or tie at every other single and at the chain 4, 5:
// profiles/r04_logs/chain_np_occ_ab.log)
#define SMRF_CHAIN_NP(T, PAT) (SMRF_CHAIN_NP_ALL ? SMRF_CHAIN_NP_ALL : (sizeof(T) == 4 ? ((PAT) <= 2 || (PAT) == 8 ? 4 : 3) : 1))
#endif

namespace smrf {

template <int NP_, int R0, int R1, int R2, int R3>
struct ChainCfg {
  static constexpr int NP = NP_, ROWS = 2 * NP_, TW = 256;
  static constexpr int NW = (R0 > 0) + (R1 > 0) + (R2 > 0) + (R3 > 0);
  static constexpr int radius(int i) { return i == 0 ? R0 : i == 1 ? R1 : i == 2 ? R2 : R3; }
  static constexpr int RM = R3 > 0 ? R3 : R2 > 0 ? R2 : R1 > 0 ? R1 : R0;     // radii ascend: the largest
  static constexpr int delay(int i) {                    // rows between the chain's input and window i's opened rows
    int d = 0;
    for (int j = 0; j <= i; ++j) d += 2 * radius(j);
    return d;
  }
  static constexpr int S = delay(NW - 1);                // rows of delay = garbage columns per side of the whole chain
  static constexpr int TWO = TW - 2 * S;                 // columns a workgroup owns
  static constexpr int kq(int i) { return (2 * radius(i) + ROWS - 1) / ROWS; }   // batches a window's `last` waits
  static constexpr int KQM = (2 * RM + ROWS - 1) / ROWS;
  // stage buffers ({A, B} cells): TWO, used alternately by the 2 NW stages of a batch (NP rows of TW + 2 RM cells each).
  // Stage s + 2 overwrites the buffer stage s read: between them lies the barrier of stage s + 1, which a wave only
  // reaches after its reads of stage s; the first stage of the next batch is behind the barrier of this batch's last.
  static constexpr int WB = TW + 2 * RM;                 // cells per buffered row (a stage of radius R uses TW + 2R of them)
  static constexpr int BUF = NP * WB;
  static constexpr int CELLS = 2 * BUF;
};

// One stage: the lanes' cells `io` (NP pairs of this lane's column) are already in `buf` (cell R + tid of row p) and a
// barrier has passed.  Reads the R cells either side, grows the disk's window minima from the own cell, folds the pair
// into the ring; io <- the two rows each pair completes (input rows - R).
// Neighbour cells per read group of a stage (0: all 2R reads of a pair at once, the form of R <= 10 in fp32).  The reads of a
// pair are 4R registers in fp32 and 8R in fp64 when taken at once - what made every fp64 single-window form spill.  In groups
// of G cells per side, the next group in flight while this one is folded into the growing window, they are 8G (16G)
// registers whatever R: the fp64 singles of chain.hip (R = 4, 5, 7, 8; 112-160 registers).  The same form for fp32
// R = 11..14 was built and measured 12-16 % slower than the fused kernels (chain.hip): tuning builds only.
#ifndef SMRF_CHAIN_GROUP
#define SMRF_CHAIN_GROUP(T, R) chain_group<T>(R)
#endif
template <typename T> constexpr int chain_group(int r) { return sizeof(T) == 4 ? (r >= 11 ? 4 : 0) : (r >= 4 ? 2 : 0); }

// The grouped form of chain_stage: the same cells, the same min / max in the same order (the window grows by one cell per
// side at a time; a width of the disk is a snapshot of the growing value), hence bit-identical results.
template <typename T, int R, bool DIL, int NP, int WB, int NACC, int G>
__device__ __forceinline__ void chain_stage_grouped(typename Vec2<T>::type* const buf, const int tid, T (&acc)[NACC],
                                                    typename Vec2<T>::type (&io)[NP]) {
  using S = DiskShape<R>;
  using T2 = typename Vec2<T>::type;
  static_assert(NACC >= 2 * R && G >= 1, "ring too short / no group size");
  constexpr int K = S::K, KR1 = S::kidx(R - 1);
  constexpr int NGR = (R + G - 1) / G;                     // read groups per pair
  constexpr int NQ = NP * NGR;                             // ... per stage: one pipeline across the pairs
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(buf + tid);
  T2 lft[2][G], rgt[2][G];
  auto issue = [&]<int Q>(std::integral_constant<int, Q>) {
    constexpr int P = Q / NGR, c0 = (Q % NGR) * G;
    [&]<int... Cc>(std::integer_sequence<int, Cc...>) {
      (([&] {
         if constexpr (c0 + Cc < R) {
           lft[Q & 1][Cc] = lds_read2<(P * WB + R - 1 - (c0 + Cc)) * (int)sizeof(T2)>(base, T());
           rgt[Q & 1][Cc] = lds_read2<(P * WB + R + 1 + (c0 + Cc)) * (int)sizeof(T2)>(base, T());
         }
       }()), ...);
    }(std::make_integer_sequence<int, G>{});
  };
  issue(std::integral_constant<int, 0>{});
  auto pair_body = [&]<int P>(std::integral_constant<int, P>) {
       T ra[K], rb[K];
       ra[0] = io[P].x;
       rb[0] = io[P].y;
       T a = ra[0], b = rb[0];
       [&]<int... Gi>(std::integer_sequence<int, Gi...>) {
         (([&] {
            constexpr int Q = P * NGR + Gi, c0 = Gi * G;
            if constexpr (Q + 1 < NQ) {
              constexpr int nxt = ((Q + 1) % NGR + 1) * G <= R ? G : R - ((Q + 1) % NGR) * G;   // cells per side of the next group
              issue(std::integral_constant<int, Q + 1>{});
              lds_wait<2 * nxt>();                           // the next group's reads may stay in flight
            } else {
              lds_wait<0>();
            }
            [&]<int... Cc>(std::integer_sequence<int, Cc...>) {
              (([&] {
                 constexpr int c = c0 + Cc;
                 if constexpr (c < R) {
                   a = op3<DIL>(a, lft[Q & 1][Cc].x, rgt[Q & 1][Cc].x);
                   b = op3<DIL>(b, lft[Q & 1][Cc].y, rgt[Q & 1][Cc].y);
                   // c + 1 cells per side reached: a width of the disk?
                   [&]<int... Kk>(std::integer_sequence<int, Kk...>) {
                     (([&] { if constexpr (S::wk(Kk + 1) == c + 1) { ra[Kk + 1] = a; rb[Kk + 1] = b; } }()), ...);
                   }(std::make_integer_sequence<int, K - 1>{});
                 }
               }()), ...);
            }(std::make_integer_sequence<int, G>{});
          }()), ...);
       }(std::make_integer_sequence<int, NGR>{});
       const T o0 = op2<DIL>(acc[0], ra[0]);                // the two rows this pair completes
       const T o1 = op3<DIL>(acc[1], ra[KR1], rb[0]);
       [&]<int... Sl>(std::integer_sequence<int, Sl...>) {  // ring: slot s <- slot s + 2 and this pair (morph_ring.h)
         (([&] {
            constexpr int da = R - Sl - 2 < 0 ? -(R - Sl - 2) : R - Sl - 2, db = R - Sl - 1 < 0 ? -(R - Sl - 1) : R - Sl - 1;
            acc[Sl] = op3<DIL>(acc[Sl + 2], ra[S::kidx(da)], rb[S::kidx(db)]);
          }()), ...);
       }(std::make_integer_sequence<int, 2 * R - 2>{});
       acc[2 * R - 2] = op2<DIL>(ra[0], rb[KR1]);
       acc[2 * R - 1] = rb[0];
       io[P].x = o0;
       io[P].y = o1;
  };
  [&]<int... P>(std::integer_sequence<int, P...>) {
    (pair_body(std::integral_constant<int, P>{}), ...);
  }(std::make_integer_sequence<int, NP>{});
}

template <typename T, int R, bool DIL, int NP, int WB, int NACC>
__device__ __forceinline__ void chain_stage(typename Vec2<T>::type* const buf, const int tid, T (&acc)[NACC],
                                            typename Vec2<T>::type (&io)[NP]) {
  using S = DiskShape<R>;
  using T2 = typename Vec2<T>::type;
  static_assert(NACC >= 2 * R, "ring too short");
  if constexpr (SMRF_CHAIN_GROUP(T, R) > 0) {
    chain_stage_grouped<T, R, DIL, NP, WB, NACC, SMRF_CHAIN_GROUP(T, R)>(buf, tid, acc, io);
    return;
  }
  constexpr int K = S::K, KR1 = S::kidx(R - 1);
  constexpr bool AHEAD = R <= 4 && NP > 1;               // the next pair's reads in flight while this pair is folded
  const unsigned base = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(buf + tid);   // cell tid = column - R
  T2 lft[AHEAD ? 2 : 1][R], rgt[AHEAD ? 2 : 1][R];       // cells at distance c + 1 either side
  auto issue = [&]<int P>(std::integral_constant<int, P>) {
    [&]<int... Cc>(std::integer_sequence<int, Cc...>) {
      ((lft[AHEAD ? (P & 1) : 0][Cc] = lds_read2<(P * WB + R - 1 - Cc) * (int)sizeof(T2)>(base, T()),
        rgt[AHEAD ? (P & 1) : 0][Cc] = lds_read2<(P * WB + R + 1 + Cc) * (int)sizeof(T2)>(base, T())), ...);
    }(std::make_integer_sequence<int, R>{});
  };
  if constexpr (AHEAD) issue(std::integral_constant<int, 0>{});
  [&]<int... P>(std::integer_sequence<int, P...>) {
    (([&] {
       constexpr int B = AHEAD ? (P & 1) : 0;
       if constexpr (AHEAD && P + 1 < NP) {
         issue(std::integral_constant<int, P + 1>{});
         lds_wait<2 * R>();                                 // the next pair's reads may stay in flight
       } else {
         if constexpr (!AHEAD) issue(std::integral_constant<int, P>{});
         lds_wait<0>();
       }
       T ra[K], rb[K];
       ra[0] = io[P].x;
       rb[0] = io[P].y;
       [&]<int... Kk>(std::integer_sequence<int, Kk...>) {  // width k from width k - 1, one cell per side at a time
         (([&] {
            constexpr int k = Kk + 1;
            T a = ra[k - 1], b = rb[k - 1];
#pragma unroll
            for (int c = S::wk(k - 1); c < S::wk(k); ++c) {
              a = op3<DIL>(a, lft[B][c].x, rgt[B][c].x);
              b = op3<DIL>(b, lft[B][c].y, rgt[B][c].y);
            }
            ra[k] = a;
            rb[k] = b;
          }()), ...);
       }(std::make_integer_sequence<int, K - 1>{});
       const T o0 = op2<DIL>(acc[0], ra[0]);                // the two rows this pair completes
       const T o1 = op3<DIL>(acc[1], ra[KR1], rb[0]);
       [&]<int... Sl>(std::integer_sequence<int, Sl...>) {  // ring: slot s <- slot s + 2 and this pair (morph_ring.h)
         (([&] {
            constexpr int da = R - Sl - 2 < 0 ? -(R - Sl - 2) : R - Sl - 2, db = R - Sl - 1 < 0 ? -(R - Sl - 1) : R - Sl - 1;
            acc[Sl] = op3<DIL>(acc[Sl + 2], ra[S::kidx(da)], rb[S::kidx(db)]);
          }()), ...);
       }(std::make_integer_sequence<int, 2 * R - 2>{});
       acc[2 * R - 2] = op2<DIL>(ra[0], rb[KR1]);
       acc[2 * R - 1] = rb[0];
       io[P].x = o0;
       io[P].y = o1;
     }()), ...);
  }(std::make_integer_sequence<int, NP>{});
}

template <typename T, int NP, int OCC, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__(256, OCC)
void chain_kernel(const ChainArgs<T> a) {
  using C = ChainCfg<NP, R0, R1, R2, R3>;
  using T2 = typename Vec2<T>::type;
  constexpr int NW = C::NW, ROWS = C::ROWS, TW = C::TW, S = C::S, TWO = C::TWO, RM = C::RM, KQM = C::KQM;
  static_assert(TWO >= 64, "chain too long for a 256-column strip");
  extern __shared__ __attribute__((aligned(16))) unsigned char smrf_lds[];
  T2* const L = reinterpret_cast<T2*>(smrf_lds);

  const int tid = threadIdx.x;
  const int xe0 = (int)blockIdx.x * TWO - S;             // column of lane 0
  const int x = xe0 + tid;
  const bool owns = tid >= S && tid < TW - S && x < a.cols;
  const int cx = smrf_fold(x, a.cols);                   // the lane's input column (reflected beyond the raster)
  const int ys = a.out_row0 + (int)blockIdx.y * a.seg;   // global output rows [ys, ye)
  const int ye = min(a.out_row0 + a.out_rows, ys + a.seg);
  const int last_in = a.in_rows - 1;
  const bool flag = a.mask != nullptr;
  float thr_lo[NW];                                      // largest float <= thr[i] (see the flag step; set by the host)
#pragma unroll
  for (int i = 0; i < NW; ++i) thr_lo[i] = a.thr_lo[i];

  T accE[NW][2 * RM], accD[NW][2 * RM];
#pragma unroll
  for (int i = 0; i < NW; ++i)
#pragma unroll
    for (int s = 0; s < 2 * RM; ++s) { accE[i][s] = ident<T>(false); accD[i][s] = ident<T>(true); }
  T2 dq[NW][KQM][NP];                                    // window i: its input surface's rows of the last kq(i) batches
#pragma unroll
  for (int i = 0; i < NW; ++i)
#pragma unroll
    for (int s = 0; s < KQM; ++s)
#pragma unroll
      for (int p = 0; p < NP; ++p) { dq[i][s][p].x = T(0); dq[i][s][p].y = T(0); }
  T2 pf[NP];
  T outv[ROWS];                                          // the last window's opened rows of a batch, stored one batch late
  bool hit[NW][ROWS];                                    // every window's flags of a batch, likewise
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    outv[r] = T(0);
#pragma unroll
    for (int i = 0; i < NW; ++i) hit[i][r] = false;
  }

  // input row yy completes the chain's output row yy - S; start so that a batch's outputs are all inside or all outside
  // the segment, S rows before the first output's own footprint
  constexpr int DELTA = (ROWS - (2 * S) % ROWS) % ROWS;
  const int ystart = ys - S - DELTA;
  RowFold rf(ystart, a.img_rows);
  auto prefetch = [&]() {
    const int l0 = rf.p - a.in_row0;
    if (rf.p + ROWS <= rf.n && l0 >= 0 && l0 + ROWS - 1 <= last_in) {
      const T* r0 = a.in + (long long)l0 * a.ld + cx;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        pf[p].x = r0[(long long)(2 * p) * a.ld];
        pf[p].y = r0[(long long)(2 * p + 1) * a.ld];
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        int la = rf.at(2 * p) - a.in_row0;
        int lb = rf.at(2 * p + 1) - a.in_row0;
        la = la < 0 ? 0 : (la > last_in ? last_in : la);   // only rows outside the segment's halo clamp
        lb = lb < 0 ? 0 : (lb > last_in ? last_in : lb);
        pf[p].x = a.in[(long long)la * a.ld + cx];
        pf[p].y = a.in[(long long)lb * a.ld + cx];
      }
    }
    rf.advance(ROWS);
  };
  // stores of the batch whose first input row was yyb: window i's flags belong to rows yyb - delay(i) + r, the output
  // to rows yyb - S + r; a row is written by the workgroup whose segment holds it
  auto epilogue = [&](int yyb) {
    if (!owns) return;
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (([&] {
         if (flag) {
           const int y0 = yyb - C::delay(I);
           const bool dense = I == 0 && a.dense0;
           const long long off0 = (long long)(y0 - a.out_row0) * a.ld + x;
           if (y0 >= ys && y0 + ROWS <= ye && !dense) {       // the common case: the uniform tests once per window and batch
             bool any = false;
#pragma unroll
             for (int r = 0; r < ROWS; ++r) any = any || hit[I][r];
             if (any) {
#pragma unroll
               for (int r = 0; r < ROWS; ++r)
                 if (hit[I][r]) {
                   a.mask[off0 + (long long)r * a.ld] = 1;
                   if (a.when != nullptr) a.when[off0 + (long long)r * a.ld] = (uint8_t)a.widx[I];
                 }
             }
           } else {
#pragma unroll
             for (int r = 0; r < ROWS; ++r) {
               const int y = y0 + r;
               if (y >= ys && y < ye) {
                 const long long off = off0 + (long long)r * a.ld;
                 if (dense) {
                   a.mask[off] = hit[I][r] ? 1 : 0;
                   if (a.when != nullptr) a.when[off] = hit[I][r] ? (uint8_t)a.widx[I] : (uint8_t)0;
                 } else if (hit[I][r]) {
                   a.mask[off] = 1;
                   if (a.when != nullptr) a.when[off] = (uint8_t)a.widx[I];
                 }
               }
             }
           }
         }
       }()), ...);
    }(std::make_integer_sequence<int, NW>{});
    const int yob = yyb - S;
    if (yob >= ys) {                                       // aligned: all of the batch's rows are at or after ys
      const long long off0 = (long long)(yob - a.out_row0) * a.ld + x;
      if (yob + ROWS <= ye) {
        if (a.nt) {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) __builtin_nontemporal_store(outv[r], &a.out[off0 + (long long)r * a.ld]);
        } else {
#pragma unroll
          for (int r = 0; r < ROWS; ++r) a.out[off0 + (long long)r * a.ld] = outv[r];
        }
      } else {
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
          if (yob + r < ye) smrf_store_out(&a.out[off0 + (long long)r * a.ld], outv[r], a.nt);
      }
    }
  };

  // the call's NaN scan riding along (smrf_progressive_filter_* with nan_aware < 0): every input cell passes through some
  // lane's prefetch, so one unordered compare per loaded value replaces a separate pass over the raster
  const bool scan = a.nan_flag != nullptr;
  bool seen_nan = false;
  prefetch();
  for (int yy0 = ystart; yy0 < ye + S; yy0 += ROWS) {
    T2 cur[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) cur[p] = pf[p];
    if (scan) {
#pragma unroll
      for (int p = 0; p < NP; ++p) seen_nan = seen_nan || cur[p].x != cur[p].x || cur[p].y != cur[p].y;
    }
    if (yy0 > ystart) epilogue(yy0 - ROWS);                // stores older than the loads issued next
    if (yy0 + ROWS < ye + S) prefetch();
    [&]<int... I>(std::integer_sequence<int, I...>) {
      (([&] {
         constexpr int R = C::radius(I), KQ = C::kq(I), WB = C::WB;
         T2* const bufE = L;                                // stages alternate between the two buffers
         T2* const bufD = L + C::BUF;
         // `last` of the rows this window completes in this batch: input row e of the batches since b - KQ
         T lastv[ROWS];
#pragma unroll
         for (int r = 0; r < ROWS; ++r) {
           const int e = r - 2 * R + KQ * ROWS, slot = e / ROWS, w = e % ROWS;
           const T2 src = slot < KQ ? dq[I][slot < KQ ? slot : 0][w / 2] : cur[w / 2];
           lastv[r] = (w & 1) ? src.y : src.x;
         }
#pragma unroll
         for (int sl = 0; sl + 1 < KQ; ++sl)
#pragma unroll
           for (int p = 0; p < NP; ++p) dq[I][sl][p] = dq[I][sl + 1][p];
#pragma unroll
         for (int p = 0; p < NP; ++p) dq[I][KQ - 1][p] = cur[p];
         // erosion stage
#pragma unroll
         for (int p = 0; p < NP; ++p)
           lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(bufE + p * WB + R + tid), cur[p]);
         lds_wait<0>();
         __syncthreads();
         chain_stage<T, R, false, NP, WB>(bufE, tid, accE[I], cur);
         // dilation stage
#pragma unroll
         for (int p = 0; p < NP; ++p)
           lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(bufD + p * WB + R + tid), cur[p]);
         lds_wait<0>();
         __syncthreads();
         chain_stage<T, R, true, NP, WB>(bufD, tid, accD[I], cur);
         // flag step of this window's rows (stored with the next batch's epilogue): the raster dtype's difference
         // compared in float64 (NumPy 2) - for fp32 rasters against thr_lo, the largest float <= thr, which decides
         // every float `diff` exactly as the float64 comparison does (diff > thr >= thr_lo one way; the other way diff >
         // thr_lo means diff >= the next float, which lies above thr) and costs one instruction instead of three
#pragma unroll
         for (int r = 0; r < ROWS; ++r) {
           const T opened = (r & 1) ? cur[r / 2].y : cur[r / 2].x;
           const T diff = lastv[r] - opened;                 // raster dtype
           if constexpr (sizeof(T) == 4) hit[I][r] = diff > thr_lo[I];
           else hit[I][r] = (double)diff > a.thr[I];
         }
       }()), ...);
    }(std::make_integer_sequence<int, NW>{});
#pragma unroll
    for (int r = 0; r < ROWS; ++r) outv[r] = (r & 1) ? cur[r / 2].y : cur[r / 2].x;
  }
  {
    const int nb = (ye + S - ystart + ROWS - 1) / ROWS;
    epilogue(ystart + (nb - 1) * ROWS);
  }
  if (scan && seen_nan) atomicOr(a.nan_flag, 1u);
}

template <typename T, int NP, int OCC, int R0, int R1, int R2, int R3>
int chain_launch(const ChainArgs<T>& a_in, hipStream_t stream) {
  using C = ChainCfg<NP, R0, R1, R2, R3>;
  using T2 = typename Vec2<T>::type;
  constexpr size_t LDS = (size_t)C::CELLS * sizeof(T2);
  auto kern = chain_kernel<T, NP, OCC, R0, R1, R2, R3>;
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
    SMRF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), C::TW, LDS));
    resident = std::max(1, nb);
    __atomic_store_n(&resident_of[dev], resident, __ATOMIC_RELEASE);
    if (smrf_sw().ring_debug)
      fprintf(stderr, "smrf chain: R=%d,%d,%d,%d %s NP=%d LDS=%zu, %d workgroups/CU resident\n", R0, R1, R2, R3,
              sizeof(T) == 4 ? "f32" : "f64", NP, LDS, resident);
  }
  ChainArgs<T> a = a_in;
  const int strips = (a.cols + C::TWO - 1) / C::TWO;
  if (a.seg <= 0) {
    // Asks for three workgroups per resident slot (round 3: chain 4, 5 on 16384^2 1.17 -> 0.87 ms - the counters showed 2.4
    // waves per SIMD where 4 were expected: with what round 5's per-workgroup timestamps show, that was a one-round launch's
    // tail - its youngest workgroups end 3 % per residency class behind - and, at the time, a second, nearly empty round
    // whenever the segment count rounded up).  smrf_pick_nseg (seg_rule.h) weighs the three rounds against the best single
    // round and takes the cheaper: three on the benchmark raster, one below ~10^8 cells.
    const int rounds = smrf_sw().chain_rounds;
    const int nseg = smrf_pick_nseg(a.out_rows, strips, resident, rounds, 2 * C::S, C::ROWS, std::max(32, 4 * C::S), smrf_sw().seg_rule);
    int seg = (a.out_rows + nseg - 1) / nseg;
    a.seg = seg;
  }
  a.seg = ((a.seg + C::ROWS - 1) / C::ROWS) * C::ROWS;
  dim3 grid(strips, (a.out_rows + a.seg - 1) / a.seg);
  hipLaunchKernelGGL(kern, grid, dim3(C::TW), LDS, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // namespace smrf
