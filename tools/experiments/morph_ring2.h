// Two-role register-ring disk erosion / dilation for the large radii (gfx950, fp32).
//
// Same result and the same LDS tables as smrf::ring_kernel (morph_ring.h), but the 2R + 1 rows of the disk are shared
// between TWO waves per 64 columns, so that a workgroup is 8 waves on the 256-column strip and the kernel fits 4 waves
// per SIMD (128 VGPRs) where the one-wave ring (2R accumulators + 2K window results) only fits 2.  A lone wave issues
// one v_min3 per 8.3 cycles, two per SIMD 5.6, four 4.8 (tools/ubench/mix_rate.hip): the consume phase of the large
// radii is bound by exactly that.
//
//   role I (waves 0-3):  rows |dy| <= D0 of the disk - the wide windows.   Ring of 2 D0 slots.
//   role O (waves 4-7):  rows D0 <= |dy| <= R - the narrow windows.        Two rings: dy in [-R, -D0] and [D0, R].
//
// An output row y collects dy = -R .. +R in time order: first in O's upper ring, then in I's ring, then in O's lower
// ring, from which it leaves complete.  The two hand-overs go through LDS, one {row, row+1} cell per lane and pair, and
// are folded in one batch later (min/max is idempotent and order-free, so the receiving ring starts from its own first
// row and takes the carried value in when it is there; both roles cover |dy| = D0, which lines a batch's retired rows up
// with the top 2 NP slots of the receiving ring at the next batch's start).  No window lookup and no table build is
// done twice; the cost is 2 LDS writes + 2 reads + 2 min per output row and one duplicated slot per ring.
#pragma once
#include "morph_ring.h"

namespace smrf {

// ring over the disk rows dy in [DLO, DHI]: N contributions per output row, N - 1 partial rows between pairs.
// Between pairs slot t belongs to output row (next input row) - DHI + t.
template <int R, int DLO, int DHI>
struct RangeCfg {
  using S = DiskShape<R>;
  static constexpr int N = DHI - DLO + 1;
  static constexpr int NS = N - 1;
  static_assert(N >= 4, "range too short for the pairwise ring update");
  static constexpr int kd(int d) { return S::kidx(d < 0 ? -d : d); }
  static constexpr int kA(int t) { return kd(DHI - t - 2); }   // new slot t in [0, N - 3): row A's width index
  static constexpr int kB(int t) { return kd(DHI - t - 1); }   //                            row B's
};

template <typename T, int R, int NP, int D0>
struct Ring2Cfg {
  using C = RingCfg<T, R, 256, NP>;
  using S = DiskShape<R>;
  static constexpr int NT = 512;                         // lanes per workgroup (two roles x 256 columns)
  static constexpr int KD = S::kidx(D0);                 // width index of |dy| = D0
  static constexpr int ROWS = 2 * NP;
  static_assert(D0 >= NP && R - D0 >= ROWS && D0 >= 2 && KD >= 2 && KD < S::K - 1, "split point out of range");
  static_assert(C::W <= NT, "one staged cell per lane");
  // hand-over cells after the tables: X1 (O's upper ring -> I), X2 (I -> O's lower ring), [NP][256] of {row, row + 1}
  static constexpr size_t X_OFF = ((size_t)NP * C::NLEV * C::WP + C::PAD);   // in T2 cells
  static constexpr size_t LDS_BYTES = (X_OFF + 2 * (size_t)NP * 256) * 2 * sizeof(T);
};

// One role's consume phase of a batch: the role's window lookups (width indices KF .. KL, ascending; index 0 is the
// lane's own cell) in pipelined groups, and the pairwise update of its rings.
//   ROLE 0: one ring, dy in [-D0, D0];  ROLE 1: rings dy in [D0, R] (completes the output rows) and [-R, -D0].
template <typename T, int R, bool DIL, int NP, int D0, int ROLE>
struct Ring2Role {
  using Q = Ring2Cfg<T, R, NP, D0>;
  using C = typename Q::C;
  using S = typename Q::S;
  using T2 = typename Vec2<T>::type;
  static constexpr int K = S::K, KD = Q::KD;
  static constexpr int KF = ROLE == 0 ? KD : 1, KL = ROLE == 0 ? K - 1 : KD;   // looked-up width indices
  static constexpr int NL = KL - KF + 1;
  static constexpr int G = 4;                            // lookups per group, two groups in flight
  static constexpr int NG = (NL + G - 1) / G;
  static constexpr int grp(int k) { return k < KF ? 0 : (k - KF) / G; }
  using RA = RangeCfg<R, ROLE == 0 ? -D0 : D0, ROLE == 0 ? D0 : R>;   // first ring (ROLE 1: the one that completes rows)
  using RB = RangeCfg<R, -R, -D0>;                                      // ROLE 1 only
  static constexpr int greads(int g) {
    int n = 0;
    for (int k = KF + g * G; k < KF + (g + 1) * G && k <= KL; ++k) n += C::nreads(S::wk(k));
    return n;
  }
  // group after which a ring's retiring rows / new slot t can be computed: when their widths are there, never before
  // a lower slot (the update runs in place, upwards), and from the first slot that needs the last group on in the
  // VALU-only tail behind the loop
  template <class RG> static constexpr int gret() { int a = grp(RG::kA(-2)), b = grp(RG::kA(-1)); return a > b ? a : b; }   // widths of dy = DHI, DHI - 1
  template <class RG> static constexpr int need(int t) { int a = grp(RG::kA(t)), b = grp(RG::kB(t)); return a > b ? a : b; }
  template <class RG> static constexpr int tail0() {
    if (gret<RG>() >= NG - 1) return 0;
    for (int t = 0; t < RG::N - 3; ++t)
      if (need<RG>(t) >= NG - 1) return t;
    return RG::N - 3;
  }
  template <class RG> static constexpr int rel(int t) {  // in-loop release group of slot t < tail0
    int g = gret<RG>();
    for (int u = 0; u <= t; ++u) g = need<RG>(u) > g ? need<RG>(u) : g;
    return g;
  }
};

template <typename T, int R, bool DIL, int NP, int D0, int ROLE, int NSA, int NSB>
__device__ __forceinline__ void ring2_consume(typename Vec2<T>::type* const L, typename Vec2<T>::type* const xw, const int par,
                                              const int col, T (&accA)[NSA], T (&accB)[NSB], T (&outv)[2 * NP]) {
  using P = Ring2Role<T, R, DIL, NP, D0, ROLE>;
  using C = typename P::C;
  using S = typename P::S;
  using RA = typename P::RA;
  using RB = typename P::RB;
  using T2 = typename Vec2<T>::type;
  constexpr int K = S::K, WP = C::WP, G = P::G, NG = P::NG, NLEV = C::NLEV, KF = P::KF, KL = P::KL;
  static_assert(NSA == RA::NS && (ROLE == 0 || NSB == RB::NS), "ring sizes");
  const unsigned lds_q = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + col + R);
  T2 own[NP];
  if constexpr (ROLE == 1) {
#pragma unroll
    for (int p = 0; p < NP; ++p) own[p] = lds_read2<0>(lds_q + (p * NLEV + par) * WP * (unsigned)sizeof(T2), T());
    lds_wait<0>();
  }
#pragma unroll
  for (int p = 0; p < NP; ++p) {
    const unsigned q = lds_q + p * NLEV * WP * (unsigned)sizeof(T2);
    __builtin_amdgcn_s_setprio(SMRF_RING_LOOKUP_PRIO);
    T ra[K], rb[K];
    T2 ta[2][G], tb[2][G], tc[2][G];
    auto issue = [&]<int GI>(std::integral_constant<int, GI>) {
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = KF + GI * G + I;
           if constexpr (k <= KL) {
             constexpr int w = S::wk(k);
             constexpr int j = C::lev(w);
             constexpr int base = C::slot_of(j) * WP;
             static_assert(C::stored(j), "lookup level not built");
             ta[GI % 2][I] = lds_read2<(base - w) * (int)sizeof(T2)>(q, T());
             tb[GI % 2][I] = lds_read2<(base + w - (1 << j) + 1) * (int)sizeof(T2)>(q, T());
             if constexpr (C::nreads(w) == 3) tc[GI % 2][I] = lds_read2<(base - w + (1 << j)) * (int)sizeof(T2)>(q, T());
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    auto reduce = [&]<int GI>(std::integral_constant<int, GI>) {
      [&]<int... I>(std::integer_sequence<int, I...>) {
        (([&] {
           constexpr int k = KF + GI * G + I;
           if constexpr (k <= KL) {
             if constexpr (C::nreads(S::wk(k)) == 3) {
               ra[k] = op3<DIL>(ta[GI % 2][I].x, tc[GI % 2][I].x, tb[GI % 2][I].x);
               rb[k] = op3<DIL>(ta[GI % 2][I].y, tc[GI % 2][I].y, tb[GI % 2][I].y);
             } else {
               ra[k] = op2<DIL>(ta[GI % 2][I].x, tb[GI % 2][I].x);
               rb[k] = op2<DIL>(ta[GI % 2][I].y, tb[GI % 2][I].y);
             }
           }
         }()), ...);
      }(std::make_integer_sequence<int, G>{});
    };
    // the two rows a ring retires with this pair (read before their slots are overwritten)
    T retA0, retA1, retB0 = T(0), retB1 = T(0);
    auto retire = [&]<class RG, int NS>(RG, T (&acc)[NS], T& r0, T& r1) {
      r0 = op2<DIL>(acc[0], ra[RG::kA(-2)]);
      r1 = op3<DIL>(acc[1], ra[RG::kA(-1)], rb[RG::kB(-1)]);
    };
    // work of one ring after lookup group GI (GI = NG: the tail behind the loop)
    auto ring_step = [&]<class RG, int NS, int GI>(RG, T (&acc)[NS], T& r0, T& r1, std::integral_constant<int, GI>) {
      constexpr int gr = P::template gret<RG>(), t0 = P::template tail0<RG>();
      if constexpr ((GI < NG && gr < NG - 1 && gr == GI) || (GI == NG && gr >= NG - 1)) retire(RG{}, acc, r0, r1);
      [&]<int... Tt>(std::integer_sequence<int, Tt...>) {
        (([&] {
           constexpr bool mine = GI == NG ? Tt >= t0 : (Tt < t0 && P::template rel<RG>(Tt) == GI);
           if constexpr (mine) acc[Tt] = op3<DIL>(acc[Tt + 2], ra[RG::kA(Tt)], rb[RG::kB(Tt)]);
         }()), ...);
      }(std::make_integer_sequence<int, RG::N - 3>{});
      if constexpr (GI == NG) {                            // the two rows that start with this pair
        acc[RG::N - 3] = op2<DIL>(ra[RG::kA(RG::N - 3)], rb[RG::kB(RG::N - 3)]);
        acc[RG::N - 2] = rb[RG::kA(RG::N - 3)];
      }
    };

    issue(std::integral_constant<int, 0>{});
    if constexpr (ROLE == 1) { ra[0] = own[p].x; rb[0] = own[p].y; }
    [&]<int... GI>(std::integer_sequence<int, GI...>) {
      (([&] {
         if constexpr (GI + 1 < NG) issue(std::integral_constant<int, GI + 1>{});
         lds_wait<(GI + 1 < NG ? P::greads(GI + 1) : 0)>();
         reduce(std::integral_constant<int, GI>{});
         ring_step(RA{}, accA, retA0, retA1, std::integral_constant<int, GI>{});
         if constexpr (ROLE == 1) ring_step(RB{}, accB, retB0, retB1, std::integral_constant<int, GI>{});
       }()), ...);
    }(std::make_integer_sequence<int, NG>{});
    __builtin_amdgcn_s_setprio(0);
    ring_step(RA{}, accA, retA0, retA1, std::integral_constant<int, NG>{});
    if constexpr (ROLE == 1) ring_step(RB{}, accB, retB0, retB1, std::integral_constant<int, NG>{});
    // hand-over: ROLE 0 passes its ring's rows on to O's lower ring, ROLE 1 its upper ring's rows to I
    T2 hx;
    if constexpr (ROLE == 0) { hx.x = retA0; hx.y = retA1; }
    else { hx.x = retB0; hx.y = retB1; outv[2 * p] = retA0; outv[2 * p + 1] = retA1; }
    xw[p * 256 + col] = hx;
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int R, bool DIL, int NP, int D0, int ROLE>
__device__ __forceinline__ void ring2_run(const DiskArgs<T>& a, typename Vec2<T>::type* const L) {
  using Q = Ring2Cfg<T, R, NP, D0>;
  using C = typename Q::C;
  using P = Ring2Role<T, R, DIL, NP, D0, ROLE>;
  using T2 = typename Vec2<T>::type;
  constexpr int WP = C::WP, ROWS = C::ROWS, NLEV = C::NLEV, W = C::W;
  T2* const X1 = L + Q::X_OFF;                           // O's upper ring -> I
  T2* const X2 = X1 + NP * 256;                          // I -> O's lower ring
  T2* const xr = ROLE == 0 ? X1 : X2;                    // read at a batch's start
  T2* const xw = ROLE == 0 ? X2 : X1;                    // written pair by pair

  const int tid = threadIdx.x;
  const int col = tid & 255;
  int bx = blockIdx.x, by = blockIdx.y;
#if SMRF_RING_XCD_REMAP
  if ((gridDim.x & 7) == 0) {
    const int id = blockIdx.y * gridDim.x + blockIdx.x, per = gridDim.x >> 3;
    const int xcd = id & 7, slot = id >> 3;
    bx = xcd * per + slot % per;
    by = slot / per;
  }
#endif
  const int x0 = bx * 256;
  const int x = x0 + col;
  const int ys = a.out_row0 + by * a.seg;
  const int ye = min(a.out_row0 + a.out_rows, ys + a.seg);
  const bool has = tid < W;                              // this lane stages (and builds) cell `tid` of the TW + 2R wide row
  const int cpos = smrf_fold(x0 - R + (has ? tid : 0), a.cols);
  const int last_in = a.in_rows - 1;
  auto phase_sync = [&]() { __syncthreads(); };
  const bool flag = a.mask != nullptr;
  const int xc = x < a.cols ? x : a.cols - 1;

  T accA[P::RA::NS], accB[ROLE == 1 ? P::RB::NS : 1];
#pragma unroll
  for (int i = 0; i < P::RA::NS; ++i) accA[i] = ident<T>(DIL);
#pragma unroll
  for (int i = 0; i < (ROLE == 1 ? P::RB::NS : 1); ++i) accB[i] = ident<T>(DIL);
  {                                                      // nothing handed over yet
    T2 id2;
    id2.x = id2.y = ident<T>(DIL);
#pragma unroll
    for (int p = 0; p < NP; ++p) xw[p * 256 + col] = id2;
  }

  T2 pf[NP];
  T outv[ROWS], lastv[ROWS];
#pragma unroll
  for (int i = 0; i < ROWS; ++i) { outv[i] = T(0); lastv[i] = T(0); }

  constexpr int DELTA = (ROWS - (2 * R) % ROWS) % ROWS;
  const int ystart = ys - R - DELTA;
  RowFold rf(ystart, a.img_rows);
  auto prefetch = [&]() {
    const int l0 = rf.p - a.in_row0;
    if (rf.p + ROWS <= rf.n && l0 >= 0 && l0 + ROWS - 1 <= last_in) {
      const T* r0 = a.in + (long long)l0 * a.ld;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        pf[p].x = r0[(long long)(2 * p) * a.ld + cpos];
        pf[p].y = r0[(long long)(2 * p + 1) * a.ld + cpos];
      }
    } else {
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        int la = rf.at(2 * p) - a.in_row0;
        int lb = rf.at(2 * p + 1) - a.in_row0;
        la = la < 0 ? 0 : (la > last_in ? last_in : la);
        lb = lb < 0 ? 0 : (lb > last_in ? last_in : lb);
        pf[p].x = a.in[(long long)la * a.ld + cpos];
        pf[p].y = a.in[(long long)lb * a.ld + cpos];
      }
    }
    rf.advance(ROWS);
  };
  auto emit = [&](int yo, long long off, T val, T lastval) {
    if (a.nan_aware) {
      const int ly = smrf_fold(yo - R, a.img_rows) - a.in_row0;
      const T first = a.in[(long long)ly * a.ld + x];
      if (first != first) val = qnan<T>();
    }
    a.out[off] = val;
    if (flag) {
      const T diff = lastval - val;
      if ((double)diff > a.thr) {
        a.mask[off] = 1;
        if (a.when != nullptr) a.when[off] = (uint8_t)a.widx;
      }
    }
  };
  auto epilogue = [&](int yyb) {
    if constexpr (ROLE == 1) {
      const int yob = yyb - R;
      if (yob < ys || x >= a.cols) return;
      const long long off0 = (long long)(yob - a.out_row0) * a.ld + x;
      if (yob + ROWS <= ye) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) emit(yob + i, off0 + (long long)i * a.ld, outv[i], lastv[i]);
      } else {
#pragma unroll
        for (int i = 0; i < ROWS; ++i)
          if (yob + i < ye) emit(yob + i, off0 + (long long)i * a.ld, outv[i], lastv[i]);
      }
    }
  };
  auto load_last = [&](int yyb) {
    if constexpr (ROLE == 1) {
      if (!flag) return;
      const int y0 = yyb - R - a.out_row0;
      if (y0 >= 0 && y0 + ROWS <= a.out_rows) {
        const T* l0 = a.last + (long long)y0 * a.ld + xc;
#pragma unroll
        for (int i = 0; i < ROWS; ++i) lastv[i] = l0[(long long)i * a.ld];
      } else {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
          int yo = y0 + i;
          yo = yo < 0 ? 0 : (yo >= a.out_rows ? a.out_rows - 1 : yo);
          lastv[i] = a.last[(long long)yo * a.ld + xc];
        }
      }
    }
  };

  prefetch();
  int par = 0;
  for (int yy0 = ystart; yy0 < ye + R; yy0 += ROWS, par ^= 1) {
    T2 v[NP][C::NPOS];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      v[p][0] = pf[p];
      if (has) lds_write2((unsigned)(size_t)(__attribute__((address_space(3))) void*)(L + (p * NLEV + par) * WP + tid), v[p][0]);
    }
    lds_wait<0>();
    phase_sync();
    // the rows the other role handed over during the previous batch: taken into registers before the next barrier,
    // the writer overwrites them in its consume phase (after the build barriers)
    T2 xf[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) xf[p] = xr[p * 256 + col];
    if (yy0 > ystart) epilogue(yy0 - ROWS);
    if (yy0 + ROWS < ye + R) prefetch();
    load_last(yy0);

    __builtin_amdgcn_s_setprio(SMRF_RING_BUILD_PRIO);
    ring_base<T, R, DIL, 256, NP, 1, 0>(L, par, tid, has, v);
    phase_sync();
    if constexpr (C::J > C::JB) {
      ring_upper<T, R, DIL, 256, NP, 1, 0>(L, tid, has, v);
      phase_sync();
    }
    __builtin_amdgcn_s_setprio(0);
    // fold the handed-over rows into the top 2 NP slots of the receiving ring: they are the rows the giving ring
    // retired during the previous batch
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      constexpr int top = P::RA::NS - ROWS;
      accA[top + 2 * p] = op2<DIL>(accA[top + 2 * p], xf[p].x);
      accA[top + 2 * p + 1] = op2<DIL>(accA[top + 2 * p + 1], xf[p].y);
    }
    ring2_consume<T, R, DIL, NP, D0, ROLE>(L, xw, par, col, accA, accB, outv);
  }
  {
    const int nb = (ye + R - ystart + ROWS - 1) / ROWS;
    epilogue(ystart + (nb - 1) * ROWS);
  }
}

template <typename T, int R, bool DIL, int NP, int D0>
__global__ __launch_bounds__(512, 4) void ring2_kernel(const DiskArgs<T> a) {
  using T2 = typename Vec2<T>::type;
  extern __shared__ __attribute__((aligned(16))) unsigned char smrf_lds[];
  T2* const L = reinterpret_cast<T2*>(smrf_lds);
  // the role is uniform per wave: a scalar branch, each side with its own rings in registers and the same barriers
  if (__builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) == 0) ring2_run<T, R, DIL, NP, D0, 0>(a, L);
  else ring2_run<T, R, DIL, NP, D0, 1>(a, L);
}

// split point: the role loads (ring slots + 2 per looked-up width) as even as they get
template <int R>
constexpr int ring2_d0(int np) {
  using S = DiskShape<R>;
  int best = -1, cost = 1 << 30;
  for (int d = (np > 2 ? np : 2); R - d >= 2 * np; ++d) {
    const int kd = S::kidx(d);
    if (kd < 2 || kd >= S::K - 1) continue;
    const int li = 2 * d + 2 * (S::K - kd), lo = 2 * (R - d + 1) + 2 * kd + 8;   // O also stores the rows
    const int c = li > lo ? li : lo;
    if (c < cost) { cost = c; best = d; }
  }
  return best;
}

#ifndef SMRF_RING2_NP
#define SMRF_RING2_NP 2
#endif
#ifndef SMRF_RING2_MIN_RADIUS
#define SMRF_RING2_MIN_RADIUS 16                          // smallest radius the two-role kernels are built for
#endif
#ifndef SMRF_RING2_D0
#define SMRF_RING2_D0(R, NP) ring2_d0<R>(NP)
#endif

template <typename T, int R, bool DIL, int NP = SMRF_RING2_NP>
int ring2_launch(const DiskArgs<T>& a_in, hipStream_t stream) {
  constexpr int D0 = SMRF_RING2_D0(R, NP);
  static_assert(D0 > 0, "radius too small for the two-role ring");
  using Q = Ring2Cfg<T, R, NP, D0>;
  using C = typename Q::C;
  auto kern = ring2_kernel<T, R, DIL, NP, D0>;
  static int resident_of[64] = {0};
  int dev = 0;
  SMRF_HIP_CHECK(hipGetDevice(&dev));
  if (dev < 0 || dev >= 64) return smrf_fail(SMRF_E_UNSUPPORTED, "device index %d out of range", dev);
  int resident = __atomic_load_n(&resident_of[dev], __ATOMIC_ACQUIRE);
  if (resident == 0) {
    if (Q::LDS_BYTES > 48 * 1024)
      SMRF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                         (int)Q::LDS_BYTES));
    int nb = 0;
    SMRF_HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(kern), Q::NT, Q::LDS_BYTES));
    resident = std::max(1, nb);
    __atomic_store_n(&resident_of[dev], resident, __ATOMIC_RELEASE);
    if (smrf_env_int("SMRF_RING_DEBUG", 0))
      fprintf(stderr, "smrf ring2: R=%d %s NP=%d D0=%d LDS=%zu, %d workgroups/CU resident\n", R, DIL ? "dilate" : "erode", NP, D0,
              Q::LDS_BYTES, resident);
  }
  DiskArgs<T> a = a_in;
  const int strips = (a.cols + 255) / 256;
  if (a.seg <= 0) {
    const int rounds = smrf_env_int("SMRF_RING_ROUNDS", 1);
    const int nseg = std::max(1, (rounds * resident * 256 + strips / 2) / strips);
    int seg = (a.out_rows + nseg - 1) / nseg;
    seg = std::max(seg, std::max(32, 4 * R));
    seg = std::min(seg, a.out_rows);
    a.seg = seg;
  }
  a.seg = ((a.seg + C::ROWS - 1) / C::ROWS) * C::ROWS;
  dim3 grid(strips, (a.out_rows + a.seg - 1) / a.seg);
  hipLaunchKernelGGL(kern, grid, dim3(Q::NT), Q::LDS_BYTES, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // namespace smrf
