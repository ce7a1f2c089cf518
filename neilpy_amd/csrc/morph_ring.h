// Register-ring disk erosion / dilation for gfx950 (MI355X), one kernel per radius.
//
// What it computes (scipy.ndimage.grey_erosion/grey_dilation with skimage's disk(R) and
// mode='reflect', as reached from neilpy.py:1670):
//     E[y, x] = min over dy in [-R, R] of  min over |dx| <= w(dy) of  Z[fold(y+dy), fold(x+dx)],
//     w(dy) = isqrt(R*R - dy*dy)
//
// How: a workgroup owns a strip of TW columns (one column per lane) and marches down the rows of
// its segment.  For every input row it stages TW+2R cells into LDS with coalesced loads, builds
// the row's power-of-two window minima (a per-row sparse table, log2(2R+1) levels) and each lane
// looks up the K distinct window minima R_w[x] of the disk (K ~ 0.59 R + 1, two LDS reads each).
// The 2R+1 output rows that the input row contributes to are 2R+1 accumulators held in
// REGISTERS: slot s belongs to output row y_in - R + s.  Moving to the next input row shifts
// every slot down by one, which costs nothing because the update
//     acc[s] = min(acc[s+1], R_w(dy = R - s))
// writes a different register than it reads; with R a template parameter every slot index and
// every window width is a compile-time constant, so the whole update is 2R straight-line
// v_min_f32 (v_max for dilation) on fixed registers.  acc[0] is complete after the update and is
// written out (coalesced).  No MFMA: there is no contraction in this computation; the kernel is
// bound by LDS reads + VALU min/max, HBM traffic is ~2 plane passes.
#pragma once
#include <utility>

#include "smrf_common.h"

namespace smrf {

constexpr int clog2(int v) {  // floor(log2(v)), v >= 1
  int l = 0;
  while ((2 << l) <= v) ++l;
  return l;
}

template <int R>
struct DiskShape {
  static constexpr int halfw(int dy) { return smrf_isqrt(R * R - dy * dy); }
  // true when w(dy) is the first occurrence of its value walking dy = R, R-1, ..., 0
  static constexpr bool first(int dy) { return dy == R || halfw(dy) != halfw(dy + 1); }
  // index of w(dy) in the ascending list of distinct half-widths (dy in [0, R])
  static constexpr int kidx(int dy) {
    int k = 0;
    for (int d = R - 1; d >= dy; --d)
      if (halfw(d) != halfw(d + 1)) ++k;
    return k;
  }
  static constexpr int K = kidx(0) + 1;  // number of distinct half-widths
  static constexpr int wk(int k) {       // k-th distinct half-width
    for (int d = R; d >= 0; --d)
      if (kidx(d) == k) return halfw(d);
    return 0;
  }
  static constexpr int J = clog2(2 * R + 1);  // highest table level
};

template <bool DIL>
__device__ __forceinline__ float op2(float a, float b) {
  float r;
  // single instruction, no canonicalising v_max in front (IEEE minNum/maxNum: a NaN operand loses)
  if constexpr (DIL) asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}
template <bool DIL>
__device__ __forceinline__ double op2(double a, double b) {
  double r;
  if constexpr (DIL) asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  else asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
  return r;
}

template <typename T> __device__ __forceinline__ T ident(bool dil);
template <> __device__ __forceinline__ float ident<float>(bool dil) { return dil ? -INFINITY : INFINITY; }
template <> __device__ __forceinline__ double ident<double>(bool dil) { return dil ? -(double)INFINITY : (double)INFINITY; }

template <typename T> __device__ __forceinline__ T qnan();
template <> __device__ __forceinline__ float qnan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double qnan<double>() { return __builtin_nan(""); }

template <typename T, int R, int TW, int B>
struct RingCfg {
  using S = DiskShape<R>;
  static constexpr int W = TW + 2 * R;                 // staged cells per row
  static constexpr int PAD = 1 << (S::J > 0 ? S::J - 1 : 0);
  static constexpr int WP = ((W + PAD + 3) / 4) * 4;   // row pitch of one table level
  static constexpr int ROWP = (S::J + 1) * WP;         // cells per staged row (all levels)
  static constexpr size_t LDS_BYTES = (size_t)(B * ROWP + PAD) * sizeof(T);
  static constexpr int E = sizeof(T) / 4;
  static constexpr int NEED = E * ((2 * R + 1) + S::K + 4 * B + 8) + 28;   // VGPR estimate
  static constexpr int OCC = NEED <= 64 ? 8 : NEED <= 96 ? 5 : NEED <= 128 ? 4 : NEED <= 168 ? 3 : NEED <= 256 ? 2 : 1;
};

template <typename T, int R, bool DIL, int TW, int B>
__global__ __launch_bounds__(TW, (RingCfg<T, R, TW, B>::OCC))
void ring_kernel(const DiskArgs<T> a) {
  using C = RingCfg<T, R, TW, B>;
  using S = typename C::S;
  constexpr int J = S::J, K = S::K, WP = C::WP, ROWP = C::ROWP;
  extern __shared__ __attribute__((aligned(16))) unsigned char smrf_lds[];
  T* const L = reinterpret_cast<T*>(smrf_lds);          // [B][J+1][WP]

  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * TW;
  const int x = x0 + tid;
  const int ys = a.out_row0 + blockIdx.y * a.seg;                       // global output rows [ys, ye)
  const int ye = min(a.out_row0 + a.out_rows, ys + a.seg);
  const bool has2 = tid < 2 * R;
  const int c0 = smrf_fold(x0 - R + tid, a.cols);
  const int c1 = has2 ? smrf_fold(x0 - R + tid + TW, a.cols) : 0;
  const int last_in = a.in_rows - 1;

  T acc[2 * R + 1];
#pragma unroll
  for (int i = 0; i <= 2 * R; ++i) acc[i] = ident<T>(DIL);

  T pf0[B], pf1[B];
  auto prefetch = [&](int yy0) {
#pragma unroll
    for (int b = 0; b < B; ++b) {
      int ly = smrf_fold(yy0 + b, a.img_rows) - a.in_row0;
      ly = ly < 0 ? 0 : (ly > last_in ? last_in : ly);   // only rows past the segment's halo clamp
      const T* row = a.in + (long long)ly * a.ld;
      pf0[b] = row[c0];
      pf1[b] = has2 ? row[c1] : T(0);
    }
  };
  prefetch(ys - R);

  for (int yy0 = ys - R; yy0 < ye + R; yy0 += B) {
    T v0[B], v1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      v0[b] = pf0[b];
      v1[b] = pf1[b];
      L[b * ROWP + tid] = v0[b];
      if (has2) L[b * ROWP + tid + TW] = v1[b];
    }
    __syncthreads();
    if (yy0 + B < ye + R) prefetch(yy0 + B);            // next batch, consumed next iteration
#pragma unroll
    for (int j = 1; j <= J; ++j) {
      const int h = 1 << (j - 1);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const T* Lm = L + b * ROWP + (j - 1) * WP;
        T* Lj = L + b * ROWP + j * WP;
        v0[b] = op2<DIL>(v0[b], Lm[tid + h]);
        Lj[tid] = v0[b];
        if (has2) {
          v1[b] = op2<DIL>(v1[b], Lm[tid + TW + h]);
          Lj[tid + TW] = v1[b];
        }
      }
      __syncthreads();
    }

#pragma unroll 1
    for (int b = 0; b < B; ++b) {
      const T* q = L + b * ROWP + tid + R;               // this lane's cell in level 0
      T rv[K];
      rv[0] = q[0];
      // window minima for the K distinct half-widths, 8 lookups per group: loads first, then mins
      [&]<int... G>(std::integer_sequence<int, G...>) {
        (([&] {
           constexpr int k0 = 1 + 8 * G;
           constexpr int n = (K - k0) < 8 ? (K - k0) : 8;
           T ta[8], tb[8];
           [&]<int... I>(std::integer_sequence<int, I...>) {
             (([&] {
                constexpr int w = S::wk(k0 + I);
                constexpr int j = clog2(2 * w + 1);
                ta[I] = q[j * WP - w];
                tb[I] = q[j * WP + w - (1 << j) + 1];
              }()), ...);
           }(std::make_integer_sequence<int, n>{});
           __builtin_amdgcn_sched_barrier(0);
           [&]<int... I>(std::integer_sequence<int, I...>) {
             ((rv[k0 + I] = op2<DIL>(ta[I], tb[I])), ...);
           }(std::make_integer_sequence<int, n>{});
           __builtin_amdgcn_sched_barrier(0);
         }()), ...);
      }(std::make_integer_sequence<int, (K - 1 + 7) / 8>{});

      // ring update: slot s <- slot s+1 combined with this row's window for dy = R - s
      [&]<int... Sl>(std::integer_sequence<int, Sl...>) {
        (([&] {
           constexpr int dy = R - Sl;
           constexpr int k = S::kidx(dy < 0 ? -dy : dy);
           acc[Sl] = op2<DIL>(acc[Sl + 1], rv[k]);
         }()), ...);
      }(std::make_integer_sequence<int, 2 * R>{});
      acc[2 * R] = rv[0];

      const int yo = yy0 + b - R;                         // output row completed by this input row
      if (yo >= ys && yo < ye && x < a.cols) {
        T val = acc[0];
        const long long off = (long long)(yo - a.out_row0) * a.ld + x;
        if (a.nan_aware) {
          // scipy: the first visited footprint element (offset (-R, 0)) decides NaN-ness
          int ly = smrf_fold(yo - R, a.img_rows) - a.in_row0;
          const T first = a.in[(long long)ly * a.ld + x];
          if (first != first) val = qnan<T>();
        }
        a.out[off] = val;
        if (a.mask != nullptr) {
          const T diff = a.last[off] - val;               // raster dtype
          if ((double)diff > a.thr) {                     // float64 comparison (NumPy 2)
            a.mask[off] = 1;
            if (a.when != nullptr) a.when[off] = (uint8_t)a.widx;
          }
        }
      }
    }
    __syncthreads();
  }
}

template <typename T, int R, bool DIL>
int ring_launch(const DiskArgs<T>& a, hipStream_t stream) {
  constexpr int TW = 256;
  constexpr int B = sizeof(T) == 4 ? 4 : 2;
  using C = RingCfg<T, R, TW, B>;
  auto kern = ring_kernel<T, R, DIL, TW, B>;
  static bool attr_done = false;
  if (!attr_done) {
    if (C::LDS_BYTES > 48 * 1024)
      SMRF_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    attr_done = true;
  }
  dim3 grid((a.cols + TW - 1) / TW, (a.out_rows + a.seg - 1) / a.seg);
  hipLaunchKernelGGL(kern, grid, dim3(TW), C::LDS_BYTES, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // namespace smrf
