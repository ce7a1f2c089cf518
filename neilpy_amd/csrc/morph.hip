// Disk erosion / dilation entry points, the footprint-gather ("direct") kernel for radii the
// ring kernels do not cover, and the progressive_filter driver (neilpy.py:1659-1680).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "smrf_common.h"
#include "morph_chain.h"

namespace {

// ------------------------------------------------------------------------------------------
// direct kernel: one lane per output cell, gathers the whole footprint through L1/L2.
// O(card(disk)) loads per cell - the fallback for radius > SMRF_RING_MAX_RADIUS and the
// independent on-device cross-check of the ring kernels in the parity tests.
// ------------------------------------------------------------------------------------------
// the flag step of one output cell (neilpy.py:1671-1674): sparse, or dense when the planes were not cleared (DiskArgs::dense)
template <typename T>
__device__ __forceinline__ void flag_cell(const DiskArgs<T>& a, long long off, T lastval, T val) {
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

template <typename T, bool DIL>
__global__ __launch_bounds__(256) void direct_kernel(const DiskArgs<T> a) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = a.out_row0 + blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= a.cols || y >= a.out_row0 + a.out_rows) return;
  const int r = a.radius;
  const long long r2 = (long long)r * r;
  T best = DIL ? -INFINITY : INFINITY;
  bool first_nan = false;
  for (int dy = -r; dy <= r; ++dy) {
    const int ly = smrf_fold(y + dy, a.img_rows) - a.in_row0;
    const T* row = a.in + (long long)ly * a.ld;
    long long rem = r2 - (long long)dy * dy;
    int w = (int)sqrt((double)rem);
    while ((long long)(w + 1) * (w + 1) <= rem) ++w;
    while ((long long)w * w > rem) --w;
    for (int dx = -w; dx <= w; ++dx) {
      const T v = row[smrf_fold(x + dx, a.cols)];
      if (dy == -r && dx == 0) first_nan = (v != v);
      if (DIL) { if (v > best) best = v; } else { if (v < best) best = v; }
    }
  }
  if (a.nan_aware && first_nan) best = DIL ? (T)NAN : (T)NAN;
  const long long off = (long long)(y - a.out_row0) * a.ld + x;
  a.out[off] = best;
  if (a.mask != nullptr) flag_cell(a, off, a.last[off], best);
}

template <typename T>
__global__ __launch_bounds__(256) void copy_flag_kernel(const DiskArgs<T> a) {   // radius 0
  const long long n = (long long)a.out_rows * a.cols;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int yy = (int)(i / a.cols), x = (int)(i % a.cols);
    const T v = a.in[(long long)(a.out_row0 + yy - a.in_row0) * a.ld + x];
    const long long off = (long long)yy * a.ld + x;
    a.out[off] = v;
    if (a.mask != nullptr) flag_cell(a, off, a.last[off], v);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void count_nan_kernel(const T* __restrict__ p, long long n,
                                                        unsigned long long* __restrict__ out) {
  unsigned long long c = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const T v = p[i];
    c += (v != v);
  }
  for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

template <typename T> struct RingFn;
template <> struct RingFn<float> {
  static int call(const DiskArgs<float>& a, int d, hipStream_t s) {
    switch (a.radius % SMRF_RING_PARTS) {
      case 0: return smrf_ring_f32_p0(a, d, s);
      case 1: return smrf_ring_f32_p1(a, d, s);
      case 2: return smrf_ring_f32_p2(a, d, s);
      case 3: return smrf_ring_f32_p3(a, d, s);
      case 4: return smrf_ring_f32_p4(a, d, s);
      case 5: return smrf_ring_f32_p5(a, d, s);
      case 6: return smrf_ring_f32_p6(a, d, s);
      default: return smrf_ring_f32_p7(a, d, s);
    }
  }
};
template <> struct RingFn<double> {
  static int call(const DiskArgs<double>& a, int d, hipStream_t s) {
    switch (a.radius % SMRF_RING_PARTS) {
      case 0: return smrf_ring_f64_p0(a, d, s);
      case 1: return smrf_ring_f64_p1(a, d, s);
      case 2: return smrf_ring_f64_p2(a, d, s);
      case 3: return smrf_ring_f64_p3(a, d, s);
      case 4: return smrf_ring_f64_p4(a, d, s);
      case 5: return smrf_ring_f64_p5(a, d, s);
      case 6: return smrf_ring_f64_p6(a, d, s);
      default: return smrf_ring_f64_p7(a, d, s);
    }
  }
};

template <typename T>
int check_band(const DiskArgs<T>& a) {
  if (!a.in || !a.out) return smrf_fail(SMRF_E_ARG, "null raster pointer");
  if (a.img_rows < 1 || a.cols < 1 || a.in_rows < 1 || a.out_rows < 1 || a.radius < 0)
    return smrf_fail(SMRF_E_ARG, "bad raster size (%d x %d, in %d, out %d, radius %d)", a.img_rows,
                     a.cols, a.in_rows, a.out_rows, a.radius);
  if (a.ld < a.cols) return smrf_fail(SMRF_E_ARG, "ld %lld < cols %d", a.ld, a.cols);
  if (a.out_row0 < 0 || a.out_row0 + a.out_rows > a.img_rows || a.in_row0 < 0 ||
      a.in_row0 + a.in_rows > a.img_rows)
    return smrf_fail(SMRF_E_ARG, "row band outside the raster");
  // every reflected row the outputs need must be inside the input band
  const int lo = a.out_row0 - a.radius, hi = a.out_row0 + a.out_rows + a.radius;
  const int span = hi - lo;
  if (span >= 2 * a.img_rows) {
    if (a.in_row0 != 0 || a.in_rows != a.img_rows)
      return smrf_fail(SMRF_E_ARG, "band must hold the whole raster when radius >= rows");
  } else {
    for (int y = lo; y < hi; ++y) {
      const int g = smrf_fold(y, a.img_rows);
      if (g < a.in_row0 || g >= a.in_row0 + a.in_rows)
        return smrf_fail(SMRF_E_ARG, "input band [%d,%d) lacks row %d needed by output rows [%d,%d) at radius %d",
                         a.in_row0, a.in_row0 + a.in_rows, g, a.out_row0, a.out_row0 + a.out_rows, a.radius);
    }
  }
  return SMRF_OK;
}

template <typename T>
int disk_filter(DiskArgs<T> a, bool dilate, int impl, hipStream_t stream) {
  if (int rc = check_band(a)) return rc;
  if (a.radius == 0) {
    const long long n = (long long)a.out_rows * a.cols;
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(copy_flag_kernel<T>, dim3(blocks), dim3(256), 0, stream, a);
    SMRF_LAUNCH_CHECK();
    return SMRF_OK;
  }
  if (impl == SMRF_IMPL_AUTO) impl = a.radius <= SMRF_RING_MAX_RADIUS ? SMRF_IMPL_RING : SMRF_IMPL_DIRECT;
  if (impl == SMRF_IMPL_RING) {
    if (a.radius > SMRF_RING_MAX_RADIUS)
      return smrf_fail(SMRF_E_UNSUPPORTED, "ring kernels cover radius <= %d (got %d)", SMRF_RING_MAX_RADIUS, a.radius);
    a.seg = smrf_sw().ring_seg;   // 0: the launcher sizes segments from its occupancy
    return RingFn<T>::call(a, dilate ? SMRF_RING_DILATE : SMRF_RING_ERODE, stream);
  }
  if (impl != SMRF_IMPL_DIRECT) return smrf_fail(SMRF_E_ARG, "unknown impl %d", impl);
  dim3 grid((a.cols + 63) / 64, (a.out_rows + 3) / 4);
  if (dilate) hipLaunchKernelGGL((direct_kernel<T, true>), grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((direct_kernel<T, false>), grid, dim3(256), 0, stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

// output cells as non-temporal stores (DiskArgs::nt): when a plane is far larger than what the caches can hand to the
// next launch (16384^2 fp32, 1 GiB per plane: -1.7 % on the whole progressive_filter; 4096^2, 64 MB: +2 %).  SMRF_NT=0|1 forces it.
template <typename T>
int nt_rule(int img_rows, int cols) {
  const int env = smrf_sw().nt;
  if (env >= 0) return env != 0;
  return (size_t)img_rows * (size_t)cols * sizeof(T) >= ((size_t)192 << 20);
}

template <typename T>
int disk_filter_api(const T* in, T* out, int img_rows, int cols, int64_t ld, int in_row0, int in_rows,
                    int out_row0, int out_rows, int radius, int is_dilate, int nan_aware, int impl, void* stream) {
  DiskArgs<T> a{};
  a.in = in; a.out = out; a.img_rows = img_rows; a.cols = cols; a.ld = ld;
  a.in_row0 = in_row0; a.in_rows = in_rows; a.out_row0 = out_row0; a.out_rows = out_rows;
  a.radius = radius; a.nan_aware = nan_aware; a.nt = nt_rule<T>(img_rows, cols);
  return disk_filter(a, is_dilate != 0, impl, (hipStream_t)stream);
}

template <typename T>
int dilate_flag_api(const T* eroded, const T* last, T* opened, uint8_t* mask, uint8_t* when, double thr,
                    int widx, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                    int out_rows, int radius, int nan_aware, int impl, void* stream, int dense = 0) {
  if (!last || !mask) return smrf_fail(SMRF_E_ARG, "null last/mask pointer");
  DiskArgs<T> a{};
  a.in = eroded; a.out = opened; a.last = last; a.mask = mask; a.when = when; a.thr = thr; a.thr_lo = smrf_float_below(thr); a.widx = widx;
  a.img_rows = img_rows; a.cols = cols; a.ld = ld;
  a.in_row0 = in_row0; a.in_rows = in_rows; a.out_row0 = out_row0; a.out_rows = out_rows;
  a.radius = radius; a.nan_aware = nan_aware; a.nt = nt_rule<T>(img_rows, cols); a.dense = dense;
  return disk_filter(a, true, impl, (hipStream_t)stream);
}

template <typename T>
int count_nan_api(const T* p, int64_t n, int64_t* h_count, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!h_count || (n > 0 && !p)) return smrf_fail(SMRF_E_ARG, "null pointer");
  *h_count = 0;
  if (n <= 0) return SMRF_OK;
  unsigned long long* d = nullptr;
  SMRF_HIP_CHECK(hipMallocAsync((void**)&d, sizeof(*d), stream));
  SMRF_HIP_CHECK(hipMemsetAsync(d, 0, sizeof(*d), stream));
  const int blocks = (int)std::min<long long>((n + 255) / 256, 2048);
  hipLaunchKernelGGL(count_nan_kernel<T>, dim3(blocks), dim3(256), 0, stream, p, (long long)n, d);
  unsigned long long h = 0;
  hipError_t e = hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, stream);
  if (e == hipSuccess) e = hipStreamSynchronize(stream);
  hipFreeAsync(d, stream);
  SMRF_HIP_CHECK(e);
  *h_count = (int64_t)h;
  return SMRF_OK;
}

template <typename T>
int open_flag_api(const T* last, T* opened, uint8_t* mask, uint8_t* when, double thr, int widx, int img_rows, int cols,
                  int64_t ld, int in_row0, int in_rows, int out_row0, int out_rows, int radius, void* stream, int dense = 0) {
  if (!smrf_fused_radius((int)sizeof(T), radius))
    return smrf_fail(SMRF_E_UNSUPPORTED, "no fused opening kernel for radius %d at this dtype", radius);
  DiskArgs<T> a{};
  a.in = last; a.out = opened; a.mask = mask; a.when = when; a.thr = thr; a.thr_lo = smrf_float_below(thr); a.widx = widx;
  a.img_rows = img_rows; a.cols = cols; a.ld = ld;
  a.in_row0 = in_row0; a.in_rows = in_rows; a.out_row0 = out_row0; a.out_rows = out_rows;
  a.radius = 2 * radius;                               // the band check: `last` must reach 2r rows beyond the outputs
  if (int rc = check_band(a)) return rc;
  a.radius = radius;
  a.last = last + (long long)(out_row0 - in_row0) * ld; // the flag step compares against the same surface
  a.nan_aware = 0;
  a.nt = nt_rule<T>(img_rows, cols);
  a.dense = mask ? dense : 0;
  a.seg = smrf_sw().ring_seg;
  return RingFn<T>::call(a, SMRF_RING_FUSED_OPEN, (hipStream_t)stream);
}

// several consecutive small windows on a row band in ONE launch (morph_chain.h), the row-band form of what
// progressive_filter_api does on a whole raster
template <typename T>
int chain_flag_api(const T* last, T* opened, uint8_t* mask, uint8_t* when, const int32_t* radii, const double* thr,
                   const int32_t* widx, int n, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                   int out_rows, void* stream) {
  if (!last || !opened || !mask || !radii || !thr || !widx) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (n < 1 || n > 4) return smrf_fail(SMRF_E_ARG, "a chain has 1..4 windows (got %d)", n);
  const int pat = smrf_chain_match((int)sizeof(T), radii, n, 1ll << 62);
  if (pat < 0 || smrf_chain_length(pat) != n)
    return smrf_fail(SMRF_E_UNSUPPORTED, "no chained launch for these %d radii at this dtype (smrf_pf_chain_length tells)", n);
  DiskArgs<T> b{};                                         // the band check: `last` must reach sum(2r) rows beyond the outputs
  b.in = last; b.out = opened; b.img_rows = img_rows; b.cols = cols; b.ld = ld;
  b.in_row0 = in_row0; b.in_rows = in_rows; b.out_row0 = out_row0; b.out_rows = out_rows;
  b.radius = smrf_chain_halo(pat);
  if (int rc = check_band(b)) return rc;
  if (b.radius >= img_rows) return smrf_fail(SMRF_E_UNSUPPORTED, "raster of %d rows is shorter than the chain's %d halo rows", img_rows, b.radius);
  ChainArgs<T> c{};
  c.in = last; c.out = opened; c.mask = mask; c.when = when;
  for (int k = 0; k < n; ++k) { c.thr[k] = thr[k]; c.widx[k] = widx[k]; }
  c.img_rows = img_rows; c.cols = cols; c.ld = ld;
  c.in_row0 = in_row0; c.in_rows = in_rows; c.out_row0 = out_row0; c.out_rows = out_rows;
  c.seg = smrf_sw().ring_seg;
  c.nt = nt_rule<T>(img_rows, cols);
  c.dense0 = 0;
  if constexpr (sizeof(T) == 4) return smrf_chain_f32(pat, c, (hipStream_t)stream);
  else return smrf_chain_f64(pat, c, (hipStream_t)stream);
}

template <typename T>
int progressive_filter_api(const T* Z, int rows, int cols, const int32_t* windows, const double* thr, int nwin,
                           uint8_t* mask, uint8_t* when, void* ws, size_t ws_bytes, int nan_aware, int impl,
                           void* stream_, float* h_window_ms = nullptr, int32_t* h_window_route = nullptr) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!Z || !mask || (nwin > 0 && (!windows || !thr))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 1 || cols < 1 || nwin < 0) return smrf_fail(SMRF_E_ARG, "bad size");
  const size_t need = smrf_progressive_filter_workspace_bytes(rows, cols, (int)sizeof(T));
  if (ws_bytes < need || !ws) return smrf_fail(SMRF_E_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, need);
  for (int i = 0; i < nwin; ++i)
    if (windows[i] < 0) return smrf_fail(SMRF_E_ARG, "negative window %d", windows[i]);
  const size_t plane = (size_t)rows * cols;
  T* E = reinterpret_cast<T*>(ws);
  T* O[2] = {E + plane, E + 2 * plane};
  // The first window's flag step writes every mask / when byte (DiskArgs::dense), so the planes are not cleared first:
  // one coalesced byte per cell instead of a memset pass plus scattered single-byte stores (window 0 flags the most cells)
  if (nwin == 0) {
    SMRF_HIP_CHECK(hipMemsetAsync(mask, 0, plane, stream));
    if (when) SMRF_HIP_CHECK(hipMemsetAsync(when, 0, plane, stream));
  }
  // nan_aware < 0: the library finds out itself.  When the call starts with a chained / table-free launch, that launch
  // carries the scan (it loads every cell of the raster; ChainArgs::nan_flag): it runs as if there were no NaN, the
  // flag is read back after it (the one synchronising readback a separate count would need as well), and in the rare
  // case it is set the call starts over on the NaN-aware two-pass kernels.  Otherwise: one count pass first.
  const int fuse_mode0 = smrf_sw().fused;
  const bool may_chain = nan_aware < 0 && nwin > 0 && fuse_mode0 != 0 && smrf_sw().chain != 0 &&
                         smrf_sw().nan_ride != 0 &&   // 0: always a separate count pass (A/B runs)
                         (impl == SMRF_IMPL_AUTO || impl == SMRF_IMPL_RING);
  const int pat0 = may_chain ? smrf_chain_match((int)sizeof(T), windows, nwin, fuse_mode0 == 2 ? (1ll << 62) : (long long)plane) : -1;
  // the flag of a speculative first launch: the first word of the workspace's E plane, which nothing touches before the
  // first two-pass window's erosion (chained / fused launches write O[0], O[1] only) - no allocation on this path
  unsigned* d_nan = nullptr;
  if (nan_aware < 0) {
    if (pat0 >= 0 && smrf_chain_halo(pat0) < rows) {
      d_nan = reinterpret_cast<unsigned*>(E);
      SMRF_HIP_CHECK(hipMemsetAsync(d_nan, 0, sizeof(unsigned), stream));
      nan_aware = 0;
    } else {
      int64_t c = 0;
      if (int rc = count_nan_api<T>(Z, (int64_t)plane, &c, stream_)) return rc;
      nan_aware = c > 0;
    }
  }
  // measurement form (smrf_progressive_filter_timed_*): an event on the stream at every window boundary
  struct Events {                                          // destroyed on every way out
    std::vector<hipEvent_t> v;
    ~Events() { for (hipEvent_t e : v) if (e) hipEventDestroy(e); }
  } evs;
  std::vector<hipEvent_t>& ev = evs.v;
  if (h_window_ms) {
    ev.assign((size_t)nwin + 1, nullptr);
    for (auto& e : ev) SMRF_HIP_CHECK(hipEventCreate(&e));
    SMRF_HIP_CHECK(hipEventRecord(ev[0], stream));
  }
  auto window_done = [&](int i, int route) -> int {
    if (h_window_route) h_window_route[i] = route;
    if (h_window_ms) SMRF_HIP_CHECK(hipEventRecord(ev[(size_t)i + 1], stream));
    return SMRF_OK;
  };
  auto finish = [&]() -> int {
    if (!h_window_ms) return SMRF_OK;
    hipError_t e = nwin > 0 ? hipEventSynchronize(ev[(size_t)nwin]) : hipSuccess;
    for (int i = 0; i < nwin && e == hipSuccess; ++i) e = hipEventElapsedTime(&h_window_ms[i], ev[(size_t)i], ev[(size_t)i + 1]);
    SMRF_HIP_CHECK(e);
    return SMRF_OK;
  };
  const T* last = Z;
  // small disks: opening + flag in ONE launch, the eroded surface never leaves the CU (morph_fused.h; 10 instead of
  // 22 B/cell in fp32), and runs of consecutive small windows (chain.hip's patterns: 1, 2, 3 | 1, 2 | 2, 3 | 4, 5, and the
  // table-free single launches 4..10) as ONE launch that only reads the first window's input and writes the last window's
  // opening (morph_chain.h).  Not for rasters with NaNs (scipy's NaN rule lives in the two-pass kernels only) - except that
  // the FIRST launch of a call with nan_aware < 0 runs speculatively and carries the NaN scan: if it meets a NaN its
  // results are discarded and the call starts over on the two-pass kernels (below).
  // SMRF_FUSED: 0 = never, 1 = default rule, 2 = every radius that has a fused kernel whatever the raster size (tests);
  // SMRF_CHAIN: 0 = no chains (every window its own launch).
  const int fuse_mode = smrf_sw().fused;
  const bool fuse_ok0 = (impl == SMRF_IMPL_AUTO || impl == SMRF_IMPL_RING) && fuse_mode != 0;
  const bool chain_ok0 = smrf_sw().chain != 0;
  int flip = 0;                                          // which of the two opened planes the next launch writes
  for (int i = 0; i < nwin;) {
    const int r = windows[i];
    T* opened = O[flip];
    flip ^= 1;
    const bool fuse_ok = !nan_aware && fuse_ok0;           // (nan_aware can change once: a speculative first launch that met a NaN)
    const bool chain_ok = fuse_ok && chain_ok0;
    const int pat = chain_ok ? smrf_chain_match((int)sizeof(T), windows + i, nwin - i, fuse_mode == 2 ? (1ll << 62) : (long long)plane) : -1;
    // a speculative first launch was decided on pat0 above: the loop must take exactly that launch, or a NaN raster would
    // run the NaN-free kernels unscanned
    if (i == 0 && d_nan && pat != pat0) return smrf_fail(SMRF_E_HIP, "internal: first launch %d is not the one the NaN scan rides in (%d)", pat, pat0);
    if (pat >= 0 && smrf_chain_halo(pat) < rows) {
      const int len = smrf_chain_length(pat);
      ChainArgs<T> c{};
      c.in = last; c.out = opened; c.mask = mask; c.when = when;
      for (int k = 0; k < len; ++k) { c.thr[k] = thr[i + k]; c.widx[k] = i + k; }
      c.img_rows = rows; c.cols = cols; c.ld = cols;
      c.in_row0 = 0; c.in_rows = rows; c.out_row0 = 0; c.out_rows = rows;
      c.seg = smrf_sw().ring_seg;
      c.nt = nt_rule<T>(rows, cols);
      c.dense0 = i == 0;
      c.nan_flag = i == 0 ? d_nan : nullptr;
      int crc;
      if constexpr (sizeof(T) == 4) crc = smrf_chain_f32(pat, c, stream);
      else crc = smrf_chain_f64(pat, c, stream);
      if (crc) return crc;
      if (i == 0 && d_nan) {                               // the speculative first launch: did it meet a NaN?
        unsigned h = 0;
        SMRF_HIP_CHECK(hipMemcpyAsync(&h, d_nan, sizeof(h), hipMemcpyDeviceToHost, stream));
        SMRF_HIP_CHECK(hipStreamSynchronize(stream));
        d_nan = nullptr;
        if (h) {                                           // start over with scipy's NaN rule (two-pass kernels only)
          nan_aware = 1;
          flip = 0;
          last = Z;
          continue;                                        // i is still 0
        }
      }
      if (nwin > 1) last = opened;
      for (int k = 0; k < len; ++k)                        // the chain's time lands on its first window, the others read ~0
        if (int rc = window_done(i + k, SMRF_ROUTE_CHAIN + k)) return rc;
      i += len;
      continue;
    }
    // above R = 8 the fused kernel's 4R warm-up rows per segment only pay on rasters large enough for long segments
    // (4096^2, windows 1..18: 1.64 ms with R <= 8 fused, 1.70 ms with 10..14 as well; 8192^2: 5.9 -> 5.2 ms with them);
    // round 5, after the launches' segmentation changed: from 20 Mi cells (5000^2: R = 11..13 -7 ... -11 %, 6000^2 -10 ... -17 %,
    // R = 14 equal; 4096^2 and below +3 ... +20 %: profiles/r05_logs/segments/min_cells_fused.log); 48 Mi until then
    if (fuse_ok && smrf_fused_radius((int)sizeof(T), r) && (r <= 8 || fuse_mode == 2 || plane >= ((size_t)20 << 20))) {
      if (int rc = open_flag_api<T>(last, opened, mask, when, thr[i], i, rows, cols, cols, 0, rows, 0, rows, r, stream_, i == 0)) return rc;
      if (nwin > 1) last = opened;
      if (int rc = window_done(i, SMRF_ROUTE_FUSED)) return rc;
      ++i;
      continue;
    }
    if (int rc = disk_filter_api<T>(last, E, rows, cols, cols, 0, rows, 0, rows, r, 0, nan_aware, impl, stream_)) return rc;
    if (int rc = dilate_flag_api<T>(E, last, opened, mask, when, thr[i], i, rows, cols, cols, 0, rows, 0, rows, r,
                                    nan_aware, impl, stream_, i == 0))
      return rc;
    if (nwin > 1) last = opened;                        // neilpy.py:1675-1676
    const int eff = impl == SMRF_IMPL_AUTO ? (r <= SMRF_RING_MAX_RADIUS ? SMRF_IMPL_RING : SMRF_IMPL_DIRECT) : impl;
    if (int rc = window_done(i, r == 0 ? SMRF_ROUTE_COPY : eff == SMRF_IMPL_RING ? SMRF_ROUTE_TWO_PASS : SMRF_ROUTE_DIRECT)) return rc;
    ++i;
  }
  return finish();
}

}  // namespace

extern "C" {

int smrf_disk_filter_f32(const float* d_in, float* d_out, int img_rows, int cols, int64_t ld, int in_row0,
                         int in_rows, int out_row0, int out_rows, int radius, int is_dilate, int nan_aware,
                         int impl, void* stream) {
  return disk_filter_api<float>(d_in, d_out, img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, radius,
                                is_dilate, nan_aware, impl, stream);
}
int smrf_disk_filter_f64(const double* d_in, double* d_out, int img_rows, int cols, int64_t ld, int in_row0,
                         int in_rows, int out_row0, int out_rows, int radius, int is_dilate, int nan_aware,
                         int impl, void* stream) {
  return disk_filter_api<double>(d_in, d_out, img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, radius,
                                 is_dilate, nan_aware, impl, stream);
}
int smrf_pf_dilate_flag_f32(const float* d_eroded, const float* d_last, float* d_opened, uint8_t* d_mask,
                            uint8_t* d_when_dropped, double threshold, int window_index, int img_rows, int cols,
                            int64_t ld, int in_row0, int in_rows, int out_row0, int out_rows, int radius,
                            int nan_aware, int impl, void* stream) {
  return dilate_flag_api<float>(d_eroded, d_last, d_opened, d_mask, d_when_dropped, threshold, window_index,
                                img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, radius, nan_aware, impl,
                                stream);
}
int smrf_pf_dilate_flag_f64(const double* d_eroded, const double* d_last, double* d_opened, uint8_t* d_mask,
                            uint8_t* d_when_dropped, double threshold, int window_index, int img_rows, int cols,
                            int64_t ld, int in_row0, int in_rows, int out_row0, int out_rows, int radius,
                            int nan_aware, int impl, void* stream) {
  return dilate_flag_api<double>(d_eroded, d_last, d_opened, d_mask, d_when_dropped, threshold, window_index,
                                 img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, radius, nan_aware, impl,
                                 stream);
}
int smrf_fused_open_supported(int elem_size, int radius) { return smrf_fused_radius(elem_size, radius) ? 1 : 0; }
int smrf_pf_open_flag_f32(const float* d_last, float* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped, double threshold,
                          int window_index, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                          int out_rows, int radius, void* stream) {
  if (!d_last || !d_opened) return smrf_fail(SMRF_E_ARG, "null raster pointer");
  return open_flag_api<float>(d_last, d_opened, d_mask, d_when_dropped, threshold, window_index, img_rows, cols, ld, in_row0,
                              in_rows, out_row0, out_rows, radius, stream);
}
int smrf_pf_open_flag_f64(const double* d_last, double* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped, double threshold,
                          int window_index, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                          int out_rows, int radius, void* stream) {
  if (!d_last || !d_opened) return smrf_fail(SMRF_E_ARG, "null raster pointer");
  return open_flag_api<double>(d_last, d_opened, d_mask, d_when_dropped, threshold, window_index, img_rows, cols, ld, in_row0,
                               in_rows, out_row0, out_rows, radius, stream);
}
int smrf_pf_chain_length(int elem_size, const int32_t* h_radii, int n, int64_t raster_cells) {
  if (!h_radii || n < 1) return 0;
  // the same switches smrf_progressive_filter_* routes by (read once at load, smrf_switches_reload): SMRF_CHAIN=0 or
  // SMRF_FUSED=0 = no chained / table-free launches, SMRF_FUSED=2 = every pattern whatever the raster size
  const SmrfSwitches& sw = smrf_sw();
  if (sw.chain == 0 || sw.fused == 0) return 0;
  const int pat = smrf_chain_match(elem_size, h_radii, n, sw.fused == 2 ? (1ll << 62) : (long long)raster_cells);
  return pat < 0 ? 0 : smrf_chain_length(pat);
}
int smrf_pf_chain_flag_f32(const float* d_last, float* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                           const int32_t* h_radii, const double* h_thresholds, const int32_t* h_window_index, int n_windows,
                           int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0, int out_rows,
                           void* stream) {
  return chain_flag_api<float>(d_last, d_opened, d_mask, d_when_dropped, h_radii, h_thresholds, h_window_index, n_windows,
                               img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, stream);
}
int smrf_pf_chain_flag_f64(const double* d_last, double* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                           const int32_t* h_radii, const double* h_thresholds, const int32_t* h_window_index, int n_windows,
                           int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0, int out_rows,
                           void* stream) {
  return chain_flag_api<double>(d_last, d_opened, d_mask, d_when_dropped, h_radii, h_thresholds, h_window_index, n_windows,
                                img_rows, cols, ld, in_row0, in_rows, out_row0, out_rows, stream);
}
size_t smrf_progressive_filter_workspace_bytes(int rows, int cols, int elem_size) {
  return (size_t)3 * (size_t)rows * (size_t)cols * (size_t)elem_size;
}
int smrf_progressive_filter_f32(const float* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask, uint8_t* d_when_dropped,
                                void* d_workspace, size_t workspace_bytes, int nan_aware, int impl, void* stream) {
  return progressive_filter_api<float>(d_Z, rows, cols, h_windows, h_thresholds, n_windows, d_mask, d_when_dropped,
                                       d_workspace, workspace_bytes, nan_aware, impl, stream);
}
int smrf_progressive_filter_f64(const double* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask, uint8_t* d_when_dropped,
                                void* d_workspace, size_t workspace_bytes, int nan_aware, int impl, void* stream) {
  return progressive_filter_api<double>(d_Z, rows, cols, h_windows, h_thresholds, n_windows, d_mask, d_when_dropped,
                                        d_workspace, workspace_bytes, nan_aware, impl, stream);
}
int smrf_progressive_filter_timed_f32(const float* d_Z, int rows, int cols, const int32_t* h_windows,
                                      const double* h_thresholds, int n_windows, uint8_t* d_mask, uint8_t* d_when_dropped,
                                      void* d_workspace, size_t workspace_bytes, int nan_aware, int impl, void* stream,
                                      float* h_window_ms, int32_t* h_window_route) {
  if (!h_window_ms) return smrf_fail(SMRF_E_ARG, "null h_window_ms");
  return progressive_filter_api<float>(d_Z, rows, cols, h_windows, h_thresholds, n_windows, d_mask, d_when_dropped,
                                       d_workspace, workspace_bytes, nan_aware, impl, stream, h_window_ms, h_window_route);
}
int smrf_progressive_filter_timed_f64(const double* d_Z, int rows, int cols, const int32_t* h_windows,
                                      const double* h_thresholds, int n_windows, uint8_t* d_mask, uint8_t* d_when_dropped,
                                      void* d_workspace, size_t workspace_bytes, int nan_aware, int impl, void* stream,
                                      float* h_window_ms, int32_t* h_window_route) {
  if (!h_window_ms) return smrf_fail(SMRF_E_ARG, "null h_window_ms");
  return progressive_filter_api<double>(d_Z, rows, cols, h_windows, h_thresholds, n_windows, d_mask, d_when_dropped,
                                        d_workspace, workspace_bytes, nan_aware, impl, stream, h_window_ms, h_window_route);
}
int smrf_count_nan_f32(const float* d_a, int64_t n, int64_t* h_count, void* stream) {
  return count_nan_api<float>(d_a, n, h_count, stream);
}
int smrf_count_nan_f64(const double* d_a, int64_t n, int64_t* h_count, void* stream) {
  return count_nan_api<double>(d_a, n, h_count, stream);
}

}  // extern "C"
