// create_dem on the device (neilpy.py:1110-1166): extent reduction, point -> cell binning with
// 64-bit atomic min on an order-preserving key, key -> float64 grid.
//
// The reference bins with a pandas groupby min/max over the flat cell index (:1151-1156).  Min
// and max are order independent, so a scatter with atomics gives the identical grid.  The key is
// the float64 bit pattern made monotone (sign flip), so unsigned 64-bit atomicMin implements an
// exact floating-point min (and, on the complemented key, max) including +-inf; all-ones marks an
// empty cell, which no non-NaN value maps to.  HBM-bound: 24 B read per point + one 8 B atomic.
#include <algorithm>

#include "smrf_common.h"

namespace {

__device__ __forceinline__ unsigned long long f64_key(double z) {
  unsigned long long k = (unsigned long long)__double_as_longlong(z);
  return (k >> 63) ? ~k : (k | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k) {
  k = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  return __longlong_as_double((long long)k);
}

__global__ __launch_bounds__(256) void extent_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                     long long n, double* __restrict__ part) {
  double xmin = INFINITY, xmax = -INFINITY, ymin = INFINITY, ymax = -INFINITY;
  bool bad = false;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double a = x[i], b = y[i];
    bad |= (a != a) | (b != b);
    xmin = fmin(xmin, a); xmax = fmax(xmax, a);
    ymin = fmin(ymin, b); ymax = fmax(ymax, b);
  }
  for (int o = 32; o > 0; o >>= 1) {
    xmin = fmin(xmin, __shfl_down(xmin, o, 64)); xmax = fmax(xmax, __shfl_down(xmax, o, 64));
    ymin = fmin(ymin, __shfl_down(ymin, o, 64)); ymax = fmax(ymax, __shfl_down(ymax, o, 64));
    bad |= (bool)__shfl_down((int)bad, o, 64);
  }
  __shared__ double s[4][4];
  __shared__ int sbad[4];
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { s[0][w] = xmin; s[1][w] = xmax; s[2][w] = ymin; s[3][w] = ymax; sbad[w] = bad; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const bool b = sbad[0] | sbad[1] | sbad[2] | sbad[3];
    // np.min / np.max propagate NaN (neilpy.py:1121-1124 would then build a NaN-sized grid)
    part[blockIdx.x * 4 + 0] = b ? NAN : fmin(fmin(s[0][0], s[0][1]), fmin(s[0][2], s[0][3]));
    part[blockIdx.x * 4 + 1] = b ? NAN : fmax(fmax(s[1][0], s[1][1]), fmax(s[1][2], s[1][3]));
    part[blockIdx.x * 4 + 2] = b ? NAN : fmin(fmin(s[2][0], s[2][1]), fmin(s[2][2], s[2][3]));
    part[blockIdx.x * 4 + 3] = b ? NAN : fmax(fmax(s[3][0], s[3][1]), fmax(s[3][2], s[3][3]));
  }
}

__global__ __launch_bounds__(256) void fill_u64_kernel(unsigned long long* __restrict__ p, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) p[i] = ~0ull;
}

struct BinArgs {
  const double *x, *y, *z;
  long long n;
  double ia, ib, ic, id, ie, jf;   // inverse affine (a, b, c, d, e, f)
  double fx0, fx1, fy0, fy1;       // filter: xedges[0], xedges[-1], yedges[-1], yedges[0]
  int use_filter;
  unsigned long long* keys;
  int rows_total, cols, row0, rows_local, is_max;
  unsigned long long* n_outside;
};

__global__ __launch_bounds__(256) void bin_kernel(const BinArgs a) {
  unsigned long long outside = 0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    const double px = a.x[i], py = a.y[i], pz = a.z[i];
    if (a.use_filter && ((px < a.fx0) | (px > a.fx1) | (py > a.fy1) | (py < a.fy0))) continue;   // :1128
    // ~t * (x, y) of the affine package: vx*sa + vy*sb + sc, each product and sum rounded (no FMA)
    const double fc = __dadd_rn(__dadd_rn(__dmul_rn(px, a.ia), __dmul_rn(py, a.ib)), a.ic);
    const double fr = __dadd_rn(__dadd_rn(__dmul_rn(px, a.id), __dmul_rn(py, a.ie)), a.jf);
    const double c = floor(fc), r = floor(fr);
    if (!(c >= 0.0 && c < (double)a.cols && r >= 0.0 && r < (double)a.rows_total)) { ++outside; continue; }
    if (pz != pz) continue;                                           // groupby min/max skip NaN
    const int ri = (int)r - a.row0;
    if (ri < 0 || ri >= a.rows_local) continue;                       // another rank's band
    unsigned long long k = f64_key(pz);
    if (a.is_max) k = ~k;
    atomicMin(a.keys + (long long)ri * a.cols + (int)c, k);
  }
  for (int o = 32; o > 0; o >>= 1) outside += __shfl_down(outside, o, 64);
  if ((threadIdx.x & 63) == 0 && outside) atomicAdd(a.n_outside, outside);
}

__global__ __launch_bounds__(256) void finalize_kernel(const unsigned long long* __restrict__ keys,
                                                       double* __restrict__ grid, uint8_t* __restrict__ empty,
                                                       long long n, int is_max) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    unsigned long long k = keys[i];
    const bool e = (k == ~0ull);
    if (is_max) k = ~k;
    grid[i] = e ? (double)NAN : key_f64(k);
    if (empty) empty[i] = e;
  }
}

// (col, row) = ~t * (x, y) as fractional pixel coordinates (neilpy.py:1772), no FMA contraction
__global__ __launch_bounds__(256) void affine_kernel(const double* __restrict__ x, const double* __restrict__ y,
                                                     long long n, double ia, double ib, double ic, double id, double ie,
                                                     double jf, double* __restrict__ col, double* __restrict__ row) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double px = x[i], py = y[i];
    col[i] = __dadd_rn(__dadd_rn(__dmul_rn(px, ia), __dmul_rn(py, ib)), ic);
    row[i] = __dadd_rn(__dadd_rn(__dmul_rn(px, id), __dmul_rn(py, ie)), jf);
  }
}

// LAS point records -> float64 coordinates: x = int32 * scale + offset, product and sum rounded
// separately (the reference's pandas arithmetic, neilpy.py:1055-1057).  Records are packed and
// unaligned (20..67+ bytes), so the three int32 are assembled from bytes.
__global__ __launch_bounds__(256) void las_decode_kernel(const uint8_t* __restrict__ rec, long long n, int reclen,
                                                         double sx, double sy, double sz, double ox, double oy, double oz,
                                                         double* __restrict__ x, double* __restrict__ y,
                                                         double* __restrict__ z) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const uint8_t* p = rec + i * reclen;
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
      v[k] = (int)((unsigned)p[4 * k] | ((unsigned)p[4 * k + 1] << 8) | ((unsigned)p[4 * k + 2] << 16) |
                   ((unsigned)p[4 * k + 3] << 24));
    x[i] = __dadd_rn(__dmul_rn((double)v[0], sx), ox);
    y[i] = __dadd_rn(__dmul_rn((double)v[1], sy), oy);
    z[i] = __dadd_rn(__dmul_rn((double)v[2], sz), oz);
  }
}

// ---- points -> destination row band (the all-to-all form of the sharded create_dem) -----------------
// band of a raster row under sharded.band_rows(): the first `rem` bands hold base + 1 rows, the others base
__device__ __forceinline__ int band_of_row(int r, int base, int rem) {
  const int split = rem * (base + 1);
  return r < split ? r / (base + 1) : rem + (r - split) / max(base, 1);
}
struct BucketArgs {
  const double *x, *y, *z;
  long long n;
  double id, ie, jf;              // row = x * id + y * ie + jf (the row half of the inverse affine)
  int rows_total, nbands, base, rem;
};
__device__ __forceinline__ int dest_band(const BucketArgs& a, long long i) {
  // the same rounding as bin_kernel: a point goes to the rank whose rows it will be binned into.  Rows outside the
  // raster (or a NaN coordinate) clamp to the nearest band, whose bin_kernel then counts the point as outside.
  const double fr = floor(__dadd_rn(__dadd_rn(__dmul_rn(a.x[i], a.id), __dmul_rn(a.y[i], a.ie)), a.jf));
  const int r = !(fr >= 0.0) ? 0 : (fr >= (double)a.rows_total ? a.rows_total - 1 : (int)fr);
  return min(band_of_row(r, a.base, a.rem), a.nbands - 1);
}
constexpr int MAXBANDS = 64;

__global__ __launch_bounds__(256) void band_count_kernel(const BucketArgs a, unsigned long long* __restrict__ counts) {
  __shared__ unsigned int h[MAXBANDS];
  if (threadIdx.x < MAXBANDS) h[threadIdx.x] = 0;
  __syncthreads();
  for (long long i0 = blockIdx.x * 256ll * 16; i0 < a.n; i0 += (long long)gridDim.x * 256 * 16)
    for (int k = 0; k < 16; ++k) {
      const long long i = i0 + k * 256 + threadIdx.x;
      if (i < a.n) atomicAdd(&h[dest_band(a, i)], 1u);
    }
  __syncthreads();
  if (threadIdx.x < a.nbands && h[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)h[threadIdx.x]);
}

// scatter into per-band runs of the packed SoA buffers: a block reserves its share of every band's run with one
// atomic per band, lanes take their slot inside the share from an LDS counter (order within a run is arbitrary:
// the min/max binning downstream is order independent)
__global__ __launch_bounds__(256) void band_pack_kernel(const BucketArgs a, unsigned long long* __restrict__ cursors,
                                                        double* __restrict__ ox, double* __restrict__ oy,
                                                        double* __restrict__ oz) {
  __shared__ unsigned int h[MAXBANDS];
  __shared__ unsigned long long basepos[MAXBANDS];
  for (long long i0 = blockIdx.x * 256ll * 16; i0 < a.n; i0 += (long long)gridDim.x * 256 * 16) {
    if (threadIdx.x < MAXBANDS) h[threadIdx.x] = 0;
    __syncthreads();
    int dest[16];
    unsigned int slot[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const long long i = i0 + k * 256 + threadIdx.x;
      dest[k] = -1;
      if (i < a.n) { dest[k] = dest_band(a, i); slot[k] = atomicAdd(&h[dest[k]], 1u); }
    }
    __syncthreads();
    if (threadIdx.x < a.nbands && h[threadIdx.x])
      basepos[threadIdx.x] = atomicAdd(&cursors[threadIdx.x], (unsigned long long)h[threadIdx.x]);
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (dest[k] < 0) continue;
      const long long i = i0 + k * 256 + threadIdx.x;
      const unsigned long long o = basepos[dest[k]] + slot[k];
      ox[o] = a.x[i]; oy[o] = a.y[i]; oz[o] = a.z[i];
    }
    __syncthreads();
  }
}

int nblocks(long long n, int cap) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, cap)); }

}  // namespace

extern "C" {

int smrf_points_extent_f64(const double* d_x, const double* d_y, int64_t n, double* h_out, void* d_workspace,
                           size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_x || !d_y || !h_out || n < 1) return smrf_fail(SMRF_E_ARG, "extent needs at least one point");
  const int blocks = nblocks(n, 1024);
  if (!d_workspace || workspace_bytes < (size_t)blocks * 4 * sizeof(double))
    return smrf_fail(SMRF_E_WORKSPACE, "extent workspace too small");
  double* part = (double*)d_workspace;
  hipLaunchKernelGGL(extent_kernel, dim3(blocks), dim3(256), 0, stream, d_x, d_y, (long long)n, part);
  SMRF_LAUNCH_CHECK();
  static thread_local double host[4 * 1024];
  SMRF_HIP_CHECK(hipMemcpyAsync(host, part, (size_t)blocks * 4 * sizeof(double), hipMemcpyDeviceToHost, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  double r[4] = {INFINITY, -INFINITY, INFINITY, -INFINITY};
  bool bad = false;
  for (int b = 0; b < blocks; ++b) {
    for (int k = 0; k < 4; ++k) bad |= (host[b * 4 + k] != host[b * 4 + k]);
    r[0] = std::min(r[0], host[b * 4 + 0]); r[1] = std::max(r[1], host[b * 4 + 1]);
    r[2] = std::min(r[2], host[b * 4 + 2]); r[3] = std::max(r[3], host[b * 4 + 3]);
  }
  for (int k = 0; k < 4; ++k) h_out[k] = bad ? NAN : r[k];
  return SMRF_OK;
}

int smrf_affine_apply_f64(const double* d_x, const double* d_y, int64_t npts, const double* h_inv, double* d_col,
                          double* d_row, void* stream) {
  if (npts < 0 || !h_inv || (npts > 0 && (!d_x || !d_y || !d_col || !d_row))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (npts == 0) return SMRF_OK;
  hipLaunchKernelGGL(affine_kernel, dim3(nblocks(npts, 8192)), dim3(256), 0, (hipStream_t)stream, d_x, d_y,
                     (long long)npts, h_inv[0], h_inv[1], h_inv[2], h_inv[3], h_inv[4], h_inv[5], d_col, d_row);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_las_decode_xyz_f64(const uint8_t* d_records, int64_t npts, int record_length, const double* h_scale_offset,
                            double* d_x, double* d_y, double* d_z, void* stream) {
  if (npts < 0 || record_length < 12 || !h_scale_offset || (npts > 0 && (!d_records || !d_x || !d_y || !d_z)))
    return smrf_fail(SMRF_E_ARG, "bad LAS decode arguments");
  if (npts == 0) return SMRF_OK;
  hipLaunchKernelGGL(las_decode_kernel, dim3(nblocks(npts, 8192)), dim3(256), 0, (hipStream_t)stream, d_records,
                     (long long)npts, record_length, h_scale_offset[0], h_scale_offset[1], h_scale_offset[2],
                     h_scale_offset[3], h_scale_offset[4], h_scale_offset[5], d_x, d_y, d_z);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

static int bucket_args(BucketArgs& a, const double* d_x, const double* d_y, const double* d_z, int64_t npts,
                       const double* h_inv, int rows_total, int nbands) {
  if (npts < 0 || !h_inv || (npts > 0 && (!d_x || !d_y))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows_total < 1 || nbands < 1 || nbands > MAXBANDS || nbands > rows_total)
    return smrf_fail(SMRF_E_ARG, "bad band split: %d rows over %d bands (at most %d bands)", rows_total, nbands, MAXBANDS);
  a.x = d_x; a.y = d_y; a.z = d_z; a.n = npts;
  a.id = h_inv[3]; a.ie = h_inv[4]; a.jf = h_inv[5];
  a.rows_total = rows_total; a.nbands = nbands; a.base = rows_total / nbands; a.rem = rows_total % nbands;
  return SMRF_OK;
}

int smrf_points_band_count_f64(const double* d_x, const double* d_y, int64_t npts, const double* h_inv, int rows_total,
                               int nbands, uint64_t* d_counts, void* stream) {
  BucketArgs a;
  if (int rc = bucket_args(a, d_x, d_y, nullptr, npts, h_inv, rows_total, nbands)) return rc;
  if (!d_counts) return smrf_fail(SMRF_E_ARG, "null pointer");
  SMRF_HIP_CHECK(hipMemsetAsync(d_counts, 0, (size_t)nbands * sizeof(uint64_t), (hipStream_t)stream));
  if (npts == 0) return SMRF_OK;
  hipLaunchKernelGGL(band_count_kernel, dim3(nblocks((npts + 15) / 16, 4096)), dim3(256), 0, (hipStream_t)stream, a,
                     (unsigned long long*)d_counts);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_points_band_pack_f64(const double* d_x, const double* d_y, const double* d_z, int64_t npts, const double* h_inv,
                              int rows_total, int nbands, uint64_t* d_cursors, double* d_out_x, double* d_out_y,
                              double* d_out_z, void* stream) {
  BucketArgs a;
  if (int rc = bucket_args(a, d_x, d_y, d_z, npts, h_inv, rows_total, nbands)) return rc;
  if (npts == 0) return SMRF_OK;
  if (!d_z || !d_cursors || !d_out_x || !d_out_y || !d_out_z) return smrf_fail(SMRF_E_ARG, "null pointer");
  hipLaunchKernelGGL(band_pack_kernel, dim3(nblocks((npts + 15) / 16, 4096)), dim3(256), 0, (hipStream_t)stream, a,
                     (unsigned long long*)d_cursors, d_out_x, d_out_y, d_out_z);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_grid_clear_u64(uint64_t* d_keys, int64_t ncells, void* stream) {
  if (!d_keys || ncells < 0) return smrf_fail(SMRF_E_ARG, "bad grid");
  if (ncells == 0) return SMRF_OK;
  hipLaunchKernelGGL(fill_u64_kernel, dim3(nblocks(ncells, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long*)d_keys, (long long)ncells);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_grid_bin_f64(const double* d_x, const double* d_y, const double* d_z, int64_t npts, const double* h_inv,
                      const double* h_filter, uint64_t* d_keys, int rows_total, int cols, int row0, int rows_local,
                      int is_max, int64_t* d_n_outside, void* stream) {
  if (!d_keys || !h_inv || !d_n_outside || (npts > 0 && (!d_x || !d_y || !d_z)))
    return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows_total < 1 || cols < 1 || row0 < 0 || rows_local < 0 || row0 + rows_local > rows_total || npts < 0)
    return smrf_fail(SMRF_E_ARG, "bad grid shape");
  if (npts == 0) return SMRF_OK;
  BinArgs a;
  a.x = d_x; a.y = d_y; a.z = d_z; a.n = npts;
  a.ia = h_inv[0]; a.ib = h_inv[1]; a.ic = h_inv[2]; a.id = h_inv[3]; a.ie = h_inv[4]; a.jf = h_inv[5];
  a.use_filter = h_filter != nullptr;
  a.fx0 = a.fx1 = a.fy0 = a.fy1 = 0.0;
  if (h_filter) { a.fx0 = h_filter[0]; a.fx1 = h_filter[1]; a.fy0 = h_filter[2]; a.fy1 = h_filter[3]; }
  a.keys = (unsigned long long*)d_keys;
  a.rows_total = rows_total; a.cols = cols; a.row0 = row0; a.rows_local = rows_local; a.is_max = is_max != 0;
  a.n_outside = (unsigned long long*)d_n_outside;
  hipLaunchKernelGGL(bin_kernel, dim3(nblocks(npts, 8192)), dim3(256), 0, (hipStream_t)stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_grid_finalize_f64(const uint64_t* d_keys, double* d_grid, uint8_t* d_empty, int64_t ncells, int is_max,
                           void* stream) {
  if (!d_keys || !d_grid || ncells < 0) return smrf_fail(SMRF_E_ARG, "bad grid");
  if (ncells == 0) return SMRF_OK;
  hipLaunchKernelGGL(finalize_kernel, dim3(nblocks(ncells, 8192)), dim3(256), 0, (hipStream_t)stream,
                     (const unsigned long long*)d_keys, d_grid, d_empty, (long long)ncells, is_max != 0);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // extern "C"
