// smrf tail on the device: slope raster (neilpy.py:1785-1786).
#include <algorithm>

#include "smrf_common.h"

namespace {

// np.gradient(Z, h): interior (f[i+1]-f[i-1])/(2h), edges one-sided first order; S = sqrt(gy^2+gx^2)
__global__ __launch_bounds__(256) void slope_kernel(const double* __restrict__ Z, double* __restrict__ S, int rows,
                                                    int cols, double h) {
  const long long n = (long long)rows * cols;
  const double h2 = 2.0 * h;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    double gy, gx;
    if (r == 0) gy = (Z[i + cols] - Z[i]) / h;
    else if (r == rows - 1) gy = (Z[i] - Z[i - cols]) / h;
    else gy = (Z[i + cols] - Z[i - cols]) / h2;
    if (c == 0) gx = (Z[i + 1] - Z[i]) / h;
    else if (c == cols - 1) gx = (Z[i] - Z[i - 1]) / h;
    else gx = (Z[i + 1] - Z[i - 1]) / h2;
    S[i] = sqrt(gy * gy + gx * gx);
  }
}

__global__ __launch_bounds__(256) void negate_kernel(const double* __restrict__ in, double* __restrict__ out,
                                                     long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) out[i] = -in[i];
}

// u = a | b | c ; Z[u] = NaN   (neilpy.py:1762-1763 and :1748)
__global__ __launch_bounds__(256) void mask_apply_kernel(double* __restrict__ Z, const uint8_t* __restrict__ a,
                                                         const uint8_t* __restrict__ b,
                                                         const uint8_t* __restrict__ c, uint8_t* __restrict__ u,
                                                         long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const bool m = (a[i] != 0) | (b ? b[i] != 0 : false) | (c ? c[i] != 0 : false);
    if (u) u[i] = m;
    if (m) Z[i] = (double)NAN;
  }
}

int grid_for(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 8192)); }

}  // namespace

extern "C" {

int smrf_negate_f64(const double* d_in, double* d_out, int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!d_in || !d_out))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (n == 0) return SMRF_OK;
  hipLaunchKernelGGL(negate_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_in, d_out, (long long)n);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_mask_apply_f64(double* d_Z, const uint8_t* d_a, const uint8_t* d_b, const uint8_t* d_c, uint8_t* d_union,
                        int64_t n, void* stream) {
  if (n < 0 || (n > 0 && (!d_Z || !d_a))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (n == 0) return SMRF_OK;
  hipLaunchKernelGGL(mask_apply_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_Z, d_a, d_b, d_c,
                     d_union, (long long)n);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_gradient_slope_f64(const double* d_Z, double* d_S, int rows, int cols, double cellsize, void* stream) {
  if (!d_Z || !d_S) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 2 || cols < 2) return smrf_fail(SMRF_E_ARG, "np.gradient needs at least 2 cells per axis (got %d x %d)", rows, cols);
  const long long n = (long long)rows * cols;
  const int blocks = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 8192));
  hipLaunchKernelGGL(slope_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, d_Z, d_S, rows, cols, cellsize);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // extern "C"
