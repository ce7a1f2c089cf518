// smrf tail on the device: slope raster (neilpy.py:1785-1786), and the perceptually scaled slope
// map pssm() of the notebook workflows (neilpy.py:846-867).
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

// pssm(): P = uint8(round(255 * (rad2deg(arctan(ve * S)) / 90))) with S the slope above, every step
// rounded to fp64 as NumPy does (no contraction; np.round = round-half-even), then the optional
// colormap lookup rgba[i] = lut[P[i]] (matplotlib: an integer image indexes the 256-entry table).
__global__ __launch_bounds__(256) void pssm_kernel(const double* __restrict__ Z, uint8_t* __restrict__ P,
                                                   double* __restrict__ rgba, const double* __restrict__ lut, int rows,
                                                   int cols, double h, double ve) {
  const long long n = (long long)rows * cols;
  const double h2 = 2.0 * h;
  const double deg = 180.0 / 3.14159265358979323846;     // np.rad2deg multiplies by 180/pi
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    double gy, gx;
    if (r == 0) gy = (Z[i + cols] - Z[i]) / h;
    else if (r == rows - 1) gy = (Z[i] - Z[i - cols]) / h;
    else gy = (Z[i + cols] - Z[i - cols]) / h2;
    if (c == 0) gx = (Z[i + 1] - Z[i]) / h;
    else if (c == cols - 1) gx = (Z[i] - Z[i - 1]) / h;
    else gx = (Z[i + 1] - Z[i - 1]) / h2;
    const double S = sqrt(gx * gx + gy * gy);
    const double q = 255.0 * ((atan(ve * S) * deg) / 90.0);
    // astype(uint8) of a NaN / out-of-range double is platform defined in NumPy; a NaN slope maps to 0 here
    const double rq = rint(q);
    const uint8_t v = rq >= 0.0 && rq <= 255.0 ? (uint8_t)rq : (uint8_t)0;
    if (P) P[i] = v;
    if (rgba) {
      const double2* l = reinterpret_cast<const double2*>(lut) + 2 * v;
      double2* o = reinterpret_cast<double2*>(rgba) + 2 * i;
      o[0] = l[0];
      o[1] = l[1];
    }
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

int smrf_pssm_f64(const double* d_Z, uint8_t* d_P, double* d_rgba, const double* d_lut, int rows, int cols,
                  double cellsize, double ve, void* stream) {
  if (!d_Z || (!d_P && !d_rgba)) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (d_rgba && !d_lut) return smrf_fail(SMRF_E_ARG, "a colormapped result needs the 256 x 4 table");
  if (rows < 2 || cols < 2) return smrf_fail(SMRF_E_ARG, "np.gradient needs at least 2 cells per axis (got %d x %d)", rows, cols);
  const long long n = (long long)rows * cols;
  hipLaunchKernelGGL(pssm_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, d_Z, d_P, d_rgba, d_lut, rows,
                     cols, cellsize, ve);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // extern "C"
