// smrf tail on the device (neilpy.py:1768-1795): the interpolating bicubic spline that
// scipy.interpolate.RectBivariateSpline(rows, cols, Z) (kx = ky = 3, s = 0; FITPACK regrid) builds,
// its evaluation at the lidar points (FITPACK bispeu / fpbisp / fpbspl) and the point test.
//
// Coefficients: the tensor-product interpolation conditions separate into one penta-diagonal
// collocation system per axis (knots = data sites without the 2nd and the 2nd-to-last, fourfold at
// the ends).  The host supplies the LU factors of both 1-D systems (neilpy_amd/spline.py); the
// kernels apply them to every column, then to every row, in place.  float64 throughout.
#include <algorithm>

#include "smrf_common.h"

namespace {

// lu = [l2 | l1 | d | u1 | u2], each m long.  Solve along axis 0 (down the rows) for every column:
// lanes = adjacent columns, so every access is coalesced.
__global__ __launch_bounds__(256) void solve_axis0_kernel(double* __restrict__ C, int rows, int cols,
                                                          const double* __restrict__ lu) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const double *l2 = lu, *l1 = lu + rows, *d = lu + 2 * rows, *u1 = lu + 3 * rows, *u2 = lu + 4 * rows;
  double y1 = 0.0, y2 = 0.0;                           // y[i-1], y[i-2]
  for (int i = 0; i < rows; ++i) {
    double y = C[(long long)i * cols + c];
    y = y - l2[i] * y2;
    y = y - l1[i] * y1;
    C[(long long)i * cols + c] = y;
    y2 = y1;
    y1 = y;
  }
  double x1 = 0.0, x2 = 0.0;                           // x[i+1], x[i+2]
  for (int i = rows - 1; i >= 0; --i) {
    double x = C[(long long)i * cols + c];
    x = x - u1[i] * x1;
    x = x - u2[i] * x2;
    x = x / d[i];
    C[(long long)i * cols + c] = x;
    x2 = x1;
    x1 = x;
  }
}

// Solve along axis 1 (along each row).  Lanes = adjacent rows; a lane walks its row, so the 8
// doubles of a cache line are consumed by 8 consecutive steps of the same lane (L1-resident).
__global__ __launch_bounds__(64) void solve_axis1_kernel(double* __restrict__ C, int rows, int cols,
                                                         const double* __restrict__ lu) {
  const int r = blockIdx.x * 64 + threadIdx.x;
  if (r >= rows) return;
  const double *l2 = lu, *l1 = lu + cols, *d = lu + 2 * cols, *u1 = lu + 3 * cols, *u2 = lu + 4 * cols;
  double* row = C + (long long)r * cols;
  double y1 = 0.0, y2 = 0.0;
  for (int j = 0; j < cols; ++j) {
    double y = row[j];
    y = y - l2[j] * y2;
    y = y - l1[j] * y1;
    row[j] = y;
    y2 = y1;
    y1 = y;
  }
  double x1 = 0.0, x2 = 0.0;
  for (int j = cols - 1; j >= 0; --j) {
    double x = row[j];
    x = x - u1[j] * x1;
    x = x - u2[j] * x2;
    x = x / d[j];
    row[j] = x;
    x2 = x1;
    x1 = x;
  }
}

// ---- the same two solves cut into chunks along the solve axis (smrf_spline_solve_ws_f64) -------------------------------
// One lane per line leaves a 8193 x 8193 raster with 8193 lanes on the whole device and every lane with a chain of 16386
// dependent steps: 7-8 ms per axis.  Both substitutions contract: in y[i] = b[i] - l1 y[i-1] - l2 y[i-2] and in
// x[i] = (y[i] - u1 x[i+1] - u2 x[i+2]) / d[i] the influence of the state decays by |l1| ~ u1 / d ~ 2 - sqrt(3) = 0.268
// per step for the interpolating cubic spline's collocation matrix (interior rows 1/6, 4/6, 1/6).  A chunk therefore
// starts SPLINE_WARM = 64 steps before its first own entry from a zero state - what is left of the unknown true state
// there is 0.268^64 = 3e-37 of it, nothing in float64 - and writes only its own entries.  The first chunk starts at the
// true beginning with the true (zero) state, so it is the sequential solve.  Out of place (in -> out per sweep): a
// chunk's warm-up reads its neighbour's input entries while the neighbour writes its results.
constexpr int SPLINE_WARM = 64;

template <bool BACK>
__global__ __launch_bounds__(256) void solve_axis0_chunk_kernel(const double* __restrict__ in, double* __restrict__ out, int rows,
                                                                int cols, const double* __restrict__ lu, int chunk) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= cols) return;
  const double *l2 = lu, *l1 = lu + rows, *d = lu + 2 * rows, *u1 = lu + 3 * rows, *u2 = lu + 4 * rows;
  const int i0 = blockIdx.y * chunk, i1 = min(rows, i0 + chunk);
  if constexpr (!BACK) {
    double y1 = 0.0, y2 = 0.0;
#pragma unroll 8
    for (int i = max(0, i0 - SPLINE_WARM); i < i1; ++i) {
      double y = in[(long long)i * cols + c];
      y = y - l2[i] * y2;
      y = y - l1[i] * y1;
      if (i >= i0) out[(long long)i * cols + c] = y;
      y2 = y1;
      y1 = y;
    }
  } else {
    double x1 = 0.0, x2 = 0.0;
#pragma unroll 8
    for (int i = min(rows, i1 + SPLINE_WARM) - 1; i >= i0; --i) {
      double x = in[(long long)i * cols + c];
      x = x - u1[i] * x1;
      x = x - u2[i] * x2;
      x = x / d[i];
      if (i < i1) out[(long long)i * cols + c] = x;
      x2 = x1;
      x1 = x;
    }
  }
}

// along each row.  One wave per 64 rows and chunk of columns; the plane is row-major, so a lane that walked its own row
// would touch a different cache line per lane and step.  Tiles of 64 rows x 64 columns go through LDS instead: loaded with
// lanes = columns (64 coalesced 512-byte row segments), swept with lanes = rows (the substitution, state in registers
// across tiles), stored with lanes = columns again.
template <bool BACK>
__global__ __launch_bounds__(64) void solve_axis1_chunk_kernel(const double* __restrict__ in, double* __restrict__ out, int rows,
                                                               int cols, const double* __restrict__ lu, int chunk) {
  __shared__ double tile[64][65];
  const int lane = threadIdx.x;
  const int r0 = blockIdx.x * 64;
  const int nr = min(64, rows - r0);
  const double *l2 = lu, *l1 = lu + cols, *d = lu + 2 * cols, *u1 = lu + 3 * cols, *u2 = lu + 4 * cols;
  const int j0 = blockIdx.y * chunk, j1 = min(cols, j0 + chunk);
  const int lo = BACK ? j0 : max(0, j0 - SPLINE_WARM);        // columns this workgroup reads: [lo, hi)
  const int hi = BACK ? min(cols, j1 + SPLINE_WARM) : j1;
  double s1 = 0.0, s2 = 0.0;                                  // y[j-1], y[j-2]  (BACK: x[j+1], x[j+2])
  const int ntiles = (hi - lo + 63) / 64;
  for (int t = 0; t < ntiles; ++t) {
    // forward: tiles ascend from lo; backward: tiles descend from hi
    const int jt = BACK ? max(lo, hi - 64 * (t + 1)) : lo + 64 * t;
    const int w = BACK ? (hi - 64 * t) - jt : min(64, hi - jt);
    __syncthreads();
    if (lane < w)
      for (int rr = 0; rr < nr; ++rr) tile[rr][lane] = in[(long long)(r0 + rr) * cols + jt + lane];
    __syncthreads();
    if (lane < nr) {
      if constexpr (!BACK) {
        for (int k = 0; k < w; ++k) {
          double y = tile[lane][k];
          y = y - l2[jt + k] * s2;
          y = y - l1[jt + k] * s1;
          tile[lane][k] = y;
          s2 = s1;
          s1 = y;
        }
      } else {
        for (int k = w - 1; k >= 0; --k) {
          double x = tile[lane][k];
          x = x - u1[jt + k] * s1;
          x = x - u2[jt + k] * s2;
          x = x / d[jt + k];
          tile[lane][k] = x;
          s2 = s1;
          s1 = x;
        }
      }
    }
    __syncthreads();
    if (lane < w && jt + lane >= j0 && jt + lane < j1)
      for (int rr = 0; rr < nr; ++rr) out[(long long)(r0 + rr) * cols + jt + lane] = tile[rr][lane];
  }
}

// chunk length along an axis of n entries whose lines fill `line_groups` workgroups: enough chunks to put about `target`
// workgroups (4096 waves either way: the sweeps are chains of dependent loads, only occupancy hides their latency) on the
// device, never shorter than 4 warm-ups (the warm-up is recomputed work: at most 25 %)
int spline_chunk(int n, int line_groups, int target) {
  const int want = std::max(1, target / std::max(1, line_groups));
  int chunk = (n + want - 1) / want;
  chunk = std::max(chunk, 4 * SPLINE_WARM);
  return std::min(chunk, n);
}

// FITPACK fpbspl, k = 3: the 4 B-splines that are non-zero on [t[l], t[l+1]) at x
__device__ __forceinline__ void bspl3(const double* __restrict__ t, int l, double x, double h[4]) {
  double hh[3];
  h[0] = 1.0;
#pragma unroll
  for (int j = 1; j <= 3; ++j) {
#pragma unroll
    for (int i = 0; i < j; ++i) hh[i] = h[i];
    h[0] = 0.0;
#pragma unroll
    for (int i = 0; i < j; ++i) {
      const int li = l + i + 1, lj = li - j;
      const double f = hh[i] / (t[li] - t[lj]);
      h[i] = h[i] + f * (t[li] - x);
      h[i + 1] = f * (x - t[lj]);
    }
  }
}

// interval l (3 <= l <= n-5) with t[l] <= x < t[l+1]; the last interval also takes x = t[n-4]
__device__ __forceinline__ int find_interval(const double* __restrict__ t, int n, double x) {
  int lo = 3, hi = n - 5;                              // fpbisp: while (arg >= t[l+1] && l != n-5) ++l
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (x >= t[mid]) lo = mid; else hi = mid - 1;
  }
  return lo;
}

struct EvalArgs {
  const double* C;        // rows x cols coefficients
  int rows, cols;
  const double *tx, *ty;  // knots along axis 0 (rows + 4) and axis 1 (cols + 4)
  const double *px, *py;  // evaluation coordinates along axis 0 / axis 1
  long long n;
  double* out;
};

// FITPACK fpbisp for scattered points (bispeu): clamp to the knot range, 4 x 4 basis, sum in fpbisp's order
__global__ __launch_bounds__(256) void eval_kernel(const EvalArgs a) {
  const int nx = a.rows + 4, ny = a.cols + 4;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < a.n; i += (long long)gridDim.x * 256) {
    double x = a.px[i], y = a.py[i];
    const double xb = a.tx[3], xe = a.tx[nx - 4], yb = a.ty[3], ye = a.ty[ny - 4];
    if (x < xb) x = xb;
    if (x > xe) x = xe;
    if (y < yb) y = yb;
    if (y > ye) y = ye;
    double v;
    if (x != x || y != y) {
      v = x + y;                                        // NaN coordinates propagate
    } else {
      const int lx = find_interval(a.tx, nx, x), ly = find_interval(a.ty, ny, y);
      double hx[4], hy[4];
      bspl3(a.tx, lx, x, hx);
      bspl3(a.ty, ly, y, hy);
      const double* c = a.C + (long long)(lx - 3) * a.cols + (ly - 3);
      double sp = 0.0;
#pragma unroll
      for (int i1 = 0; i1 < 4; ++i1) {
#pragma unroll
        for (int j1 = 0; j1 < 4; ++j1) sp = sp + c[(long long)i1 * a.cols + j1] * hx[i1] * hy[j1];
      }
      v = sp;
    }
    a.out[i] = v;
  }
}

// is_object_point = abs(elev - z) > elevation_threshold + elevation_scaler * slope   (neilpy.py:1794-1795)
__global__ __launch_bounds__(256) void classify_kernel(const double* __restrict__ elev, const double* __restrict__ slope,
                                                       const double* __restrict__ z, long long n, double thr,
                                                       double scaler, uint8_t* __restrict__ out) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double required = thr + (scaler * slope[i]);
    out[i] = fabs(elev[i] - z[i]) > required;
  }
}

int grid_for(long long n) { return (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, 8192)); }

}  // namespace

extern "C" {

int smrf_spline_solve_f64(double* d_C, int rows, int cols, const double* d_lu_rows, const double* d_lu_cols,
                          void* stream) {
  if (!d_C || !d_lu_rows || !d_lu_cols) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 4 || cols < 4)
    return smrf_fail(SMRF_E_ARG, "a bicubic spline needs at least 4 x 4 cells (got %d x %d)", rows, cols);
  hipLaunchKernelGGL(solve_axis0_kernel, dim3((cols + 255) / 256), dim3(256), 0, (hipStream_t)stream, d_C, rows, cols,
                     d_lu_rows);
  hipLaunchKernelGGL(solve_axis1_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, d_C, rows, cols,
                     d_lu_cols);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_spline_solve_ws_f64(double* d_C, double* d_scratch, int rows, int cols, const double* d_lu_rows,
                             const double* d_lu_cols, void* stream_) {
  if (!d_C || !d_scratch || !d_lu_rows || !d_lu_cols) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 4 || cols < 4)
    return smrf_fail(SMRF_E_ARG, "a bicubic spline needs at least 4 x 4 cells (got %d x %d)", rows, cols);
  hipStream_t st = (hipStream_t)stream_;
  const int g0 = (cols + 255) / 256, c0 = spline_chunk(rows, g0, 1024);
  const dim3 grid0(g0, (rows + c0 - 1) / c0);
  hipLaunchKernelGGL(solve_axis0_chunk_kernel<false>, grid0, dim3(256), 0, st, (const double*)d_C, d_scratch, rows, cols, d_lu_rows, c0);
  hipLaunchKernelGGL(solve_axis0_chunk_kernel<true>, grid0, dim3(256), 0, st, (const double*)d_scratch, d_C, rows, cols, d_lu_rows, c0);
  const int g1 = (rows + 63) / 64, c1 = spline_chunk(cols, g1, 4096);
  const dim3 grid1(g1, (cols + c1 - 1) / c1);
  hipLaunchKernelGGL(solve_axis1_chunk_kernel<false>, grid1, dim3(64), 0, st, (const double*)d_C, d_scratch, rows, cols, d_lu_cols, c1);
  hipLaunchKernelGGL(solve_axis1_chunk_kernel<true>, grid1, dim3(64), 0, st, (const double*)d_scratch, d_C, rows, cols, d_lu_cols, c1);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_spline_eval_f64(const double* d_C, int rows, int cols, const double* d_tx, const double* d_ty,
                         const double* d_px, const double* d_py, int64_t npts, double* d_out, void* stream) {
  if (npts < 0 || !d_C || !d_tx || !d_ty || (npts > 0 && (!d_px || !d_py || !d_out)))
    return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 4 || cols < 4) return smrf_fail(SMRF_E_ARG, "a bicubic spline needs at least 4 x 4 cells");
  if (npts == 0) return SMRF_OK;
  EvalArgs a{d_C, rows, cols, d_tx, d_ty, d_px, d_py, (long long)npts, d_out};
  hipLaunchKernelGGL(eval_kernel, dim3(grid_for(npts)), dim3(256), 0, (hipStream_t)stream, a);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int smrf_classify_points_f64(const double* d_elev, const double* d_slope, const double* d_z, int64_t npts,
                             double elevation_threshold, double elevation_scaler, uint8_t* d_is_object, void* stream) {
  if (npts < 0 || (npts > 0 && (!d_elev || !d_slope || !d_z || !d_is_object))) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (npts == 0) return SMRF_OK;
  hipLaunchKernelGGL(classify_kernel, dim3(grid_for(npts)), dim3(256), 0, (hipStream_t)stream, d_elev, d_slope, d_z,
                     (long long)npts, elevation_threshold, elevation_scaler, d_is_object);
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

}  // extern "C"
