// inpaint_nans_by_springs on the device (neilpy.py:1227-1271): matrix-free LSQR.
//
// The reference assembles a sparse incidence matrix S (one row per 4-neighbour grid edge with at
// least one NaN endpoint, +1 at the lower flat index, -1 at the higher, :1238-1260) and calls
// scipy.sparse.linalg.lsqr(S[:, nan], -S[:, known] @ A[known]) with default tolerances (:1263-1264).
// The answer is LSQR's iterate at its early stop, not the converged harmonic fill, so this file
// reproduces scipy's recurrence, operation order and stopping rule (scipy lsqr.py:324-555,
// damp = 0) in float64 with no FMA contraction, and never forms S:
//   - vectors over unknowns (x, v, w) are full rasters that stay 0 on known cells;
//   - the edge vector u is two planes: uh[i][j] joins (i,j)-(i,j+1), uv[i][j] joins (i,j)-(i+1,j);
//   - S v on an edge is v[lo] - v[hi];  S^T u on a cell is -u_up - u_left + u_right + u_down,
//     accumulated in that order (the order scipy's CSC product visits the springs);
//   - u and v are stored UNSCALED with their scale (1/beta, 1/alfa) kept as a scalar and applied
//     on every read, which reproduces scipy's "u = (1/beta) * u" rounding exactly and makes every
//     kernel in-place safe (a thread only writes the entries it owns).
// The scalar recurrence runs in one-lane kernels between the vector kernels, so an iteration is
// six launches and no host round trip; the host polls the stop flag every few iterations.
// Reductions are two-stage with a fixed tree (no float atomics): results are reproducible.
// HBM-bound: per iteration about 3 plane reads + 2 plane writes of each of u (2 planes), v, w, x.
#include <algorithm>
#include <cmath>

#include "smrf_common.h"

namespace {

constexpr int MAXB = 1024;   // partial sums per reduction

struct Sc {
  double alfa, beta, inv_alfa, inv_beta;
  double rhobar, phibar, bnorm, anorm, ddnorm, xxnorm, z, cs2, sn2;
  double t1, t2, inv_rho, xnorm, tau;
  double atol, btol, ctol;
  long long itn, iter_lim, nunk;
  int istop, done, beta_pos, pad;
};

__device__ __forceinline__ double block_sum(double s, double* red) {
  for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) red[w] = s;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = ((red[0] + red[1]) + (red[2] + red[3]));
  return t;   // valid on thread 0
}

__device__ __forceinline__ bool stopped(const Sc* sc) { return sc->done != 0 || sc->istop != 0; }

// ---- setup ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_kernel(const double* __restrict__ A, uint8_t* __restrict__ hole,
                                                   long long n, double* __restrict__ part) {
  __shared__ double red[4];
  double c = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double a = A[i];
    const bool h = a != a;
    hole[i] = h;
    c += h ? 1.0 : 0.0;
  }
  const double t = block_sum(c, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

// rhs = -S_known @ A_known  (neilpy.py:1263): on an active edge, K[hi] - K[lo] with K = 0 at holes
__global__ __launch_bounds__(256) void rhs_kernel(const double* __restrict__ A, const uint8_t* __restrict__ hole,
                                                  double* __restrict__ uh, double* __restrict__ uv, int rows,
                                                  int cols, double* __restrict__ x, double* __restrict__ v,
                                                  double* __restrict__ w, double* __restrict__ part) {
  __shared__ double red[4];
  const long long n = (long long)rows * cols;
  double s = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const bool h0 = hole[i];
    const double k0 = h0 ? 0.0 : A[i];
    double eh = 0.0, ev = 0.0;
    if (c + 1 < cols) {
      const bool h1 = hole[i + 1];
      if (h0 | h1) eh = (h1 ? 0.0 : A[i + 1]) - k0;
    }
    if (r + 1 < rows) {
      const bool h1 = hole[i + cols];
      if (h0 | h1) ev = (h1 ? 0.0 : A[i + cols]) - k0;
    }
    uh[i] = eh;
    uv[i] = ev;
    x[i] = 0.0;
    v[i] = 0.0;
    w[i] = 0.0;
    s += eh * eh;
    s += ev * ev;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__device__ __forceinline__ double final_sum(const double* part, int nb, double* red) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  return block_sum(s, red);
}

__global__ __launch_bounds__(256) void s_count(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  const double t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    sc->nunk = (long long)t;
    if (sc->iter_lim < 0) sc->iter_lim = 2 * sc->nunk;
    if (sc->nunk == 0) sc->done = 1;
  }
}

__global__ __launch_bounds__(256) void s_bnorm(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  const double t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    const double b = sqrt(t);
    sc->bnorm = b;
    sc->beta = b;
    sc->beta_pos = b > 0;
    sc->inv_beta = b > 0 ? 1 / b : 1.0;
    sc->alfa = 0.0;
    sc->inv_alfa = 1.0;
  }
}

// ---- v = S^T u_s - beta * v_s  (u_s = inv_beta*u, v_s = inv_alfa*v), partial |v|^2 -----------
__global__ __launch_bounds__(256) void atu_kernel(const double* __restrict__ uh, const double* __restrict__ uv,
                                                  const uint8_t* __restrict__ hole, double* __restrict__ v,
                                                  int rows, int cols, const Sc* __restrict__ sc,
                                                  double* __restrict__ part) {
  __shared__ double red[4];
  if (stopped(sc) || !sc->beta_pos) return;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, beta = sc->beta;
  const long long n = (long long)rows * cols;
  double s = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (!hole[i]) continue;
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    double y = 0.0;
    if (r > 0) y = y - ib * uv[i - cols];
    if (c > 0) y = y - ib * uh[i - 1];
    if (c + 1 < cols) y = y + ib * uh[i];
    if (r + 1 < rows) y = y + ib * uv[i];
    const double nv = y - beta * (ia * v[i]);
    v[i] = nv;
    s += nv * nv;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void s_init_alfa(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  if (sc->done) return;
  double t = 0.0;
  if (sc->beta_pos) t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    const double a = sc->beta_pos ? sqrt(t) : 0.0;
    sc->alfa = a;
    sc->inv_alfa = a > 0 ? 1 / a : 1.0;
    sc->rhobar = a;
    sc->phibar = sc->beta;
    if (a * sc->beta == 0) sc->done = 1;      // arnorm == 0: x = 0 is the answer (lsqr.py:386-390)
  }
}

__global__ __launch_bounds__(256) void w_init_kernel(const double* __restrict__ v, double* __restrict__ w,
                                                     long long n, const Sc* __restrict__ sc) {
  if (sc->done) return;
  const double ia = sc->inv_alfa;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) w[i] = ia * v[i];
}

// ---- u = S v_s - alfa * u_s, partial |u|^2 ---------------------------------------------------
__global__ __launch_bounds__(256) void av_kernel(double* __restrict__ uh, double* __restrict__ uv,
                                                 const uint8_t* __restrict__ hole, const double* __restrict__ v,
                                                 int rows, int cols, const Sc* __restrict__ sc,
                                                 double* __restrict__ part) {
  __shared__ double red[4];
  if (stopped(sc)) return;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, alfa = sc->alfa;
  const long long n = (long long)rows * cols;
  double s = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
    const bool h0 = hole[i];
    const double v0 = ia * v[i];
    if (c + 1 < cols) {
      if (h0 | hole[i + 1]) {
        const double nu = (v0 - ia * v[i + 1]) - alfa * (ib * uh[i]);
        uh[i] = nu;
        s += nu * nu;
      }
    }
    if (r + 1 < rows) {
      if (h0 | hole[i + cols]) {
        const double nu = (v0 - ia * v[i + cols]) - alfa * (ib * uv[i]);
        uv[i] = nu;
        s += nu * nu;
      }
    }
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void s_beta(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  if (stopped(sc)) return;
  const double t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    const double b = sqrt(t);
    sc->beta = b;
    sc->beta_pos = b > 0;
    if (b > 0) {
      sc->inv_beta = 1 / b;
      sc->anorm = sqrt(sc->anorm * sc->anorm + sc->alfa * sc->alfa + b * b);
    } else {
      sc->inv_beta = 1.0;   // scipy leaves u unscaled when beta == 0
    }
  }
}

__device__ __forceinline__ double sgn(double a) { return a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0); }

__global__ __launch_bounds__(256) void s_alfa_rot(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  if (stopped(sc)) return;
  double t = 0.0;
  if (sc->beta_pos) t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    if (sc->beta_pos) {
      const double a = sqrt(t);
      sc->alfa = a;
      sc->inv_alfa = a > 0 ? 1 / a : 1.0;
    }
    const double alfa = sc->alfa, beta = sc->beta;
    // cs, sn, rho = _sym_ortho(rhobar, beta)   (lsqr.py:62-94)
    const double a = sc->rhobar, b = beta;
    double cs, sn, rho;
    if (b == 0) { cs = sgn(a); sn = 0; rho = fabs(a); }
    else if (a == 0) { cs = 0; sn = sgn(b); rho = fabs(b); }
    else if (fabs(b) > fabs(a)) { const double tau = a / b; sn = sgn(b) / sqrt(1 + tau * tau); cs = sn * tau; rho = b / sn; }
    else { const double tau = b / a; cs = sgn(a) / sqrt(1 + tau * tau); sn = cs * tau; rho = a / cs; }
    const double theta = sn * alfa;
    sc->rhobar = -cs * alfa;
    const double phi = cs * sc->phibar;
    sc->phibar = sn * sc->phibar;
    const double tau = sn * phi;
    sc->t1 = phi / rho;
    sc->t2 = -theta / rho;
    sc->inv_rho = 1 / rho;
    // the norm(x) estimate (lsqr.py:474-483)
    const double delta = sc->sn2 * rho;
    const double gambar = -sc->cs2 * rho;
    const double rhs = phi - delta * sc->z;
    const double zbar = rhs / gambar;
    const double xnorm = sqrt(sc->xxnorm + zbar * zbar);
    const double gamma = sqrt(gambar * gambar + theta * theta);
    sc->cs2 = gambar / gamma;
    sc->sn2 = theta / gamma;
    sc->z = rhs / gamma;
    sc->xxnorm = sc->xxnorm + sc->z * sc->z;
    sc->xnorm = xnorm;   // for s_tests
    sc->tau = tau;
  }
}

// ---- x += t1*w ; w = v_s + t2*w ; partial |w/rho|^2 ------------------------------------------
__global__ __launch_bounds__(256) void xw_kernel(double* __restrict__ x, double* __restrict__ w,
                                                 const double* __restrict__ v, const uint8_t* __restrict__ hole,
                                                 long long n, const Sc* __restrict__ sc, double* __restrict__ part) {
  __shared__ double red[4];
  if (stopped(sc)) return;
  const double t1 = sc->t1, t2 = sc->t2, ir = sc->inv_rho, ia = sc->inv_alfa;
  double s = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    if (!hole[i]) continue;
    const double ws = w[i];
    const double dk = ir * ws;
    x[i] = x[i] + t1 * ws;
    w[i] = ia * v[i] + t2 * ws;
    s += dk * dk;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

__global__ __launch_bounds__(256) void s_tests(const double* part, int nb, Sc* sc) {
  __shared__ double red[4];
  if (stopped(sc)) return;
  const double t = final_sum(part, nb, red);
  if (threadIdx.x == 0) {
    const double EPS = 2.220446049250313e-16;
    const double xnorm = sc->xnorm, tau = sc->tau;
    const double nd = sqrt(t);
    sc->ddnorm = sc->ddnorm + nd * nd;
    sc->itn += 1;
    const double anorm = sc->anorm, bnorm = sc->bnorm;
    const double acond = anorm * sqrt(sc->ddnorm);
    const double rnorm = sqrt(sc->phibar * sc->phibar);
    const double arnorm = sc->alfa * fabs(tau);
    const double test1 = rnorm / bnorm;
    const double test2 = arnorm / (anorm * rnorm + EPS);
    const double test3 = 1 / (acond + EPS);
    const double t1 = test1 / (1 + anorm * xnorm / bnorm);
    const double rtol = sc->btol + sc->atol * anorm * xnorm / bnorm;
    int istop = 0;
    if (sc->itn >= sc->iter_lim) istop = 7;
    if (1 + test3 <= 1) istop = 6;
    if (1 + test2 <= 1) istop = 5;
    if (1 + t1 <= 1) istop = 4;
    if (test3 <= sc->ctol) istop = 3;
    if (test2 <= sc->atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    sc->istop = istop;
  }
}

__global__ __launch_bounds__(256) void scatter_kernel(double* __restrict__ A, const double* __restrict__ x,
                                                      const uint8_t* __restrict__ hole, long long n) {
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    if (hole[i]) A[i] = x[i];
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

extern "C" {

size_t smrf_springs_workspace_bytes(int rows, int cols) {
  const size_t n = (size_t)rows * (size_t)cols;
  return 5 * align_up(n * sizeof(double)) + align_up(n) + align_up(MAXB * sizeof(double)) +
         align_up(sizeof(Sc));
}

int smrf_springs_lsqr_f64(double* d_A, int rows, int cols, double atol, double btol, double conlim,
                          int64_t iter_lim, int* h_istop, int64_t* h_itn, int64_t* h_n_unknown,
                          void* d_workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_A || !h_istop || !h_itn) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad raster size %d x %d", rows, cols);
  if (!d_workspace || workspace_bytes < smrf_springs_workspace_bytes(rows, cols))
    return smrf_fail(SMRF_E_WORKSPACE, "springs workspace too small");
  const long long n = (long long)rows * cols;
  char* p = (char*)d_workspace;
  const size_t pl = align_up((size_t)n * sizeof(double));
  double* x = (double*)p; p += pl;
  double* v = (double*)p; p += pl;
  double* w = (double*)p; p += pl;
  double* uh = (double*)p; p += pl;
  double* uv = (double*)p; p += pl;
  uint8_t* hole = (uint8_t*)p; p += align_up((size_t)n);
  double* part = (double*)p; p += align_up(MAXB * sizeof(double));
  Sc* sc = (Sc*)p;

  const int nb = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, MAXB));
  Sc h{};
  h.atol = atol; h.btol = btol; h.ctol = conlim > 0 ? 1 / conlim : 0.0;
  h.cs2 = -1.0; h.iter_lim = iter_lim;
  SMRF_HIP_CHECK(hipMemcpyAsync(sc, &h, sizeof(h), hipMemcpyHostToDevice, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));   // h is a stack object

  hipLaunchKernelGGL(mask_kernel, dim3(nb), dim3(256), 0, stream, d_A, hole, n, part);
  hipLaunchKernelGGL(s_count, dim3(1), dim3(256), 0, stream, part, nb, sc);
  hipLaunchKernelGGL(rhs_kernel, dim3(nb), dim3(256), 0, stream, d_A, hole, uh, uv, rows, cols, x, v, w, part);
  hipLaunchKernelGGL(s_bnorm, dim3(1), dim3(256), 0, stream, part, nb, sc);
  hipLaunchKernelGGL(atu_kernel, dim3(nb), dim3(256), 0, stream, uh, uv, hole, v, rows, cols, sc, part);
  hipLaunchKernelGGL(s_init_alfa, dim3(1), dim3(256), 0, stream, part, nb, sc);
  hipLaunchKernelGGL(w_init_kernel, dim3(nb), dim3(256), 0, stream, v, w, n, sc);
  SMRF_LAUNCH_CHECK();

  Sc out{};
  SMRF_HIP_CHECK(hipMemcpyAsync(&out, sc, sizeof(out), hipMemcpyDeviceToHost, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  const long long lim = out.iter_lim;
  int chunk = 4;
  while (!out.done && out.istop == 0 && out.itn < lim) {
    for (int k = 0; k < chunk; ++k) {
      hipLaunchKernelGGL(av_kernel, dim3(nb), dim3(256), 0, stream, uh, uv, hole, v, rows, cols, sc, part);
      hipLaunchKernelGGL(s_beta, dim3(1), dim3(256), 0, stream, part, nb, sc);
      hipLaunchKernelGGL(atu_kernel, dim3(nb), dim3(256), 0, stream, uh, uv, hole, v, rows, cols, sc, part);
      hipLaunchKernelGGL(s_alfa_rot, dim3(1), dim3(256), 0, stream, part, nb, sc);
      hipLaunchKernelGGL(xw_kernel, dim3(nb), dim3(256), 0, stream, x, w, v, hole, n, sc, part);
      hipLaunchKernelGGL(s_tests, dim3(1), dim3(256), 0, stream, part, nb, sc);
    }
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipMemcpyAsync(&out, sc, sizeof(out), hipMemcpyDeviceToHost, stream));
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
    chunk = std::min(32, chunk * 2);
  }
  if (out.nunk > 0) {
    hipLaunchKernelGGL(scatter_kernel, dim3(nb), dim3(256), 0, stream, d_A, x, hole, n);
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  }
  *h_istop = out.istop;
  *h_itn = (int64_t)out.itn;
  if (h_n_unknown) *h_n_unknown = (int64_t)out.nunk;
  return SMRF_OK;
}

}  // extern "C"
