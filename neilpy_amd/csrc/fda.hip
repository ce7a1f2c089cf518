// inpaint_nans_by_fda on the device (neilpy.py:1170-1216): matrix-free LSQR on the finite-difference
// (second-difference) equations.
//
// The reference assembles, per raster cell i, one equation row: the vertical second difference
// [1, -2, 1] when the cell is not in the first or last raster row, plus the horizontal one when it
// is not in the first or last column (COO duplicates on the centre are summed: -4 in the interior,
// -2 on the border rows/columns, no row at the four corners, :1180-1194).  It then keeps the rows
// that touch a NaN cell, ONCE PER NaN ENTRY of the row (`k = fda[:, nan].nonzero()[0]` lists a row
// index once for every stored entry, :1207-1209), and calls
// scipy.sparse.linalg.lsqr(fda[k][:, nan], (-fda[:, known] @ A[known])[k]) with default
// tolerances (:1210).  `fast=True` only pre-filters rows that could not touch a NaN anyway, so it
// does not change the result.  As with the springs, the answer is LSQR's iterate at its stop, so
// this file follows scipy's recurrence (lsqr_core.h) and its products' accumulation order:
//   - x, v, w are rasters that stay 0 on known cells; u is one raster over equation cells with the
//     multiplicity cnt[i] (0..5 = NaN cells among centre/up/down/left/right of the stencil) beside it;
//   - A v on an equation cell: sum over its NaN stencil cells in ascending flat index (up, left,
//     centre, right, down), as scipy's CSR product does;
//   - A^T u on a NaN cell: contributions of the equation rows above, left, own, right, below in
//     that order, each added cnt times (the duplicated rows of scipy's CSC product);
//   - norms count every equation cnt times.
// Single device only (the reference's callers use it on small rasters; smrf() does not call it).
#include "lsqr_core.h"

namespace {

struct Fda {
  double *x, *v, *w, *u;
  uint8_t *hole, *cnt;
  double* part;
  double* red;
  Sc* sc;
  int rows, cols;
  int nxcd;           // XCDs of the device (lsqr_tile's placement)
};

__device__ __forceinline__ bool has_v(const Fda& b, int r) { return r >= 1 && r <= b.rows - 2; }
__device__ __forceinline__ bool has_h(const Fda& b, int c) { return c >= 1 && c <= b.cols - 2; }

__global__ __launch_bounds__(256) void fda_mask_kernel(const double* __restrict__ A, const Fda b) {
  __shared__ double red[4];
  const long long n = (long long)b.rows * b.cols;
  double c = 0.0;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    const double a = A[i];
    const bool h = a != a;
    b.hole[i] = h;
    c += h ? 1.0 : 0.0;
  }
  const double t = block_sum(c, red);
  if (threadIdx.x == 0) b.part[blockIdx.x] = t;
}

__global__ void fda_count(const Fda b) {
  Sc* sc = b.sc;
  sc->nunk = (long long)b.red[0];
  if (sc->iter_lim < 0) sc->iter_lim = 2 * sc->nunk;
  if (sc->nunk == 0) sc->done = 1;
}

// multiplicity and right-hand side of every equation cell: rhs = (-fda[:, known]) @ A[known]
__global__ __launch_bounds__(256) void fda_rhs_kernel(const double* __restrict__ A, const Fda b) {
  __shared__ double red[4];
  const int rows = b.rows, cols = b.cols;
  double s = 0.0;
  SMRF_FOR_CELLS(rows, cols) {
    const bool pv = has_v(b, r), ph = has_h(b, c);
    int cnt = 0;
    double y = 0.0;
    if (pv | ph) {
      const double cc = -2.0 * ((pv ? 1 : 0) + (ph ? 1 : 0));
      if (pv) { if (b.hole[i - cols]) ++cnt; else y = y + (-1.0) * A[i - cols]; }
      if (ph) { if (b.hole[i - 1]) ++cnt; else y = y + (-1.0) * A[i - 1]; }
      if (b.hole[i]) ++cnt; else y = y + (-cc) * A[i];
      if (ph) { if (b.hole[i + 1]) ++cnt; else y = y + (-1.0) * A[i + 1]; }
      if (pv) { if (b.hole[i + cols]) ++cnt; else y = y + (-1.0) * A[i + cols]; }
    }
    b.cnt[i] = (uint8_t)cnt;
    const double ui = cnt > 0 ? y : 0.0;
    b.u[i] = ui;
    b.x[i] = 0.0;
    b.v[i] = 0.0;
    b.w[i] = 0.0;
    s += cnt * (ui * ui);
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) b.part[blockIdx.y * gridDim.x + blockIdx.x] = t;
}

__global__ void fda_bnorm(const Fda b) {
  Sc* sc = b.sc;
  const double bn = sqrt(b.red[0]);
  sc->bnorm = bn;
  sc->beta = bn;
  sc->beta_pos = bn > 0;
  sc->inv_beta = bn > 0 ? 1 / bn : 1.0;
  sc->alfa = 0.0;
  sc->inv_alfa = 1.0;
}

// A^T u_s on one NaN cell: the equations whose stencil holds it, each cnt times, in SciPy's CSC order
__device__ __forceinline__ double fda_col_dot(const Fda& b, long long i, int r, int c, double ib) {
  const int rows = b.rows, cols = b.cols;
  double y = 0.0;
  if (r >= 1 && has_v(b, r - 1)) {                      // the row above holds this cell as its "down" entry
    const double t = ib * b.u[i - cols];
    for (int k = b.cnt[i - cols]; k > 0; --k) y = y + t;
  }
  if (c >= 1 && has_h(b, c - 1)) {
    const double t = ib * b.u[i - 1];
    for (int k = b.cnt[i - 1]; k > 0; --k) y = y + t;
  }
  {
    const bool pv = has_v(b, r), ph = has_h(b, c);
    if (pv | ph) {
      const double cc = -2.0 * ((pv ? 1 : 0) + (ph ? 1 : 0));
      const double t = cc * (ib * b.u[i]);
      for (int k = b.cnt[i]; k > 0; --k) y = y + t;
    }
  }
  if (c + 1 < cols && has_h(b, c + 1)) {
    const double t = ib * b.u[i + 1];
    for (int k = b.cnt[i + 1]; k > 0; --k) y = y + t;
  }
  if (r + 1 < rows && has_v(b, r + 1)) {
    const double t = ib * b.u[i + cols];
    for (int k = b.cnt[i + cols]; k > 0; --k) y = y + t;
  }
  return y;
}

// v = A^T u_s - beta * v_s on the NaN cells, partial |v|^2 (the set-up's first v; smrf_fda_apply_f64)
__global__ __launch_bounds__(256) void fda_atu_kernel(const Fda b) {
  __shared__ double red[4];
  const Sc* sc = b.sc;
  if (stopped(sc) || !sc->beta_pos) return;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, beta = sc->beta;
  const int rows = b.rows, cols = b.cols;
  double s = 0.0;
  const LsqrTile tl = lsqr_tile(b.nxcd);                      // XCD-aware placement of the walk (lsqr_core.h)
  SMRF_FOR_CELLS_T(tl, rows, cols, cols) {
    if (!b.hole[i]) continue;
    const double nv = fda_col_dot(b, i, r, c, ib) - beta * (ia * b.v[i]);
    b.v[i] = nv;
    s += nv * nv;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) b.part[SMRF_TILE_SLOT(tl)] = t;
}

// The iteration's v pass with the x / w / dk steps riding in it (round 5; springs.hip: atuxw_kernel has the derivation):
//   w_{k-1} = v_{k-1} / alfa_{k-1} + t2_{k-1} w_{k-2} (k = 1: w_0 = v_0 / alfa_0), dk_k = w_{k-1} / rho_k, |dk|^2,
//   k even: x_k = (x_{k-2} + t1_{k-1} w_{k-2}) + t1_k w_{k-1};  v_k = A^T u_k - beta_k v_{k-1}, |v|^2.
// part[0..] <- |v|^2, part[MAXB..] <- |dk|^2.  Same operations per entry and the same cells per partial sum as the
// xw + atu passes it replaces: x, istop, itn are bit-identical.
__global__ __launch_bounds__(256) void fda_atuxw_kernel(const Fda b) {
  __shared__ double red[4];
  __shared__ double red2[4];
  const Sc* sc = b.sc;
  if (stopped(sc)) return;
  const long long itn = sc->itn;
  const bool first = itn == 0, xupd = (itn & 1) != 0, bpos = sc->beta_pos != 0;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, beta = sc->beta;
  const double t1 = sc->t1, t1p = sc->t1_prev, t2 = sc->t2, ir = sc->inv_rho;
  double s = 0.0, sd = 0.0;
  const LsqrTile tl = lsqr_tile(b.nxcd);
  SMRF_FOR_CELLS_T(tl, b.rows, b.cols, b.cols) {
    if (!b.hole[i]) continue;
    const double vs = ia * b.v[i];
    double wn;
    if (first) {
      wn = vs;
    } else {
      const double wo = b.w[i];
      wn = vs + t2 * wo;
      if (xupd) b.x[i] = (b.x[i] + t1p * wo) + t1 * wn;
    }
    b.w[i] = wn;
    const double dk = ir * wn;
    sd += dk * dk;
    if (bpos) {
      const double nv = fda_col_dot(b, i, r, c, ib) - beta * vs;
      b.v[i] = nv;
      s += nv * nv;
    }
  }
  const double t = block_sum(s, red);
  const double td = block_sum(sd, red2);
  if (threadIdx.x == 0) { b.part[SMRF_TILE_SLOT(tl)] = t; b.part[MAXB + SMRF_TILE_SLOT(tl)] = td; }
}

__global__ void fda_init_alfa(const Fda b) {
  Sc* sc = b.sc;
  if (sc->done) return;
  const double a = sc->beta_pos ? sqrt(b.red[0]) : 0.0;
  sc->alfa = a;
  sc->inv_alfa = a > 0 ? 1 / a : 1.0;
  sc->rhobar = a;
  sc->phibar = sc->beta;
  if (a * sc->beta == 0) sc->done = 1;      // arnorm == 0: x = 0 is the answer (lsqr.py:386-390)
}

// A v_s on one equation cell (ascending flat index of the stencil's NaN cells)
__device__ __forceinline__ double fda_row_dot(const Fda& b, long long i, int r, int c, double ia) {
  const bool pv = has_v(b, r), ph = has_h(b, c);
  const int cols = b.cols;
  double y = 0.0;
  if (pv && b.hole[i - cols]) y = y + ia * b.v[i - cols];
  if (ph && b.hole[i - 1]) y = y + ia * b.v[i - 1];
  if (b.hole[i]) y = y + (-2.0 * ((pv ? 1 : 0) + (ph ? 1 : 0))) * (ia * b.v[i]);
  if (ph && b.hole[i + 1]) y = y + ia * b.v[i + 1];
  if (pv && b.hole[i + cols]) y = y + ia * b.v[i + cols];
  return y;
}

// u = A v_s - alfa * u_s, partial |u|^2
__global__ __launch_bounds__(256) void fda_av_kernel(const Fda b) {
  __shared__ double red[4];
  const Sc* sc = b.sc;
  if (stopped(sc)) return;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, alfa = sc->alfa;
  double s = 0.0;
  const LsqrTile tl = lsqr_tile(b.nxcd);                      // XCD-aware placement of the walk (lsqr_core.h)
  SMRF_FOR_CELLS_T(tl, b.rows, b.cols, b.cols) {
    const int cnt = b.cnt[i];
    if (cnt == 0) continue;
    const double nu = fda_row_dot(b, i, r, c, ia) - alfa * (ib * b.u[i]);
    b.u[i] = nu;
    s += cnt * (nu * nu);
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) b.part[SMRF_TILE_SLOT(tl)] = t;
}

__global__ __launch_bounds__(256) void fda_scatter_kernel(double* __restrict__ A, const Fda b) {
  const long long n = (long long)b.rows * b.cols;
  const bool pend = (b.sc->itn & 1) != 0;                 // stopped at an odd iteration: x still lacks t1_k w_{k-1} (fda_atuxw_kernel)
  const double t1 = b.sc->t1;
  for (long long i = blockIdx.x * 256ll + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    if (b.hole[i]) A[i] = pend ? b.x[i] + t1 * b.w[i] : b.x[i];
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct FdaLayout { size_t plane, bytes, x, v, w, u, hole, cnt, part, red, sc, total; };
FdaLayout fda_layout(int rows, int cols) {
  FdaLayout L;
  const size_t n = (size_t)rows * (size_t)cols;
  L.plane = align_up(n * sizeof(double));
  L.bytes = align_up(n);
  size_t o = 0;
  L.x = o; o += L.plane;
  L.v = o; o += L.plane;
  L.w = o; o += L.plane;
  L.u = o; o += L.plane;
  L.hole = o; o += L.bytes;
  L.cnt = o; o += L.bytes;
  L.part = o; o += align_up(2 * MAXB * sizeof(double));
  L.red = o; o += 256;
  L.sc = o; o += align_up(sizeof(Sc));
  L.total = o;
  return L;
}

}  // namespace

extern "C" {

size_t smrf_fda_workspace_bytes(int rows, int cols) { return fda_layout(rows, cols).total; }

int smrf_fda_lsqr_f64(double* d_A, int rows, int cols, double atol, double btol, double conlim, int64_t iter_lim,
                      int* h_istop, int64_t* h_itn, int64_t* h_n_unknown, void* d_workspace, size_t workspace_bytes,
                      void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_A || !h_istop || !h_itn) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad raster size %d x %d", rows, cols);
  const FdaLayout L = fda_layout(rows, cols);
  if (!d_workspace || workspace_bytes < L.total) return smrf_fail(SMRF_E_WORKSPACE, "fda workspace too small");
  char* p = (char*)d_workspace;
  Fda b;
  b.x = (double*)(p + L.x); b.v = (double*)(p + L.v); b.w = (double*)(p + L.w); b.u = (double*)(p + L.u);
  b.hole = (uint8_t*)(p + L.hole); b.cnt = (uint8_t*)(p + L.cnt);
  b.part = (double*)(p + L.part); b.red = (double*)(p + L.red); b.sc = (Sc*)(p + L.sc);
  b.rows = rows; b.cols = cols; b.nxcd = lsqr_xcd_count();

  Sc h{};
  h.atol = atol; h.btol = btol; h.ctol = conlim > 0 ? 1 / conlim : 0.0;
  h.cs2 = -1.0; h.iter_lim = iter_lim;
  SMRF_HIP_CHECK(hipMemcpyAsync(b.sc, &h, sizeof(h), hipMemcpyHostToDevice, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));   // h is a stack object

  const long long n = (long long)rows * cols;
  const int nb1 = (int)std::max<long long>(1, std::min<long long>((n + 255) / 256, MAXB));
  const int cb = (cols + 255) / 256;
  const dim3 g2(cb, std::max(1, std::min(rows, MAXB / std::max(cb, 1))));
  const int nb = (int)(g2.x * g2.y);
  auto reduce = [&](int count) {
    hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)b.part, count, b.red);
  };
  hipLaunchKernelGGL(fda_mask_kernel, dim3(nb1), dim3(256), 0, stream, (const double*)d_A, b);
  reduce(nb1);
  hipLaunchKernelGGL(fda_count, dim3(1), dim3(1), 0, stream, b);
  hipLaunchKernelGGL(fda_rhs_kernel, g2, dim3(256), 0, stream, (const double*)d_A, b);
  reduce(nb);
  hipLaunchKernelGGL(fda_bnorm, dim3(1), dim3(1), 0, stream, b);
  hipLaunchKernelGGL(fda_atu_kernel, g2, dim3(256), 0, stream, b);
  reduce(nb);
  hipLaunchKernelGGL(fda_init_alfa, dim3(1), dim3(1), 0, stream, b);
  SMRF_LAUNCH_CHECK();

  Sc out{};
  SMRF_HIP_CHECK(hipMemcpyAsync(&out, b.sc, sizeof(out), hipMemcpyDeviceToHost, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  const long long lim = out.iter_lim;
  // Iteration k = [w_{k-1}, dk_k, (x), v_k] [alfa_k, rotation, tests_k] [u_{k+1}] [beta_{k+1}, rho_{k+1}], as in springs.hip
  if (!out.done && out.istop == 0 && out.itn < lim) {
    hipLaunchKernelGGL(fda_av_kernel, g2, dim3(256), 0, stream, b);
    hipLaunchKernelGGL((reduce_scalar_kernel<4, Fda>), dim3(1), dim3(256), 0, stream, b, nb);
    SMRF_LAUNCH_CHECK();
  }
  int chunk = 4;
  while (!out.done && out.istop == 0 && out.itn < lim) {
    for (int k = 0; k < chunk; ++k) {
      hipLaunchKernelGGL(fda_atuxw_kernel, g2, dim3(256), 0, stream, b);
      hipLaunchKernelGGL((reduce_scalar_kernel<3, Fda>), dim3(1), dim3(256), 0, stream, b, nb);
      hipLaunchKernelGGL(fda_av_kernel, g2, dim3(256), 0, stream, b);
      hipLaunchKernelGGL((reduce_scalar_kernel<4, Fda>), dim3(1), dim3(256), 0, stream, b, nb);
    }
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipMemcpyAsync(&out, b.sc, sizeof(out), hipMemcpyDeviceToHost, stream));
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
    chunk = std::min(64, chunk * 2);
  }
  if (out.nunk > 0) {
    hipLaunchKernelGGL(fda_scatter_kernel, dim3(nb1), dim3(256), 0, stream, d_A, b);
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  }
  *h_istop = out.istop;
  *h_itn = (int64_t)out.itn;
  if (h_n_unknown) *h_n_unknown = (int64_t)out.nunk;
  return SMRF_OK;
}

// Diagnostic entry (tests only; not used by inpaint_nans_by_fda): the operator the solver iterates with, applied
// once.  d_rhs / d_cnt receive the right-hand side and the multiplicity of every equation cell, d_Av = A v on the
// equation cells (v read on the NaN cells of d_A), d_Atu = A^T u on the NaN cells (u read on the equation cells,
// every equation counted cnt times).  tests/test_fda.py compares them with the reference's explicit sparse
// system (neilpy.py:1180-1209) so that the LSQR iteration-count tolerance is not the only guard on the stencils.
int smrf_fda_apply_f64(const double* d_A, int rows, int cols, const double* d_v, const double* d_u, double* d_rhs,
                       uint8_t* d_cnt, double* d_Av, double* d_Atu, void* d_workspace, size_t workspace_bytes,
                       void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_A || !d_v || !d_u || !d_rhs || !d_cnt || !d_Av || !d_Atu) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad raster size %d x %d", rows, cols);
  const FdaLayout L = fda_layout(rows, cols);
  if (!d_workspace || workspace_bytes < L.total) return smrf_fail(SMRF_E_WORKSPACE, "fda workspace too small");
  char* p = (char*)d_workspace;
  Fda b;
  b.x = (double*)(p + L.x); b.v = (double*)(p + L.v); b.w = (double*)(p + L.w); b.u = (double*)(p + L.u);
  b.hole = (uint8_t*)(p + L.hole); b.cnt = (uint8_t*)(p + L.cnt);
  b.part = (double*)(p + L.part); b.red = (double*)(p + L.red); b.sc = (Sc*)(p + L.sc);
  b.rows = rows; b.cols = cols; b.nxcd = lsqr_xcd_count();
  Sc h{};
  h.cs2 = -1.0; h.iter_lim = -1;
  h.inv_alfa = 1.0; h.inv_beta = 1.0; h.beta_pos = 1;     // alfa = beta = 0: the kernels compute the bare products
  SMRF_HIP_CHECK(hipMemcpyAsync(b.sc, &h, sizeof(h), hipMemcpyHostToDevice, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  const size_t n = (size_t)rows * cols;
  const int nb1 = (int)std::max<long long>(1, std::min<long long>(((long long)n + 255) / 256, MAXB));
  const int cb = (cols + 255) / 256;
  const dim3 g2(cb, std::max(1, std::min(rows, MAXB / std::max(cb, 1))));
  hipLaunchKernelGGL(fda_mask_kernel, dim3(nb1), dim3(256), 0, stream, d_A, b);
  hipLaunchKernelGGL(fda_rhs_kernel, g2, dim3(256), 0, stream, d_A, b);
  SMRF_LAUNCH_CHECK();
  SMRF_HIP_CHECK(hipMemcpyAsync(d_rhs, b.u, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  SMRF_HIP_CHECK(hipMemcpyAsync(d_cnt, b.cnt, n, hipMemcpyDeviceToDevice, stream));
  SMRF_HIP_CHECK(hipMemcpyAsync(b.v, d_v, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(fda_av_kernel, g2, dim3(256), 0, stream, b);          // u = A v - 0 * u
  SMRF_LAUNCH_CHECK();
  SMRF_HIP_CHECK(hipMemcpyAsync(d_Av, b.u, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  SMRF_HIP_CHECK(hipMemcpyAsync(b.u, d_u, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  hipLaunchKernelGGL(fda_atu_kernel, g2, dim3(256), 0, stream, b);         // v = A^T u - 0 * v
  SMRF_LAUNCH_CHECK();
  SMRF_HIP_CHECK(hipMemcpyAsync(d_Atu, b.v, n * sizeof(double), hipMemcpyDeviceToDevice, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  return SMRF_OK;
}

}  // extern "C"
