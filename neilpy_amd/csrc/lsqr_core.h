// SciPy's LSQR recurrence on the device, shared by the matrix-free solvers (springs.hip, fda.hip):
// the scalars of scipy/sparse/linalg/_isolve/lsqr.py:324-555 (damp = 0) in float64 with no FMA
// contraction, stepped by one-lane device code between the vector kernels, plus the fixed-tree
// block reduction the vector kernels end in.  Everything lives in an anonymous namespace: each
// translation unit gets its own copy.
#pragma once
#include <algorithm>
#include <cmath>

#include "smrf_common.h"

namespace {

#ifndef SMRF_LSQR_MAXB
#define SMRF_LSQR_MAXB 2048
#endif
constexpr int MAXB = SMRF_LSQR_MAXB;   // most blocks per vector kernel = partial sums per reduction

struct Sc {
  double alfa, beta, inv_alfa, inv_beta;
  double rhobar, phibar, bnorm, anorm, ddnorm, xxnorm, z, cs2, sn2;
  double t1, t2, inv_rho, xnorm, tau;
  double atol, btol, ctol;
  long long itn, iter_lim, nunk;
  int istop, done, beta_pos, pad;
  // the split rotation of the single-device spring solver (rho_step / alfa_rest_step below)
  double cs, sn, rho, phi, t1_prev;
};

// 2-D walk over the band's cells without an integer division per cell: blockIdx.x picks 256 columns,
// blockIdx.y strides over the rows.  Defines r, c and the flat index i.
#define SMRF_FOR_CELLS(rows_, cols_)                                                        \
  for (int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x; r < (rows_); r += gridDim.y) \
    for (long long i = (long long)r * (cols_) + c; c < (cols_) && i >= 0; i = -1)

// the same walk over planes whose rows are ld_ >= cols_ cells apart (i indexes the padded planes; same cells per thread in the
// same order as SMRF_FOR_CELLS, so every partial sum is bit-identical whatever the pitch)
#define SMRF_FOR_CELLS_P(rows_, cols_, ld_)                                                  \
  for (int r = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x; r < (rows_); r += gridDim.y) \
    for (long long i = (long long)r * (ld_) + c; c < (cols_) && i >= 0; i = -1)

// XCD-aware placement of the 2-D walk's tiles.  Workgroups are dealt round-robin over the device's XCDs in dispatch order
// (id = blockIdx.y * gridDim.x + blockIdx.x, XCD = id % nxcd; 8 on an MI355X in SPX mode), each XCD with its own L2; in the
// plain walk the tile that owns row r + 1 of a column chunk therefore sits on another XCD than the tile that owns row r, and
// the stencils' vertical neighbours (v[i + ld], uv[i - ld], the hole bytes) are fetched through two L2s: the hardware
// counters showed 124.5 B per cell and iteration on 8193^2 where the planes' own traffic is 107
// (profiles/r04_lsqr_traffic.md).  Here XCD x takes a contiguous run of tiles in column-chunk-major order (t -> chunk t / gy,
// row phase t % gy), so a chunk's row phases - which march down the raster together - share one L2.  Placement only: tile
// (tx, ty) walks exactly the cells block (tx, ty) of the plain walk does and writes its partial sum to the same slot, so
// every sum is bit-identical whatever nxcd is.  nxcd comes from the device (hipDeviceAttributeNumberOfXccs, lsqr_xcd_count
// below: other partition modes and parts have other counts); nxcd <= 1 is the plain walk.
#ifndef SMRF_LSQR_XCD
#define SMRF_LSQR_XCD 1
#endif
struct LsqrTile { int x, y; };
__device__ __forceinline__ LsqrTile lsqr_tile(int nxcd) {
#if !SMRF_LSQR_XCD
  return LsqrTile{(int)blockIdx.x, (int)blockIdx.y};
#endif
  if (nxcd <= 1) return LsqrTile{(int)blockIdx.x, (int)blockIdx.y};
  const int gx = gridDim.x, gy = gridDim.y, total = gx * gy;
  const int id = blockIdx.y * gx + blockIdx.x;
  int xcd, slot, q, rem;
  if (nxcd == 8) { xcd = id & 7; slot = id >> 3; q = total >> 3; rem = total & 7; }
  else { xcd = id % nxcd; slot = id / nxcd; q = total / nxcd; rem = total % nxcd; }
  // tiles dealt to the XCDs before this one: XCD y gets ceil((total - y) / nxcd) of them
  const int t = xcd * q + (xcd < rem ? xcd : rem) + slot;
  return LsqrTile{t / gy, t % gy};
}
// the walk of SMRF_FOR_CELLS_P for tile tl_ (an LsqrTile) instead of (blockIdx.x, blockIdx.y)
#define SMRF_FOR_CELLS_T(tl_, rows_, cols_, ld_)                                             \
  for (int r = (tl_).y, c = (tl_).x * 256 + threadIdx.x; r < (rows_); r += gridDim.y)       \
    for (long long i = (long long)r * (ld_) + c; c < (cols_) && i >= 0; i = -1)
#define SMRF_TILE_SLOT(tl_) ((tl_).y * gridDim.x + (tl_).x)

// XCDs of the current device (cached per device; 1 = plain walk when the runtime cannot tell)
inline int lsqr_xcd_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 1; }
  int n = __atomic_load_n(&cached[dev], __ATOMIC_ACQUIRE);
  if (n == 0) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeNumberOfXccs, dev) != hipSuccess || n < 1) { (void)hipGetLastError(); n = 1; }
    __atomic_store_n(&cached[dev], n, __ATOMIC_RELEASE);
  }
  return n;
}

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

// block partials -> red[0]
__global__ __launch_bounds__(256) void reduce_kernel(const double* __restrict__ part, int nb, double* __restrict__ out) {
  __shared__ double red[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) s += part[i];
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) out[0] = t;
}

__device__ void beta_step(Sc* sc, double sum_u) {
  const double bt = sqrt(sum_u);
  sc->beta = bt;
  sc->beta_pos = bt > 0;
  if (bt > 0) {
    sc->inv_beta = 1 / bt;
    sc->anorm = sqrt(sc->anorm * sc->anorm + sc->alfa * sc->alfa + bt * bt);
  } else {
    sc->inv_beta = 1.0;   // scipy leaves u unscaled when beta == 0
  }
}

__device__ __forceinline__ double sgn(double a) { return a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0); }

// The plane rotation and what follows it (lsqr.py:443-483) in two halves: cs, sn, rho, phi - hence t1 = phi / rho and
// 1 / rho - need only rhobar and beta, so they exist as soon as |u|^2 is reduced (rho_step), one vector pass BEFORE alfa
// does; the rest needs alfa (alfa_rest_step).  That lets the x and dk = w / rho steps of iteration k ride in the pass that
// makes v_k (which reads the old v and w anyway) instead of a pass of their own.  Same operations on the same operands in
// the same order per quantity as SciPy's loop body: every scalar is bit-identical
// (tests/test_host_logic.py::test_split_rotation_equals_alfa_rot_step replays both orders in Python floats).
__device__ void rho_step(Sc* sc) {                         // right after beta_step of the same iteration
  const double a = sc->rhobar, b = sc->beta;
  double cs, sn, rho;
  if (b == 0) { cs = sgn(a); sn = 0; rho = fabs(a); }
  else if (a == 0) { cs = 0; sn = sgn(b); rho = fabs(b); }
  else if (fabs(b) > fabs(a)) { const double tau = a / b; sn = sgn(b) / sqrt(1 + tau * tau); cs = sn * tau; rho = b / sn; }
  else { const double tau = b / a; cs = sgn(a) / sqrt(1 + tau * tau); sn = cs * tau; rho = a / cs; }
  sc->cs = cs; sc->sn = sn; sc->rho = rho;
  const double phi = cs * sc->phibar;
  sc->phi = phi;
  sc->t1_prev = sc->t1;                                    // the x step of the iteration before, when it was left pending
  sc->t1 = phi / rho;
  sc->inv_rho = 1 / rho;
}
__device__ void alfa_rest_step(Sc* sc, double sum_v) {     // after |v|^2: alfa and everything of alfa_rot_step that needs it
  if (sc->beta_pos) {
    const double a = sqrt(sum_v);
    sc->alfa = a;
    sc->inv_alfa = a > 0 ? 1 / a : 1.0;
  }
  const double alfa = sc->alfa;
  const double cs = sc->cs, sn = sc->sn, rho = sc->rho, phi = sc->phi;
  const double theta = sn * alfa;
  sc->rhobar = -cs * alfa;
  sc->phibar = sn * sc->phibar;
  const double tau = sn * phi;
  sc->t2 = -theta / rho;
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
  sc->xnorm = xnorm;
  sc->tau = tau;
}

__device__ void tests_step(Sc* sc, double sum_dk) {
  const double EPS = 2.220446049250313e-16;
  const double xnorm = sc->xnorm, tau = sc->tau;
  const double nd = sqrt(sum_dk);
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

// reduce the block partials and run the scalar steps in the same launch (one block)
// WHAT 3: part[] = |v|^2 -> alfa + rest of the rotation, part[MAXB..] = |dk|^2 -> tests;  4: part[] = |u|^2 -> beta + rho_step
template <int WHAT, typename B>   // B has .part and .sc
__global__ __launch_bounds__(256) void reduce_scalar_kernel(const B b, int nb) {
  static_assert(WHAT == 3 || WHAT == 4, "3: alfa + tests, 4: beta + rho");
  __shared__ double red[4];
  __shared__ double red2[4];
  Sc* sc = b.sc;
  if (stopped(sc)) return;
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) {
    s0 += b.part[i];
    if (WHAT == 3) s1 += b.part[MAXB + i];
  }
  const double t0 = block_sum(s0, red);
  const double t1 = WHAT == 3 ? block_sum(s1, red2) : 0.0;
  if (threadIdx.x == 0) {
    if (WHAT == 3) {
      alfa_rest_step(sc, t0);
      tests_step(sc, t1);
    }
    if (WHAT == 4) {
      beta_step(sc, t0);
      rho_step(sc);
    }
  }
}


}  // namespace
