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
// The scalar recurrence runs in one-lane kernels between the vector kernels, so an iteration needs
// no host round trip; the host polls the stop flag every few iterations.
// Row-band form: every kernel works on rows [0, rows) of a band whose planes carry one halo row
// above and one below (v and the hole mask below, uv above); the phase entry point lets the host
// exchange those rows and all-reduce the one scalar between phases (neilpy_amd/sharded.py).  The
// single-device solver is the same kernels on a band that is the whole raster.
// Reductions are two-stage with a fixed tree (no float atomics): results are reproducible.
// HBM-bound: per iteration about 3 plane reads + 2 plane writes of each of u (2 planes), v, w, x.
#include "lsqr_core.h"

namespace {

// One row band of the raster.  Plane pointers address the first OWN row; row -1 and row `rows` are
// halo rows (allocated always, meaningful only where has_above / has_below).
struct Band {
  double *x, *v, *w, *uh, *uv;
  uint8_t* hole;
  double* abelow;     // the raster row just below the band (has_below)
  double* part;       // MAXB block partials
  double* red;        // red[0]: the phase's sum (local; the host all-reduces it between phases)
  Sc* sc;
  int rows, cols, has_above, has_below;
  long long ld;       // cells between plane rows (>= cols; the raster A itself is always cols apart)
  int nxcd;           // XCDs of the device (lsqr_tile's placement; from the runtime, 1 = plain walk)
};


// ---- setup ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mask_kernel(const double* __restrict__ A, const Band b) {
  __shared__ double red[4];
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  double c_holes = 0.0;
  SMRF_FOR_CELLS_P(rows, cols, ld) {
    const double a = A[(long long)r * cols + c];
    const bool h = a != a;
    b.hole[i] = h;
    c_holes += h ? 1.0 : 0.0;
  }
  const double t = block_sum(c_holes, red);
  if (threadIdx.x == 0) b.part[blockIdx.y * gridDim.x + blockIdx.x] = t;
}

// rhs = -S_known @ A_known  (neilpy.py:1263): on an active edge, K[hi] - K[lo] with K = 0 at holes
__global__ __launch_bounds__(256) void rhs_kernel(const double* __restrict__ A, const Band b) {
  __shared__ double red[4];
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  double s = 0.0;
  SMRF_FOR_CELLS_P(rows, cols, ld) {
    const long long ia = (long long)r * cols + c;        // A is cols apart, the planes ld
    const bool h0 = b.hole[i];
    const double k0 = h0 ? 0.0 : A[ia];
    double eh = 0.0, ev = 0.0;
    if (c + 1 < cols) {
      const bool h1 = b.hole[i + 1];
      if (h0 | h1) eh = (h1 ? 0.0 : A[ia + 1]) - k0;
    }
    if (r + 1 < rows || b.has_below) {
      const bool h1 = b.hole[i + ld];
      const double a1 = r + 1 < rows ? A[ia + cols] : b.abelow[c];
      if (h0 | h1) ev = (h1 ? 0.0 : a1) - k0;
    }
    b.uh[i] = eh;
    b.uv[i] = ev;
    b.x[i] = 0.0;
    b.v[i] = 0.0;
    b.w[i] = 0.0;
    s += eh * eh;
    s += ev * ev;
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) b.part[blockIdx.y * gridDim.x + blockIdx.x] = t;
}

__global__ void s_count(const Band b) {
  Sc* sc = b.sc;
  sc->nunk = (long long)b.red[0];
  if (sc->iter_lim < 0) sc->iter_lim = 2 * sc->nunk;
  if (sc->nunk == 0) sc->done = 1;
}

__global__ void s_bnorm(const Band b) {
  Sc* sc = b.sc;
  const double bn = sqrt(b.red[0]);
  sc->bnorm = bn;
  sc->beta = bn;
  sc->beta_pos = bn > 0;
  sc->inv_beta = bn > 0 ? 1 / bn : 1.0;
  sc->alfa = 0.0;
  sc->inv_alfa = 1.0;
}

// ---- v = S^T u_s - beta * v_s  (u_s = inv_beta*u, v_s = inv_alfa*v), partial |v|^2: the set-up's first v ----------
// (inside the iteration atuxw_kernel makes v together with the x / w / dk steps)
__global__ __launch_bounds__(256) void atu_kernel(const Band b) {
  __shared__ double red[4];
  const Sc* sc = b.sc;
  if (stopped(sc)) return;
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  const LsqrTile tl = lsqr_tile(b.nxcd);
  if (!sc->beta_pos) return;                             // beta == 0 (b = 0): v and alfa stay as they are (lsqr.py:434-441)
  const double ib = sc->inv_beta, ia = sc->inv_alfa, beta = sc->beta;
  double s = 0.0;
  // The hole byte decides whether a cell loads anything else; it is requested one row of the walk ahead.  Worth 4 % per
  // iteration where holes are sparse (10.8 % of the cells: 1.112 -> 1.070 ms on 8193^2), nothing where they are dense.  Same
  // cells in the same order: every sum is bit-identical.  (Round 4 also built the rows' loads as ONE unconditional batch
  // behind a wave-wide test - hipcc waits for every conditional load before it issues the next, five round trips per row
  // here: 2-7 % SLOWER at both densities, profiles/r04_lsqr_traffic.md; eight waves per SIMD already cover the latencies, and
  // the cells a wave-wide test no longer skips cost traffic.)
  {
    const int c = tl.x * 256 + (int)threadIdx.x;
    if (c < cols) {
      int r = tl.y;
      uint8_t h = r < rows ? b.hole[(long long)r * ld + c] : (uint8_t)0;
      for (; r < rows; r += gridDim.y) {
        const long long i = (long long)r * ld + c;
        const int rn = r + (int)gridDim.y;
        const uint8_t hn = rn < rows ? b.hole[(long long)rn * ld + c] : (uint8_t)0;
        if (h) {
          double y = 0.0;
          if (r > 0 || b.has_above) y = y - ib * b.uv[i - ld];
          if (c > 0) y = y - ib * b.uh[i - 1];
          if (c + 1 < cols) y = y + ib * b.uh[i];
          if (r + 1 < rows || b.has_below) y = y + ib * b.uv[i];
          const double nv = y - beta * (ia * b.v[i]);
          b.v[i] = nv;
          s += nv * nv;
        }
        h = hn;
      }
    }
  }
  const double t = block_sum(s, red);
  if (threadIdx.x == 0) b.part[SMRF_TILE_SLOT(tl)] = t;
}

// block partials of two sums -> red[0], red[1]
__global__ __launch_bounds__(256) void reduce2_kernel(const double* __restrict__ part, int nb, double* __restrict__ out) {
  __shared__ double red[4];
  __shared__ double red2[4];
  double s0 = 0.0, s1 = 0.0;
  for (int i = threadIdx.x; i < nb; i += 256) { s0 += part[i]; s1 += part[MAXB + i]; }
  const double t0 = block_sum(s0, red);
  const double t1 = block_sum(s1, red2);
  if (threadIdx.x == 0) { out[0] = t0; out[1] = t1; }
}

__global__ void s_init_alfa(const Band b) {
  Sc* sc = b.sc;
  if (sc->done) return;
  const double a = sc->beta_pos ? sqrt(b.red[0]) : 0.0;
  sc->alfa = a;
  sc->inv_alfa = a > 0 ? 1 / a : 1.0;
  sc->rhobar = a;
  sc->phibar = sc->beta;
  if (a * sc->beta == 0) sc->done = 1;      // arnorm == 0: x = 0 is the answer (lsqr.py:386-390)
}

// ---- the iteration since round 5 (single device and row bands): the x / w / dk steps ride in the pass that makes v ----
// Iteration k was [atu: v_k, |v|^2] [alfa, rotation] [xwav: x_k, w_k, |dk|^2; u_{k+1}, |u|^2] [tests; beta]: 13 plane
// touches (atu 3 reads + 1 write, xwav 5 + 4).  The rotation's rho, phi - hence t1 = phi / rho and 1 / rho - need only
// rhobar and beta (lsqr_core.h: rho_step), so they exist BEFORE the v pass, and that pass reads v_{k-1} and can read w_{k-2}:
//   atuxw (iteration k):  w_{k-1} = v_{k-1} / alfa_{k-1} + t2_{k-1} w_{k-2}   (k = 1: w_0 = v_0 / alfa_0 - no w_init pass)
//                         dk_k = w_{k-1} / rho_k, |dk|^2;   v_k = S^T u_k - beta_k v_{k-1}, |v|^2
//                         k even: x_k = (x_{k-2} + t1_{k-1} w_{k-2}) + t1_k w_{k-1}  - both steps, from registers
//   [alfa_k, rest of the rotation, tests_k]      av2: u_{k+1} = S v_k - alfa_k u_k, |u|^2      [beta_{k+1}, rho_step]
// x is read and written every SECOND iteration only: 12 touches instead of 13 (99 instead of 106 B per cell and iteration
// with the hole bytes), no separate w_init pass, w never initialised by the set-up.  A solve that stops at an odd k leaves
// x one step behind; scatter_kernel adds t1_k w_{k-1} (w still holds it: av2 and the next atuxw return at once).
// Every vector entry goes through the same operations on the same operands in the same order as before (x_k's two
// roundings, w, dk, v), every partial sum runs over the same cells in the same order into the same slot: x, istop and
// itn are bit-identical to round 4's four-launch form (tools/lsqr_ab.py against a round-4 build; the goldens' istop / itn).
// The row-band form runs the same two vector kernels as phases (PH_AV, PH_ATUXW) with the host's halo rows and all-reduces
// between them: 12 plane touches per iteration where round 4's band phases (av, atu + |w|^2, xw) made 15, and |dk|^2 is now
// SciPy's own sum over (w / rho)^2 (round 4's band form used |w|^2 / rho^2, a rounding apart).
__global__ __launch_bounds__(256) void atuxw_kernel(const Band b) {
  __shared__ double red[4];
  __shared__ double red2[4];
  const Sc* sc = b.sc;
  if (stopped(sc)) return;
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  const LsqrTile tl = lsqr_tile(b.nxcd);
  const long long itn = sc->itn;                           // k - 1
  const bool first = itn == 0, xupd = (itn & 1) != 0, bpos = sc->beta_pos != 0;
  const double ib = sc->inv_beta, ia = sc->inv_alfa, beta = sc->beta;
  const double t1 = sc->t1, t1p = sc->t1_prev, t2 = sc->t2, ir = sc->inv_rho;
  double s = 0.0, sd = 0.0;
  {
    const int c = tl.x * 256 + (int)threadIdx.x;
    if (c < cols) {
      int r = tl.y;
      uint8_t h = r < rows ? b.hole[(long long)r * ld + c] : (uint8_t)0;
      for (; r < rows; r += gridDim.y) {
        const long long i = (long long)r * ld + c;
        const int rn = r + (int)gridDim.y;
        const uint8_t hn = rn < rows ? b.hole[(long long)rn * ld + c] : (uint8_t)0;   // one row of the walk ahead (atu_kernel)
        if (h) {
          const double vs = ia * b.v[i];
          double wn;
          if (first) {
            wn = vs;                                       // w_0 = v_0 / alfa_0 (lsqr.py:379)
          } else {
            const double wo = b.w[i];
            wn = vs + t2 * wo;                             // w_{k-1} (lsqr.py:461, of the iteration before)
            // x_{k-1} then x_k (lsqr.py:460), two roundings as before.  (Non-temporal accesses for x - touched every second
            // iteration only - measured +1.2 % per iteration: profiles/r05_lsqr_split.md)
            if (xupd) b.x[i] = (b.x[i] + t1p * wo) + t1 * wn;
          }
          b.w[i] = wn;
          const double dk = ir * wn;                       // lsqr.py:459
          sd += dk * dk;
          if (bpos) {                                      // beta == 0: v and alfa stay (lsqr.py:434-441)
            double y = 0.0;
            if (r > 0 || b.has_above) y = y - ib * b.uv[i - ld];
            if (c > 0) y = y - ib * b.uh[i - 1];
            if (c + 1 < cols) y = y + ib * b.uh[i];
            if (r + 1 < rows || b.has_below) y = y + ib * b.uv[i];
            const double nv = y - beta * vs;
            b.v[i] = nv;
            s += nv * nv;
          }
        }
        h = hn;
      }
    }
  }
  const double t = block_sum(s, red);
  const double td = block_sum(sd, red2);
  if (threadIdx.x == 0) { b.part[SMRF_TILE_SLOT(tl)] = t; b.part[MAXB + SMRF_TILE_SLOT(tl)] = td; }
}

// ---- u = S v_s - alfa u_s, partial |u|^2 (the hole bytes a cell's tests need are requested one row of the walk ahead) ----
__global__ __launch_bounds__(256) void av2_kernel(const Band b) {
  __shared__ double red[4];
  const Sc* sc = b.sc;
  if (stopped(sc)) return;
  const double ia = sc->inv_alfa, ib = sc->inv_beta, alfa = sc->alfa;
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  double su = 0.0;
  const LsqrTile tl = lsqr_tile(b.nxcd);
  {
    const int c = tl.x * 256 + (int)threadIdx.x;
    if (c < cols) {
      const bool has_r = c + 1 < cols;
      int r = tl.y;
      uint8_t h0 = 0, h1 = 0, hd = 0;
      const int rv = b.has_below ? rows : rows - 1;        // rows with a vertical spring below them (band form: the halo row)
      if (r < rows) {
        const long long i0 = (long long)r * ld + c;
        h0 = b.hole[i0];
        if (has_r) h1 = b.hole[i0 + 1];
        if (r < rv) hd = b.hole[i0 + ld];
      }
      for (; r < rows; r += gridDim.y) {
        const long long i = (long long)r * ld + c;
        const int rn = r + (int)gridDim.y;
        uint8_t n0 = 0, n1 = 0, nd = 0;
        if (rn < rows) {
          const long long in = (long long)rn * ld + c;
          n0 = b.hole[in];
          if (has_r) n1 = b.hole[in + 1];
          if (rn < rv) nd = b.hole[in + ld];
        }
        // v[i] is read before the hole tests on purpose: making it (or a per-tile activity byte) conditional turns independent
        // loads into dependent ones and measured 4-10 % SLOWER on 8192^2 at every hole pattern (gpurun_out/r02/lsqr_ab*.log);
        // planes of known-only regions are never touched as it is
        const double v0 = ia * b.v[i];
        if (has_r) {
          if (h0 | h1) {
            const double nu = (v0 - ia * b.v[i + 1]) - alfa * (ib * b.uh[i]);
            b.uh[i] = nu;
            su += nu * nu;
          }
        }
        if (r < rv) {
          if (h0 | hd) {
            const double nu = (v0 - ia * b.v[i + ld]) - alfa * (ib * b.uv[i]);
            b.uv[i] = nu;
            su += nu * nu;
          }
        }
        h0 = n0; h1 = n1; hd = nd;
      }
    }
  }
  const double tu = block_sum(su, red);
  if (threadIdx.x == 0) b.part[SMRF_TILE_SLOT(tl)] = tu;
}

// row-band form of the scalar steps (the host all-reduces b.red between the phases)
__global__ void s_beta_rho(const Band b) {
  if (stopped(b.sc)) return;
  beta_step(b.sc, b.red[0]);
  rho_step(b.sc);
}
__global__ void s_alfa_tests(const Band b) {               // red[0] = |v|^2, red[1] = |dk|^2 (one 2-element all-reduce)
  if (stopped(b.sc)) return;
  alfa_rest_step(b.sc, b.red[0]);
  tests_step(b.sc, b.red[1]);
}

// set-up of the single-device solver in ONE plane pass: hole mask (from A itself: the neighbours' NaN-ness is read off the
// raster, not off a mask written by an earlier launch), right-hand side, v = 0, x = 0 at the holes; partial hole count and
// partial |b|^2 - mask_kernel + rhs_kernel's cells in rhs_kernel's order (the same |b|^2 bits), without their second read
// of the raster and without w (atuxw_kernel's first pass writes it)
__global__ __launch_bounds__(256) void setup_kernel(const double* __restrict__ A, const Band b) {
  __shared__ double red[4];
  __shared__ double red2[4];
  const int rows = b.rows, cols = b.cols;
  const long long ld = b.ld;
  double s = 0.0, nh = 0.0;
  SMRF_FOR_CELLS_P(rows, cols, ld) {
    const long long ia = (long long)r * cols + c;
    const double a0 = A[ia];
    const bool h0 = a0 != a0;
    const double k0 = h0 ? 0.0 : a0;
    double eh = 0.0, ev = 0.0;
    if (c + 1 < cols) {
      const double a1 = A[ia + 1];
      const bool h1 = a1 != a1;
      if (h0 | h1) eh = (h1 ? 0.0 : a1) - k0;
    }
    if (r + 1 < rows) {
      const double a1 = A[ia + cols];
      const bool h1 = a1 != a1;
      if (h0 | h1) ev = (h1 ? 0.0 : a1) - k0;
    }
    b.hole[i] = h0;
    b.uh[i] = eh;
    b.uv[i] = ev;
    b.v[i] = 0.0;
    if (h0) b.x[i] = 0.0;
    nh += h0 ? 1.0 : 0.0;
    s += eh * eh;
    s += ev * ev;
  }
  const double t = block_sum(s, red);
  const double tn = block_sum(nh, red2);
  if (threadIdx.x == 0) {
    b.part[blockIdx.y * gridDim.x + blockIdx.x] = tn;
    b.part[MAXB + blockIdx.y * gridDim.x + blockIdx.x] = t;
  }
}

__global__ void s_count_bnorm(const Band b) {             // red[0] = holes, red[1] = |b|^2: s_count then s_bnorm
  Sc* sc = b.sc;
  sc->nunk = (long long)b.red[0];
  if (sc->iter_lim < 0) sc->iter_lim = 2 * sc->nunk;
  if (sc->nunk == 0) sc->done = 1;
  const double bn = sqrt(b.red[1]);
  sc->bnorm = bn;
  sc->beta = bn;
  sc->beta_pos = bn > 0;
  sc->inv_beta = bn > 0 ? 1 / bn : 1.0;
  sc->alfa = 0.0;
  sc->inv_alfa = 1.0;
}

// pend: the solve stopped at an odd iteration k - x still lacks t1_k w_{k-1} (atuxw_kernel)
__global__ __launch_bounds__(256) void scatter_kernel(double* __restrict__ A, const Band b) {
  const bool pend = (b.sc->itn & 1) != 0;
  const double t1 = b.sc->t1;
  SMRF_FOR_CELLS_P(b.rows, b.cols, b.ld)
    if (b.hole[i]) A[(long long)r * b.cols + c] = pend ? b.x[i] + t1 * b.w[i] : b.x[i];
}

size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

// workspace carve-up shared by the single-device solver and the band phases.
// offsets (bytes): x, v, w, uh, uv (first ALLOCATED row = halo above), hole, abelow, part, red, sc
struct Layout {
  size_t plane, x, v, w, uh, uv, hole, abelow, part, red, sc, total;
};
// Plane pitch of the single-device solver.  create_dem's grids are one cell wider than a round number as a rule, so rows of
// cols doubles start anywhere in a cache line.  Rows padded to 32 doubles (256 B) measure 2-5 % faster per iteration at 2049 ...
// 16385 columns; on 32769 columns (1.07e9 cells, 8.6 GB per plane) a 4 KB pitch is worth 11 % (26.6 -> 23.6 ms per iteration
// against 22.3 ms on 32768 columns) while it costs 5-9 % at 8193 and 16385 (gpurun_out/r02/lsqr_pitch3.log, lsqr_pitch4.log),
// hence the size rule.  Pad cells are never touched; every thread walks the same cells in the same order as without padding, so
// all sums, the stop iteration and the result are bit-identical.  The band form uses the same pitch; the host
// (neilpy_amd/sharded.py) addresses its halo rows through smrf_springs_band_layout, which reports it.
long long padded_pitch(int cols) {
  const long long a = cols >= 24576 ? 512 : 32;
  return (((long long)cols + a - 1) / a) * a;
}

Layout layout_of(int rows, int cols, long long ld) {
  Layout L;
  const size_t n2 = (size_t)(rows + 2) * (size_t)ld;
  L.plane = align_up(n2 * sizeof(double));
  size_t o = 0;
  L.x = o; o += L.plane;
  L.v = o; o += L.plane;
  L.w = o; o += L.plane;
  L.uh = o; o += L.plane;
  L.uv = o; o += L.plane;
  L.hole = o; o += align_up(n2);
  L.abelow = o; o += align_up((size_t)cols * sizeof(double));
  L.part = o; o += align_up(2 * MAXB * sizeof(double));
  L.red = o; o += 256;
  L.sc = o; o += align_up(sizeof(Sc));
  L.total = o;
  return L;
}
Band band_of(void* ws, int rows, int cols, long long ld, int has_above, int has_below) {
  const Layout L = layout_of(rows, cols, ld);
  char* p = (char*)ws;
  Band b;
  b.x = (double*)(p + L.x) + ld;
  b.v = (double*)(p + L.v) + ld;
  b.w = (double*)(p + L.w) + ld;
  b.uh = (double*)(p + L.uh) + ld;
  b.uv = (double*)(p + L.uv) + ld;
  b.hole = (uint8_t*)(p + L.hole) + ld;
  b.abelow = (double*)(p + L.abelow);
  b.part = (double*)(p + L.part);
  b.red = (double*)(p + L.red);
  b.sc = (Sc*)(p + L.sc);
  b.rows = rows; b.cols = cols; b.ld = ld; b.has_above = has_above; b.has_below = has_below;
  b.nxcd = lsqr_xcd_count();
  return b;
}
enum Phase {
  PH_MASK = 0,        // hole mask of own rows, red[0] = local unknown count
  PH_RHS = 1,         // (after hole/A halo + count all-reduce) rhs, red[0] = local |b|^2
  PH_BNORM = 2,       // (after all-reduce) beta = |b|
  PH_ATU = 3,         // set-up only, (after uv halo): v = S^T u, red[0] = local |v|^2
  PH_INIT_ALFA = 4,   // (after all-reduce) alfa
  // one iteration (round 5: the split rotation, atuxw_kernel / av2_kernel - the single-device solver's kernels):
  PH_AV = 5,          // (after v halo)   u = S v - alfa u, red[0] = local |u|^2
  PH_BETA_RHO = 6,    // (after all-reduce) beta, anorm, rho_step: cs, sn, rho, t1, 1/rho
  PH_ATUXW = 7,       // (after uv halo)  w, dk, (x every second iteration), v;  red[0] = local |v|^2, red[1] = local |dk|^2
  PH_ALFA_TESTS = 8,  // (after ONE 2-element all-reduce) alfa, rest of the rotation, stopping tests, itn += 1
  PH_SCATTER = 10,    // A[hole] = x (+ t1 w when the solve stopped at an odd iteration)
};

// 2-D launch of the stencil kernels: 256 columns per block, rows strided over gridDim.y; at most MAXB blocks
dim3 grid2d(const Band& b) {
  const int cb = (b.cols + 255) / 256;
  const int rb = std::max(1, std::min(b.rows, MAXB / std::max(cb, 1)));
  return dim3(cb, std::max(rb, 1));
}

int run_phase(int phase, double* A, const Band& b, hipStream_t st) {
  const dim3 g2 = grid2d(b);
  const int nb2 = (int)(g2.x * g2.y);
  auto reduce = [&](int count) { hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), 0, st, (const double*)b.part, count, b.red); };
  switch (phase) {
    case PH_MASK: hipLaunchKernelGGL(mask_kernel, g2, dim3(256), 0, st, (const double*)A, b); reduce(nb2); break;
    case PH_RHS:
      hipLaunchKernelGGL(s_count, dim3(1), dim3(1), 0, st, b);
      hipLaunchKernelGGL(rhs_kernel, g2, dim3(256), 0, st, (const double*)A, b);
      reduce(nb2);
      break;
    case PH_BNORM: hipLaunchKernelGGL(s_bnorm, dim3(1), dim3(1), 0, st, b); break;
    case PH_ATU: hipLaunchKernelGGL(atu_kernel, g2, dim3(256), 0, st, b); reduce(nb2); break;
    case PH_INIT_ALFA: hipLaunchKernelGGL(s_init_alfa, dim3(1), dim3(1), 0, st, b); break;   // (w_0 is made by the first atuxw pass)
    case PH_AV: hipLaunchKernelGGL(av2_kernel, g2, dim3(256), 0, st, b); reduce(nb2); break;
    case PH_BETA_RHO: hipLaunchKernelGGL(s_beta_rho, dim3(1), dim3(1), 0, st, b); break;
    case PH_ATUXW:
      hipLaunchKernelGGL(atuxw_kernel, g2, dim3(256), 0, st, b);
      hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, st, (const double*)b.part, nb2, b.red);
      break;
    case PH_ALFA_TESTS: hipLaunchKernelGGL(s_alfa_tests, dim3(1), dim3(1), 0, st, b); break;
    case PH_SCATTER: hipLaunchKernelGGL(scatter_kernel, g2, dim3(256), 0, st, A, b); break;
    default: return smrf_fail(SMRF_E_ARG, "unknown springs phase %d", phase);
  }
  SMRF_LAUNCH_CHECK();
  return SMRF_OK;
}

int init_scalars(const Band& b, double atol, double btol, double conlim, int64_t iter_lim, hipStream_t st) {
  Sc h{};
  h.atol = atol; h.btol = btol; h.ctol = conlim > 0 ? 1 / conlim : 0.0;
  h.cs2 = -1.0; h.iter_lim = iter_lim;
  SMRF_HIP_CHECK(hipMemcpyAsync(b.sc, &h, sizeof(h), hipMemcpyHostToDevice, st));
  SMRF_HIP_CHECK(hipStreamSynchronize(st));   // h is a stack object
  return SMRF_OK;
}

}  // namespace

extern "C" {

size_t smrf_springs_workspace_bytes(int rows, int cols) { return layout_of(rows, cols, padded_pitch(cols)).total; }

int smrf_springs_lsqr_f64(double* d_A, int rows, int cols, double atol, double btol, double conlim,
                          int64_t iter_lim, int* h_istop, int64_t* h_itn, int64_t* h_n_unknown,
                          void* d_workspace, size_t workspace_bytes, void* stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  if (!d_A || !h_istop || !h_itn) return smrf_fail(SMRF_E_ARG, "null pointer");
  if (rows < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad raster size %d x %d", rows, cols);
  if (!d_workspace || workspace_bytes < smrf_springs_workspace_bytes(rows, cols))
    return smrf_fail(SMRF_E_WORKSPACE, "springs workspace too small");
  const Band b = band_of(d_workspace, rows, cols, padded_pitch(cols), 0, 0);
  if (int rc = init_scalars(b, atol, btol, conlim, iter_lim, stream)) return rc;
  const dim3 g2 = grid2d(b);
  const int nb = (int)(g2.x * g2.y);
  // set-up: one plane pass (mask + right-hand side + counts), the first v = S^T u, alfa; then u_1 and beta_1 + rho_step
  hipLaunchKernelGGL(setup_kernel, g2, dim3(256), 0, stream, (const double*)d_A, b);
  hipLaunchKernelGGL(reduce2_kernel, dim3(1), dim3(256), 0, stream, (const double*)b.part, nb, b.red);
  hipLaunchKernelGGL(s_count_bnorm, dim3(1), dim3(1), 0, stream, b);
  hipLaunchKernelGGL(atu_kernel, g2, dim3(256), 0, stream, b);
  hipLaunchKernelGGL(reduce_kernel, dim3(1), dim3(256), 0, stream, (const double*)b.part, nb, b.red);
  hipLaunchKernelGGL(s_init_alfa, dim3(1), dim3(1), 0, stream, b);
  SMRF_LAUNCH_CHECK();
  Sc out{};
  SMRF_HIP_CHECK(hipMemcpyAsync(&out, b.sc, sizeof(out), hipMemcpyDeviceToHost, stream));
  SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  const long long lim = out.iter_lim;
  if (!out.done && out.istop == 0 && out.itn < lim) {
    hipLaunchKernelGGL(av2_kernel, g2, dim3(256), 0, stream, b);
    hipLaunchKernelGGL((reduce_scalar_kernel<4, Band>), dim3(1), dim3(256), 0, stream, b, nb);
    SMRF_LAUNCH_CHECK();
  }
  int chunk = 4;
  while (!out.done && out.istop == 0 && out.itn < lim) {
    for (int k = 0; k < chunk; ++k) {
      // Iteration k = [w_{k-1}, dk_k, (x), v_k] [alfa_k, rotation, tests_k] [u_{k+1}] [beta_{k+1}, rho_{k+1}]  (atuxw_kernel)
      hipLaunchKernelGGL(atuxw_kernel, g2, dim3(256), 0, stream, b);
      hipLaunchKernelGGL((reduce_scalar_kernel<3, Band>), dim3(1), dim3(256), 0, stream, b, nb);
      hipLaunchKernelGGL(av2_kernel, g2, dim3(256), 0, stream, b);
      hipLaunchKernelGGL((reduce_scalar_kernel<4, Band>), dim3(1), dim3(256), 0, stream, b, nb);
    }
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipMemcpyAsync(&out, b.sc, sizeof(out), hipMemcpyDeviceToHost, stream));
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
    chunk = std::min(32, chunk * 2);
  }
  if (out.nunk > 0) {
    hipLaunchKernelGGL(scatter_kernel, g2, dim3(256), 0, stream, d_A, b);
    SMRF_LAUNCH_CHECK();
    SMRF_HIP_CHECK(hipStreamSynchronize(stream));
  }
  *h_istop = out.istop;
  *h_itn = (int64_t)out.itn;
  if (h_n_unknown) *h_n_unknown = (int64_t)out.nunk;
  return SMRF_OK;
}

size_t smrf_springs_band_workspace_bytes(int rows_local, int cols) { return layout_of(rows_local, cols, padded_pitch(cols)).total; }

int smrf_springs_band_layout(int rows_local, int cols, int64_t* h_out) {
  if (!h_out || rows_local < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad band");
  const Layout L = layout_of(rows_local, cols, padded_pitch(cols));
  h_out[0] = (int64_t)L.v;       // v plane    (rows_local + 2 rows, h_out[6] doubles apart, cols of them used; row 0 = halo above)
  h_out[1] = (int64_t)L.uv;      // uv plane   (same shape)
  h_out[2] = (int64_t)L.hole;    // hole plane (rows_local + 2 rows, h_out[6] bytes apart)
  h_out[3] = (int64_t)L.abelow;  // cols doubles: the raster row below the band
  h_out[4] = (int64_t)L.red;     // 2 doubles: the phase sums to all-reduce (PH_ATU uses both, the others the first)
  h_out[5] = (int64_t)L.total;
  h_out[6] = (int64_t)padded_pitch(cols);   // cells between the rows of every plane (>= cols)
  return SMRF_OK;
}

int smrf_springs_band_begin(int rows_local, int cols, double atol, double btol, double conlim, int64_t iter_lim,
                            void* d_workspace, size_t workspace_bytes, void* stream) {
  if (rows_local < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad band size %d x %d", rows_local, cols);
  if (!d_workspace || workspace_bytes < layout_of(rows_local, cols, padded_pitch(cols)).total)
    return smrf_fail(SMRF_E_WORKSPACE, "springs band workspace too small");
  const Band b = band_of(d_workspace, rows_local, cols, padded_pitch(cols), 0, 0);
  SMRF_HIP_CHECK(hipMemsetAsync(d_workspace, 0, layout_of(rows_local, cols, padded_pitch(cols)).total, (hipStream_t)stream));
  return init_scalars(b, atol, btol, conlim, iter_lim, (hipStream_t)stream);
}

int smrf_springs_band_phase(int phase, double* d_A_band, int rows_local, int cols, int has_above, int has_below,
                            void* d_workspace, size_t workspace_bytes, void* stream) {
  if (!d_A_band || rows_local < 1 || cols < 1) return smrf_fail(SMRF_E_ARG, "bad band");
  if (!d_workspace || workspace_bytes < layout_of(rows_local, cols, padded_pitch(cols)).total)
    return smrf_fail(SMRF_E_WORKSPACE, "springs band workspace too small");
  const Band b = band_of(d_workspace, rows_local, cols, padded_pitch(cols), has_above != 0, has_below != 0);
  return run_phase(phase, d_A_band, b, (hipStream_t)stream);
}

int smrf_springs_band_status(const void* d_workspace, int rows_local, int cols, int* h_istop, int64_t* h_itn,
                             int64_t* h_n_unknown, int* h_done, void* stream) {
  if (!d_workspace || !h_istop || !h_itn || !h_done) return smrf_fail(SMRF_E_ARG, "null pointer");
  const Band b = band_of(const_cast<void*>(d_workspace), rows_local, cols, padded_pitch(cols), 0, 0);
  Sc out{};
  SMRF_HIP_CHECK(hipMemcpyAsync(&out, b.sc, sizeof(out), hipMemcpyDeviceToHost, (hipStream_t)stream));
  SMRF_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  *h_istop = out.istop;
  *h_itn = (int64_t)out.itn;
  *h_done = out.done || out.istop != 0 || out.itn >= out.iter_lim;
  if (h_n_unknown) *h_n_unknown = (int64_t)out.nunk;
  return SMRF_OK;
}

}  // extern "C"
