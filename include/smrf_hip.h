/*
 * smrf_hip.h -- C ABI of libsmrf_hip.so: the MI355X (gfx950) implementation of neilpy's
 * SMRF bare-earth path.
 *
 * The reference (thomaspingel/neilpy) is a pure-Python function library with no FFI of its
 * own; the interfaces this library replaces are therefore the Python callables of the path
 * and the third-party primitives they dispatch to.  Each entry point cites what it replaces
 * (paths relative to the reference checkout):
 *
 *   smrf_disk_filter_*        skimage.morphology.erosion/dilation with disk(r) as reached by
 *                             opening(last_surface, disk(window)), neilpy/neilpy.py:1670
 *                             (-> scipy.ndimage.grey_erosion/grey_dilation, mode='reflect')
 *   smrf_progressive_filter_* progressive_filter(), neilpy/neilpy.py:1659-1680
 *   smrf_points_extent_f64,
 *   smrf_grid_*               create_dem(), neilpy/neilpy.py:1110-1166 (pandas groupby min/max
 *                             at :1151-1156, affine index arithmetic at :1141-1143)
 *   smrf_points_band_count_f64, smrf_points_band_pack_f64
 *                             the same index arithmetic (:1141-1143) used to route points to the rank that owns
 *                             their raster row when the cloud is sharded over GPUs (no counterpart in the reference)
 *   smrf_springs_lsqr_f64     inpaint_nans_by_springs(), neilpy/neilpy.py:1227-1271, whose
 *                             solve is scipy.sparse.linalg.lsqr (:1264)
 *   smrf_fda_lsqr_f64         inpaint_nans_by_fda(), neilpy/neilpy.py:1170-1216
 *   smrf_fda_apply_f64        (diagnostic) the same operator applied once, for the structural parity test
 *   smrf_gradient_slope_f64   np.gradient + sqrt, neilpy/neilpy.py:1785-1786
 *   smrf_pssm_f64             pssm(), neilpy/neilpy.py:846-867
 *   smrf_las_decode_xyz_f64   the coordinate decode of read_las(), neilpy/neilpy.py:903-1087 (host
 *                             side: neilpy_amd/las.py)
 *   smrf_spline_solve_f64, smrf_spline_solve_ws_f64, smrf_spline_eval_f64, smrf_classify_points_f64
 *                             RectBivariateSpline(...).ev and the point test, neilpy/neilpy.py:1768-1795
 *   smrf_negate_f64, smrf_mask_apply_f64
 *                             the elementwise glue of smrf(), neilpy/neilpy.py:1744, :1748, :1762-1763
 *
 * Conventions
 *   - every pointer named d_* is DEVICE memory (hipMalloc or a torch CUDA tensor's data_ptr);
 *     h_* is host memory.  The library never allocates or frees caller-visible memory; scratch
 *     comes from the caller through (workspace, workspace_bytes).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Unless stated, calls
 *     only enqueue work on `stream` and return without synchronising.
 *   - return value: 0 on success, a negative SMRF_E_* code on failure; smrf_last_error() gives
 *     a human-readable message for the calling thread.  No exceptions cross the boundary.
 *   - rasters are row-major, row 0 = north, `ld` = elements between consecutive rows.
 *   - there is no CPU fallback: without a HIP device every compute entry returns SMRF_E_HIP.
 */
#ifndef SMRF_HIP_H
#define SMRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMRF_ABI_VERSION 1

#if defined(__GNUC__)
#define SMRF_API __attribute__((visibility("default")))
#else
#define SMRF_API
#endif

#define SMRF_OK 0
#define SMRF_E_ARG (-1)       /* invalid argument (null pointer, bad size, band does not cover halo) */
#define SMRF_E_HIP (-2)       /* a HIP runtime call failed, or no device */
#define SMRF_E_WORKSPACE (-3) /* workspace too small */
#define SMRF_E_RANGE (-4)     /* a point falls outside the grid (create_dem with edges=) */
#define SMRF_E_UNSUPPORTED (-5)

/* implementation selector for the disk filter (tests exercise both; 0 is the product default) */
#define SMRF_IMPL_AUTO 0
#define SMRF_IMPL_RING 1   /* register-ring kernels, radius 1..SMRF_RING_MAX_RADIUS */
#define SMRF_IMPL_DIRECT 2 /* footprint-gather kernel, any radius */
#define SMRF_RING_MAX_RADIUS 64

SMRF_API int smrf_abi_version(void);
SMRF_API const char* smrf_last_error(void);
/* number of visible HIP devices (0 when there is none); never fails */
SMRF_API int smrf_device_count(void);
/* (diagnostic) The SMRF_* environment switches (SMRF_FUSED, SMRF_CHAIN, SMRF_NT, SMRF_RING_DUAL, SMRF_RING_SEG, ... - developer
 * A/B runs and the parity tests that force every launch variant on small rasters) are read ONCE, when the library first needs
 * them; no launch path calls getenv.  A process that changes them afterwards calls this to have them read again.  Not
 * thread-safe against concurrent launches. */
SMRF_API void smrf_switches_reload(void);

/* ------------------------------------------------------------------------------------------
 * Grey erosion / dilation by skimage's disk(radius) with scipy's mode='reflect' borders.
 *
 * The raster has `img_rows` x `cols` cells.  `d_in` holds global rows
 * [in_row0, in_row0+in_rows), `d_out` receives global rows [out_row0, out_row0+out_rows);
 * both pointers address the first row they hold.  Single GPU: in_row0 = out_row0 = 0 and
 * in_rows = out_rows = img_rows.  Row-band sharding: every row reflect(y) for
 * y in [out_row0-radius, out_row0+out_rows+radius) must lie inside the input band
 * (otherwise SMRF_E_ARG).
 *
 * nan_aware != 0 reproduces scipy's NaN rule (the result is NaN iff the first visited
 * footprint element, offset (-radius, 0), is NaN; other NaNs are ignored); 0 assumes the input
 * has no NaN and skips the extra read.
 * ------------------------------------------------------------------------------------------ */
SMRF_API int smrf_disk_filter_f32(const float* d_in, float* d_out, int img_rows, int cols, int64_t ld,
                         int in_row0, int in_rows, int out_row0, int out_rows, int radius,
                         int is_dilate, int nan_aware, int impl, void* stream);
SMRF_API int smrf_disk_filter_f64(const double* d_in, double* d_out, int img_rows, int cols, int64_t ld,
                         int in_row0, int in_rows, int out_row0, int out_rows, int radius,
                         int is_dilate, int nan_aware, int impl, void* stream);

/* One progressive_filter step after the erosion: opened = dilate(eroded, disk(radius)), then
 *   new_obj = (double)(last - opened) > threshold   (subtraction in the raster dtype,
 *   comparison in float64: NumPy-2 promotion of neilpy.py:1671)
 *   mask |= new_obj;  when_dropped[new_obj] = window_index   (d_when_dropped may be NULL)
 * d_last, d_opened, d_mask, d_when_dropped address global row out_row0. */
SMRF_API int smrf_pf_dilate_flag_f32(const float* d_eroded, const float* d_last, float* d_opened,
                            uint8_t* d_mask, uint8_t* d_when_dropped, double threshold,
                            int window_index, int img_rows, int cols, int64_t ld, int in_row0,
                            int in_rows, int out_row0, int out_rows, int radius, int nan_aware,
                            int impl, void* stream);
SMRF_API int smrf_pf_dilate_flag_f64(const double* d_eroded, const double* d_last, double* d_opened,
                            uint8_t* d_mask, uint8_t* d_when_dropped, double threshold,
                            int window_index, int img_rows, int cols, int64_t ld, int in_row0,
                            int in_rows, int out_row0, int out_rows, int radius, int nan_aware,
                            int impl, void* stream);

/* One progressive_filter window in ONE launch for the small disks (csrc/morph_fused.h): opened =
 * dilate(erode(last, disk(radius)), disk(radius)) and the flag step of smrf_pf_dilate_flag_*, the eroded
 * surface never leaving the CU.  Row-band form as above, with the band of `last` reaching 2*radius rows
 * (reflected at the raster's true borders) beyond the output rows: d_last holds global rows in_row0 ..
 * in_row0 + in_rows - 1; d_opened, d_mask, d_when_dropped address global row out_row0; the same `last` values
 * are used for the flag comparison.  d_mask may be NULL (opening only).  No NaN rule: the caller must not hand
 * it rasters with NaNs.  Returns SMRF_E_UNSUPPORTED for radii without a fused kernel
 * (smrf_fused_open_supported tells which: fp32 1..8 and 10..14, fp64 1..6). */
SMRF_API int smrf_fused_open_supported(int elem_size, int radius);
SMRF_API int smrf_pf_open_flag_f32(const float* d_last, float* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                          double threshold, int window_index, int img_rows, int cols, int64_t ld, int in_row0,
                          int in_rows, int out_row0, int out_rows, int radius, void* stream);
SMRF_API int smrf_pf_open_flag_f64(const double* d_last, double* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                          double threshold, int window_index, int img_rows, int cols, int64_t ld, int in_row0,
                          int in_rows, int out_row0, int out_rows, int radius, void* stream);

/* Several CONSECUTIVE progressive_filter windows with small disks in ONE launch (csrc/morph_chain.h): opens `last` with
 * disk(h_radii[0]), flags, opens the result with disk(h_radii[1]), flags, ...; only the LAST opened surface is written
 * (d_opened), every window's flags go to d_mask / d_when_dropped (window i writes h_window_index[i]) exactly as a sequence of
 * smrf_pf_open_flag_* calls would.  Replaces n iterations of the loop of neilpy/neilpy.py:1667-1676.  Row-band form: the band
 * of `last` must reach sum(2 * radius) rows (reflected at the raster's true borders) beyond the output rows; flags are
 * written for the output rows only.  No NaN rule.  smrf_pf_chain_length: how many of the windows at the head of h_radii one
 * launch takes on a raster of raster_cells cells (0: none; e.g. 3 for 1, 2, 3, ...; 1 = a table-free single launch);
 * smrf_pf_chain_flag_* returns SMRF_E_UNSUPPORTED unless n_windows is a length this function reports for those radii
 * at some raster size. */
SMRF_API int smrf_pf_chain_length(int elem_size, const int32_t* h_radii, int n, int64_t raster_cells);
SMRF_API int smrf_pf_chain_flag_f32(const float* d_last, float* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                           const int32_t* h_radii, const double* h_thresholds, const int32_t* h_window_index,
                           int n_windows, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                           int out_rows, void* stream);
SMRF_API int smrf_pf_chain_flag_f64(const double* d_last, double* d_opened, uint8_t* d_mask, uint8_t* d_when_dropped,
                           const int32_t* h_radii, const double* h_thresholds, const int32_t* h_window_index,
                           int n_windows, int img_rows, int cols, int64_t ld, int in_row0, int in_rows, int out_row0,
                           int out_rows, void* stream);

/* Whole progressive_filter on one device.  d_Z is read only.  h_windows / h_thresholds are HOST
 * arrays of n_windows entries; thresholds are slope_threshold*(windows*cellsize) evaluated by
 * the caller in float64 (neilpy.py:1661).  d_mask (rows*cols bytes, 0/1) and d_when_dropped
 * (may be NULL) are overwritten.  Workspace: smrf_progressive_filter_workspace_bytes().
 * nan_aware < 0: the library counts NaNs itself (one small synchronising readback). */
SMRF_API size_t smrf_progressive_filter_workspace_bytes(int rows, int cols, int elem_size);
SMRF_API int smrf_progressive_filter_f32(const float* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask,
                                uint8_t* d_when_dropped, void* d_workspace,
                                size_t workspace_bytes, int nan_aware, int impl, void* stream);
SMRF_API int smrf_progressive_filter_f64(const double* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask,
                                uint8_t* d_when_dropped, void* d_workspace,
                                size_t workspace_bytes, int nan_aware, int impl, void* stream);

/* Measurement form of the same call (bench.py's per-class roofline, tools/): an event is recorded on `stream` at
 * every window boundary and, after the last window has finished (the call synchronises), h_window_ms[i] holds the
 * device time of window i and h_window_route[i] (may be NULL) how it ran: SMRF_ROUTE_*.  Same result, same routing
 * as smrf_progressive_filter_*; no reference counterpart (the reference has no timers). */
#define SMRF_ROUTE_TWO_PASS 0   /* ring erosion, then ring dilation + flag: 5s + 2 B/cell (s = sizeof element) */
#define SMRF_ROUTE_FUSED 1      /* one fused opening + flag launch: 2s + 2 B/cell */
#define SMRF_ROUTE_DIRECT 2     /* footprint-gather kernels, two passes (radius > SMRF_RING_MAX_RADIUS or impl = direct) */
#define SMRF_ROUTE_COPY 3       /* radius 0 */
#define SMRF_ROUTE_CHAIN 4      /* table-free launch of morph_chain.h: route = 4 + position in a chain of windows opened in one launch (its
                                 * time is reported on position 0, the others read ~0); a single window is a chain of one */
SMRF_API int smrf_progressive_filter_timed_f32(const float* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask,
                                uint8_t* d_when_dropped, void* d_workspace, size_t workspace_bytes, int nan_aware,
                                int impl, void* stream, float* h_window_ms, int32_t* h_window_route);
SMRF_API int smrf_progressive_filter_timed_f64(const double* d_Z, int rows, int cols, const int32_t* h_windows,
                                const double* h_thresholds, int n_windows, uint8_t* d_mask,
                                uint8_t* d_when_dropped, void* d_workspace, size_t workspace_bytes, int nan_aware,
                                int impl, void* stream, float* h_window_ms, int32_t* h_window_route);

/* number of NaN cells of a contiguous array, written to *h_count (synchronises `stream`) */
SMRF_API int smrf_count_nan_f32(const float* d_a, int64_t n, int64_t* h_count, void* stream);
SMRF_API int smrf_count_nan_f64(const double* d_a, int64_t n, int64_t* h_count, void* stream);

/* ------------------------------------------------------------------------------------------
 * create_dem
 * ------------------------------------------------------------------------------------------ */
/* h_out[4] = min x, max x, min y, max y over n points (synchronises `stream`).
 * workspace: 4 * 1024 doubles. */
SMRF_API int smrf_points_extent_f64(const double* d_x, const double* d_y, int64_t n, double* h_out,
                           void* d_workspace, size_t workspace_bytes, void* stream);
/* LAS point records (packed, record_length bytes each, x/y/z int32 first) -> float64 coordinates
 * x = X * scale + offset (neilpy.py:1055-1057 in read_las); h_scale_offset = {sx, sy, sz, ox, oy, oz} */
SMRF_API int smrf_las_decode_xyz_f64(const uint8_t* d_records, int64_t npts, int record_length,
                            const double* h_scale_offset, double* d_x, double* d_y, double* d_z,
                            void* stream);
/* fractional pixel coordinates (col, row) = ~t * (x, y) with h_inv = (a,b,c,d,e,f) of the inverse
 * transform, products and sums rounded separately like the affine package (neilpy.py:1772) */
SMRF_API int smrf_affine_apply_f64(const double* d_x, const double* d_y, int64_t npts, const double* h_inv,
                          double* d_col, double* d_row, void* stream);
/* keys: one uint64 per cell; all-ones = empty */
SMRF_API int smrf_grid_clear_u64(uint64_t* d_keys, int64_t ncells, void* stream);
/* col = floor(x*inv[0] + y*inv[1] + inv[2]), row = floor(x*inv[3] + y*inv[4] + inv[5]) with
 * separately rounded multiplies and adds (no FMA), inv = coefficients (a,b,c,d,e,f) of the
 * inverse affine transform.  Points with NaN z are skipped (pandas min/max skip NaN).
 * h_filter (may be NULL) = {xedges[0], xedges[-1], yedges[-1], yedges[0]}: points outside are
 * dropped first (neilpy.py:1128).  Rows [row0, row0+rows_local) of the rows_total x cols grid
 * are held in d_keys (row-band sharding; single GPU: row0 = 0, rows_local = rows_total);
 * points of other bands are ignored, points outside the whole grid are counted in
 * *d_n_outside (device int64, caller zeroes it). */
SMRF_API int smrf_grid_bin_f64(const double* d_x, const double* d_y, const double* d_z, int64_t npts,
                      const double* h_inv, const double* h_filter, uint64_t* d_keys,
                      int rows_total, int cols, int row0, int rows_local, int is_max,
                      int64_t* d_n_outside, void* stream);
/* keys -> float64 grid, empty -> NaN; d_empty (may be NULL) gets 1 where empty */
SMRF_API int smrf_grid_finalize_f64(const uint64_t* d_keys, double* d_grid, uint8_t* d_empty,
                           int64_t ncells, int is_max, void* stream);

/* Sharded create_dem without replicated points (the all-to-all form; create_dem semantics of
 * neilpy/neilpy.py:1141-1156 are unchanged).  Every rank holds 1/N of the points; each point belongs to the
 * rank whose row band holds floor(row) of `~t * (x, y)` (h_inv = the six inverse-affine coefficients, the same
 * rounding as smrf_grid_bin_f64); bands are sharded.band_rows(): the first rows_total % nbands bands hold one
 * row more.  nbands <= 64.
 *   smrf_points_band_count_f64: d_counts[nbands] <- points of this rank per destination band.
 *   smrf_points_band_pack_f64:  points copied into d_out_{x,y,z} grouped by destination band; d_cursors[nbands]
 *     holds each band's first output index on entry (exclusive prefix sum of the counts) and its end on return.
 *     Order inside a band's run is arbitrary (the binning is order independent).
 * The host exchanges the runs (counts first) with all_to_all over RCCL and bins what it receives with
 * smrf_grid_bin_f64(row0, rows_local). */
SMRF_API int smrf_points_band_count_f64(const double* d_x, const double* d_y, int64_t npts, const double* h_inv,
                               int rows_total, int nbands, uint64_t* d_counts, void* stream);
SMRF_API int smrf_points_band_pack_f64(const double* d_x, const double* d_y, const double* d_z, int64_t npts,
                              const double* h_inv, int rows_total, int nbands, uint64_t* d_cursors,
                              double* d_out_x, double* d_out_y, double* d_out_z, void* stream);

/* ------------------------------------------------------------------------------------------
 * inpaint_nans_by_springs: matrix-free LSQR on the raster's edge planes, scipy's recurrence
 * and stopping rule (damp = 0).  d_A (rows x cols, contiguous, float64) is updated in place:
 * NaN cells receive the solution, known cells are untouched.  iter_lim < 0 -> 2*n_unknown.
 * Synchronous: returns after the solve; *h_istop, *h_itn as scipy reports them.
 * ------------------------------------------------------------------------------------------ */
SMRF_API size_t smrf_springs_workspace_bytes(int rows, int cols);
SMRF_API int smrf_springs_lsqr_f64(double* d_A, int rows, int cols, double atol, double btol,
                          double conlim, int64_t iter_lim, int* h_istop, int64_t* h_itn,
                          int64_t* h_n_unknown, void* d_workspace, size_t workspace_bytes,
                          void* stream);

/* Row-band form of the same solver for rasters sharded over several GPUs.  The band
 * holds rows_local x cols cells of d_A_band.  The workspace keeps every plane with one halo row
 * above and one below; smrf_springs_band_layout() gives the byte offsets the host needs to
 * exchange halo rows and to all-reduce the phase sums:
 *   h_out[0] v plane, h_out[1] uv plane (rows_local + 2 rows, h_out[6] doubles apart, the first cols used; row 0 = halo above),
 *   h_out[2] hole plane (rows_local + 2 rows, h_out[6] bytes apart), h_out[6] the planes' row pitch in cells (>= cols),
 *   h_out[3] cols doubles = raster row
 *   below the band, h_out[4] two doubles = the phase's local sums (phase 7 fills both: |v|^2 and |dk|^2,
 *   to be all-reduced as ONE 2-element buffer; every other phase uses the first), h_out[5] total bytes.
 * Phases (in order; "<- X" = what the host must have delivered before the phase):
 *   0 mask+count | 1 rhs <- hole halo below, A row below, all-reduced count | 2 |b| <- all-reduce
 *   3 first v = S^T u <- uv halo above | 4 first alfa <- all-reduce
 *   loop (the split rotation of lsqr.py:424-555, csrc/lsqr_core.h; the single-device solver's kernels):
 *         5 u = S v - alfa u <- v halo below | 6 beta, rho, t1 <- all-reduce
 *         7 w, dk, (x every second iteration), v = S^T u - beta v <- uv halo above
 *         8 alfa, rest of the rotation, stopping tests <- all-reduce (2 values)
 *   10 scatter the solution into d_A_band (adds the pending x step when the solve stopped at an odd iteration).
 * Two halo rows and two all-reduces per iteration, 12 plane touches (round 4's phases 5, 6, 3, 7, 8, 9: 15).
 * neilpy_amd/sharded.py drives it over torch.distributed (RCCL). */
SMRF_API size_t smrf_springs_band_workspace_bytes(int rows_local, int cols);
SMRF_API int smrf_springs_band_layout(int rows_local, int cols, int64_t* h_out);
SMRF_API int smrf_springs_band_begin(int rows_local, int cols, double atol, double btol, double conlim,
                            int64_t iter_lim, void* d_workspace, size_t workspace_bytes,
                            void* stream);
SMRF_API int smrf_springs_band_phase(int phase, double* d_A_band, int rows_local, int cols, int has_above,
                            int has_below, void* d_workspace, size_t workspace_bytes, void* stream);
SMRF_API int smrf_springs_band_status(const void* d_workspace, int rows_local, int cols, int* h_istop,
                             int64_t* h_itn, int64_t* h_n_unknown, int* h_done, void* stream);

/* inpaint_nans_by_fda(A), neilpy/neilpy.py:1170-1216: the NaN cells of d_A (rows x cols, float64,
 * C order) are replaced in place by scipy.sparse.linalg.lsqr's iterate on the second-difference
 * equations that touch a NaN cell, every equation weighted by its number of NaN stencil cells
 * as the reference's row selection does (:1207-1210).  Arguments as smrf_springs_lsqr_f64.
 * Single device. */
SMRF_API size_t smrf_fda_workspace_bytes(int rows, int cols);
SMRF_API int smrf_fda_lsqr_f64(double* d_A, int rows, int cols, double atol, double btol, double conlim,
                      int64_t iter_lim, int* h_istop, int64_t* h_itn, int64_t* h_n_unknown,
                      void* d_workspace, size_t workspace_bytes, void* stream);
/* Diagnostic (tests): one application of the operator smrf_fda_lsqr_f64 iterates with, against the
 * reference's explicit sparse system (neilpy/neilpy.py:1180-1209).  All buffers are rows x cols rasters on
 * the device: d_rhs / d_cnt <- right-hand side and multiplicity (NaN entries, :1207-1209) of every
 * equation cell; d_Av <- A v on the equation cells (v read on the NaN cells of d_A); d_Atu <- A^T u on the
 * NaN cells (u read on the equation cells, each counted d_cnt times).  Workspace as smrf_fda_workspace_bytes. */
SMRF_API int smrf_fda_apply_f64(const double* d_A, int rows, int cols, const double* d_v, const double* d_u,
                      double* d_rhs, uint8_t* d_cnt, double* d_Av, double* d_Atu, void* d_workspace,
                      size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * smrf tail
 * ------------------------------------------------------------------------------------------ */
/* S = sqrt(gy^2 + gx^2), (gy, gx) = np.gradient(Z, cellsize) (second-order interior,
 * first-order one-sided edges).  rows, cols >= 2. */
SMRF_API int smrf_gradient_slope_f64(const double* d_Z, double* d_S, int rows, int cols, double cellsize,
                            void* stream);

/* pssm(), neilpy/neilpy.py:846-867 (the bonemaps of examples/smrf/ *.ipynb): slope as above, then
 * P = uint8(round(255 * (rad2deg(arctan(ve * S)) / 90))) into d_P (rows x cols, nullable) and, when
 * d_rgba is given, the colormap lookup d_rgba[i][0..3] = d_lut[P[i]][0..3] (d_lut: 256 x 4 doubles,
 * what plt.cm.bone_r / plt.cm.bone hold; d_rgba: rows x cols x 4 doubles). */
SMRF_API int smrf_pssm_f64(const double* d_Z, uint8_t* d_P, double* d_rgba, const double* d_lut, int rows, int cols,
                  double cellsize, double ve, void* stream);

/* Interpolating bicubic spline of scipy.interpolate.RectBivariateSpline(rows, cols, Z) with its
 * defaults kx = ky = 3, s = 0 (neilpy.py:1773, :1788): d_C holds the raster on entry and the
 * B-spline coefficients on return.  d_lu_rows / d_lu_cols: banded LU factors (5 x rows, 5 x cols
 * doubles: l2, l1, d, u1, u2) of the per-axis collocation systems; neilpy_amd/spline.py builds them. */
SMRF_API int smrf_spline_solve_f64(double* d_C, int rows, int cols, const double* d_lu_rows,
                          const double* d_lu_cols, void* stream);
/* The same solve with a scratch plane of rows x cols doubles, which lets every line be cut into chunks that run side by
 * side (each chunk re-derives its start state from 64 entries of warm-up; the substitutions contract by 0.268 per entry,
 * so nothing of the cut is left in float64).  d_C: raster in, coefficients out; d_scratch: clobbered.  This is the entry
 * neilpy_amd.smrf uses; the one above is the line-by-line sequential form the tests compare it with. */
SMRF_API int smrf_spline_solve_ws_f64(double* d_C, double* d_scratch, int rows, int cols, const double* d_lu_rows,
                             const double* d_lu_cols, void* stream);
/* .ev(px, py) of that spline (FITPACK bispeu): px runs along the rows axis, py along the columns
 * axis; d_tx (rows + 4) and d_ty (cols + 4) are the knots; arguments are clamped to the knot range. */
SMRF_API int smrf_spline_eval_f64(const double* d_C, int rows, int cols, const double* d_tx, const double* d_ty,
                         const double* d_px, const double* d_py, int64_t npts, double* d_out,
                         void* stream);
/* is_object_point = abs(elev - z) > elevation_threshold + elevation_scaler * slope (neilpy.py:1794-1795) */
SMRF_API int smrf_classify_points_f64(const double* d_elev, const double* d_slope, const double* d_z,
                             int64_t npts, double elevation_threshold, double elevation_scaler,
                             uint8_t* d_is_object, void* stream);

/* out = -in (smrf runs the low-outlier filter on -Zmin, neilpy.py:1744) */
SMRF_API int smrf_negate_f64(const double* d_in, double* d_out, int64_t n, void* stream);
/* u = a | b | c (b, c, d_union may be NULL); Z[u] = NaN  (neilpy.py:1748 and :1762-1763) */
SMRF_API int smrf_mask_apply_f64(double* d_Z, const uint8_t* d_a, const uint8_t* d_b, const uint8_t* d_c,
                        uint8_t* d_union, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SMRF_HIP_H */
