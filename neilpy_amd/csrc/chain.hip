// Instantiates the chained small-window kernels of morph_chain.h and matches window lists against them.
#include <cmath>

#include "morph_chain.h"

namespace {

struct Pattern { int n; int r[4]; long long min_cells; long long min_cells_f64; };   // min_cells < 0: no kernel at that dtype
// The launches that exist, in the order they are tried (NP row pairs per batch - SMRF_CHAIN_NP - and the occupancy a kernel
// is built for are per pattern, below).  A single window is a chain of one: the same table-free stages, which up to
// R = 10 beat the table-building fused kernel of morph_fused.h (and the two ring passes of R = 9): at R <= 6 they run at the
// device's copy rate (0.55 ms for the 10 B/cell of a 16384^2 fp32 window).  min_cells: the smallest raster a pattern is
// taken for (a chain's segments start sum(2R) rows early and its strips lose sum(2R) columns per side).
// Measured on 16384^2 fp32 against round 2's one fused launch (two ring passes at R = 9) per window, ms
// (profiles/r03_chain_windows.md): 1, 2, 3: 0.79 against 1.96; 4, 5: 0.84 against 1.42 (on 4096^2 two single launches
// win: 0.103 against 0.114); 6: 0.55 against 0.76; 7: 0.67 against 0.79; 8: 0.65 against 0.84; 9: 0.72 against 1.07;
// 10: 0.79 against 0.99; a chain 6, 7 takes 1.31 (two singles 1.22; round 4, built for 3 waves per SIMD - 154-158 registers,
// no scratch: 1.28 against 1.18, profiles/r04_logs/chain_6_7_ab.log), a chain 8, 9 (172 registers, two workgroups per CU)
// 3.1 against 2.0: neither exists.
// Round 4 (grouped neighbour reads, chain_stage_grouped; profiles/r04_chain_grouped.md): fp64 singles exist at R = 4, 5, 7, 8
// (8192^2: 0.279 against the fused opening's 0.374 ms at R = 4, 0.308 / 0.330 at 5, 0.425 against two ring passes' 0.549 at 7,
// 0.461 / 0.531 at 8; R = 6 loses to the fused kernel by 7 %, R = 9, 10 - 174-186 registers, two waves per SIMD - to the ring
// passes by 14-20 %; on 4096^2 only R = 4 and 7 still win).  The fp32 singles R = 11..14 in the grouped form (139-166
// registers, 3 waves per SIMD) measured 12-16 % SLOWER than the fused kernels (0.99 / 1.01 / 1.08 / 1.14 against 0.85 / 0.89 /
// 0.96 / 1.02 ms on 16384^2): the cell-by-cell window growth costs R min / max per row and stage where the table costs
// K - 1 + ~3, and from R = 11 that outweighs the table's two extra barriers.  They do not exist.
// The fp64 chain 1, 2, 3 (134 registers at one row pair per batch, 3 waves per SIMD): 0.578 against 0.615 ms for chain 1, 2 + the
// fused R = 3 on 8192^2, slower on 4096^2 and 1024^2 (profiles/r04_logs/chain_123_f64_ab.log): from 48 Mi cells in round 4.
// Round 5: the thresholds below were measured again after the launches' segmentation changed (seg_rule.h: one round cut by the
// cost model on rasters this small; profiles/r05_logs/segments/min_cells_f32.log, min_cells_f64.log: default routing against
// every kind that exists on 1024^2 ... 6000^2).  fp32: the chain 4, 5 wins from 5000^2 (-6 %, 6000^2 -11 %; loses 7-19 % on
// 2048^2 and 4096^2), the singles R = 9, 10 win 17-27 % on 5000^2 and 6000^2; R = 9 also wins 9-13 % on 1024^2 ... 3000^2 and ties
// on 4096^2 (any size now), R = 10 ties below 5000^2.  fp64: the chain
// 1, 2, 3 and the single R = 5 win on every raster tried (-4 ... -16 %), R = 7 from 2048^2 (-7 %), R = 8 from 4096^2
// (-11 ... -21 %; +10 ... +14 % below).
constexpr long long kLarge = 20ll << 20;
constexpr long long kMid = 16ll << 20;
constexpr long long kSmall = 4ll << 20;
constexpr long long kNever = -1;
constexpr Pattern kPatterns[] = {{3, {1, 2, 3, 0}, 0, 0}, {2, {1, 2, 0, 0}, 0, 0}, {2, {2, 3, 0, 0}, 0, 0}, {2, {4, 5, 0, 0}, kLarge, kNever},
                                 {1, {4, 0, 0, 0}, 0, 0}, {1, {5, 0, 0, 0}, 0, 0}, {1, {6, 0, 0, 0}, 0, kNever}, {1, {7, 0, 0, 0}, 0, kSmall},
                                 {1, {8, 0, 0, 0}, 0, kMid}, {1, {9, 0, 0, 0}, 0, kNever}, {1, {10, 0, 0, 0}, kLarge, kNever}};
constexpr int kNPatterns = (int)(sizeof(kPatterns) / sizeof(kPatterns[0]));

#ifndef SMRF_CHAIN_OCC
#define SMRF_CHAIN_OCC 4       // waves per SIMD the chain kernels are built for (tuning builds override)
#endif
// per pattern (fp32 singles; tuning builds override): waves per SIMD a kernel is built for
// (round 4: the single R = 7, pattern 7, fits 5 waves per SIMD - 0.641 -> 0.583 ms on 16384^2; at 5 the singles from R = 8 up and the chain 4, 5
// spill, at 3 R = 9, 10 gained on one box and lost on the next: profiles/r04_logs/chain_np_occ_ab.log)
#ifndef SMRF_CHAIN_OCC_OF
#define SMRF_CHAIN_OCC_OF(PAT) ((PAT) == 7 ? 5 : SMRF_CHAIN_OCC)
#endif
// the grouped fp64 singles hold 112-160 registers: built for 3 waves per SIMD
template <typename T>
int launch(int pat, const ChainArgs<T>& a_in, hipStream_t s) {
  ChainArgs<T> a = a_in;
  for (int i = 0; i < 4; ++i) {                            // largest float <= thr (morph_chain.h, flag step)
    a.thr_lo[i] = smrf_float_below(a.thr[i]);
  }
  constexpr bool F32 = sizeof(T) == 4;
  // a single window of radius R: fp32 as round 3 built them; F64OK: the grouped fp64 form exists (NP = 1, 3 waves per SIMD)
#define SMRF_SINGLE(PAT, R, F64OK)                                                                                        \
    case PAT:                                                                                                            \
      if constexpr (F32) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, PAT), SMRF_CHAIN_OCC_OF(PAT), R, 0, 0, 0>(a, s);   \
      else if constexpr (F64OK) return smrf::chain_launch<T, 1, 3, R, 0, 0, 0>(a, s);                                    \
      else break;
  switch (pat) {
    case 0:   // fp64: one row pair per batch, built for 3 waves per SIMD (134 registers; at 4 it spills)
      if constexpr (F32) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 0), SMRF_CHAIN_OCC, 1, 2, 3, 0>(a, s);
      else return smrf::chain_launch<T, 1, 3, 1, 2, 3, 0>(a, s);
    case 1: return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 1), SMRF_CHAIN_OCC, 1, 2, 0, 0>(a, s);
    case 2: return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 2), SMRF_CHAIN_OCC, 2, 3, 0, 0>(a, s);
    case 3: if constexpr (F32) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 3), SMRF_CHAIN_OCC, 4, 5, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    SMRF_SINGLE(4, 4, true) SMRF_SINGLE(5, 5, true) SMRF_SINGLE(6, 6, false) SMRF_SINGLE(7, 7, true) SMRF_SINGLE(8, 8, true)
    SMRF_SINGLE(9, 9, false) SMRF_SINGLE(10, 10, false)
    default: break;
  }
#undef SMRF_SINGLE
  return smrf_fail(SMRF_E_ARG, "no chain kernel for pattern %d at this dtype", pat);
}

}  // namespace

int smrf_chain_match(int elem_size, const int32_t* windows, int n, long long cells) {
  for (int p = 0; p < kNPatterns; ++p) {
    const long long mc = elem_size == 8 ? kPatterns[p].min_cells_f64 : kPatterns[p].min_cells;
    if (kPatterns[p].n > n || mc < 0 || cells < mc) continue;
    bool ok = true;
    for (int i = 0; i < kPatterns[p].n; ++i) ok = ok && windows[i] == kPatterns[p].r[i];
    if (ok) return p;
  }
  return -1;
}
int smrf_chain_length(int pat) { return pat >= 0 && pat < kNPatterns ? kPatterns[pat].n : 0; }
int smrf_chain_halo(int pat) {
  int s = 0;
  if (pat >= 0 && pat < kNPatterns)
    for (int i = 0; i < kPatterns[pat].n; ++i) s += 2 * kPatterns[pat].r[i];
  return s;
}
int smrf_chain_f32(int pat, const ChainArgs<float>& a, hipStream_t s) { return launch<float>(pat, a, s); }
int smrf_chain_f64(int pat, const ChainArgs<double>& a, hipStream_t s) { return launch<double>(pat, a, s); }
