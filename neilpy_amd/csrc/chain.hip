// Instantiates the chained small-window kernels of morph_chain.h and matches window lists against them.
#include <cmath>

#include "morph_chain.h"

namespace {

struct Pattern { int n; int r[4]; long long min_cells; };
// The launches that exist, in the order they are tried (NP row pairs per batch - SMRF_CHAIN_NP - and the occupancy a kernel
// is built for are per pattern, below).  A single window is a chain of one: the same table-free stages, which up to
// R = 10 beat the table-building fused kernel of morph_fused.h (and the two ring passes of R = 9): at R <= 6 they run at the
// device's copy rate (0.55 ms for the 10 B/cell of a 16384^2 fp32 window).  min_cells: the smallest raster a pattern is
// taken for (a chain's segments start sum(2R) rows early and its strips lose sum(2R) columns per side).
// Measured on 16384^2 fp32 against round 2's one fused launch (two ring passes at R = 9) per window, ms
// (profiles/r03_chain_windows.md): 1, 2, 3: 0.79 against 1.96; 4, 5: 0.84 against 1.42 (on 4096^2 two single launches
// win: 0.103 against 0.114); 6: 0.55 against 0.76; 7: 0.67 against 0.79; 8: 0.65 against 0.84; 9: 0.72 against 1.07;
// 10: 0.79 against 0.99; 11..14 lose (1.05 against 0.86 at 11); a chain 6, 7 takes 1.31 (two singles 1.22), a chain 8, 9
// (172 registers, two workgroups per CU) 3.1 against 2.0: neither exists.
constexpr long long kLarge = 48ll << 20;
constexpr Pattern kPatterns[] = {{3, {1, 2, 3, 0}, 0}, {2, {1, 2, 0, 0}, 0}, {2, {2, 3, 0, 0}, 0}, {2, {4, 5, 0, 0}, kLarge},
                                 {1, {4, 0, 0, 0}, 0}, {1, {5, 0, 0, 0}, 0}, {1, {6, 0, 0, 0}, 0}, {1, {7, 0, 0, 0}, 0},
                                 {1, {8, 0, 0, 0}, 0}, {1, {9, 0, 0, 0}, kLarge}, {1, {10, 0, 0, 0}, kLarge}};
constexpr int kNPatterns = (int)(sizeof(kPatterns) / sizeof(kPatterns[0]));

#ifndef SMRF_CHAIN_OCC
#define SMRF_CHAIN_OCC 4       // waves per SIMD the chain kernels are built for (tuning builds override)
#endif

template <typename T>
int launch(int pat, const ChainArgs<T>& a_in, hipStream_t s) {
  ChainArgs<T> a = a_in;
  for (int i = 0; i < 4; ++i) {                            // largest float <= thr (morph_chain.h, flag step)
    a.thr_lo[i] = smrf_float_below(a.thr[i]);
  }
  switch (pat) {
    case 0: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 0), SMRF_CHAIN_OCC, 1, 2, 3, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 1: return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 1), SMRF_CHAIN_OCC, 1, 2, 0, 0>(a, s);
    case 2: return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 2), SMRF_CHAIN_OCC, 2, 3, 0, 0>(a, s);
    case 3: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 3), SMRF_CHAIN_OCC, 4, 5, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 4: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 4), SMRF_CHAIN_OCC, 4, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 5: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 5), SMRF_CHAIN_OCC, 5, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 6: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 6), SMRF_CHAIN_OCC, 6, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 7: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 7), SMRF_CHAIN_OCC, 7, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 8: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 8), SMRF_CHAIN_OCC, 8, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 9: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 9), SMRF_CHAIN_OCC, 9, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    case 10: if constexpr (sizeof(T) == 4) return smrf::chain_launch<T, SMRF_CHAIN_NP(T, 10), SMRF_CHAIN_OCC, 10, 0, 0, 0>(a, s); else break;   // fp32 only: the fp64 form spills
    default: break;
  }
  return smrf_fail(SMRF_E_ARG, "no chain kernel for pattern %d at this dtype", pat);
}

}  // namespace

int smrf_chain_match(int elem_size, const int32_t* windows, int n, long long cells) {
  for (int p = 0; p < kNPatterns; ++p) {
    if (kPatterns[p].n > n || cells < kPatterns[p].min_cells) continue;
    if (elem_size == 8 && p != 1 && p != 2) continue;      // fp64: only the chains whose kernels hold their rings in registers
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
