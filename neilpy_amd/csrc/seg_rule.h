// How a marching launch (ring / fused / chain kernels) is cut into row segments.  Plain C++, no HIP: included by
// smrf_common.h and compiled on its own by tests/test_host_logic.py.
#pragma once

// Rows of segments for a launch of `strips` workgroups per row of segments over `rows` output rows, on 256 CUs that hold
// `resident` workgroups each; every segment marches `warm` extra rows first and is a multiple of `batch` rows.
// Rule 0 (default, round 5): the count that minimises  k x (segment + warm + 20) / X(k),  k = workgroups on the busiest CU,
// X(k) = 1, 1.57, 2.1, 2.37 (+0.1 per further one) = what k resident workgroups of these kernels get out of a CU beside one -
// measured on 1024^2 ... 16384^2 rasters (profiles/r05_segment_balance.md section 4): a lone workgroup runs twice as fast as one of three, so
// a raster too small to give every slot a long segment is better cut so that the workgroup count fills the CUs
// k times exactly (1024^2 windows 15..50: 5.4 -> 2.7 ms, 2048^2: 5.5 -> 3.8, 4096^2 -8 %); for a large raster it is one full
// round as before; a launch that asks for several rounds is taken at its word only when that is cheaper.  Rules 1, 2: one full round of `rounds x resident x 256 / strips` segments, rounded to nearest (rounds
// 1-4; a second, nearly empty round whenever that rounds up) or down, with segments of at least `min_seg` rows.
inline double smrf_cu_rate(int k) {   // what k resident workgroups get out of a CU beside one
  return k <= 1 ? 1.0 : k == 2 ? 1.57 : k == 3 ? 2.1 : 2.37 + 0.1 * (k - 4);
}
inline int smrf_pick_nseg(int rows, int strips, int resident, int rounds, int warm, int batch, int min_seg, int rule) {
  // the full-rounds count of rules 1, 2 (and rule 0's candidate for a launch that asks for several rounds)
  const int full = (rounds * resident * 256 + (rule == 1 ? strips / 2 : 0)) / strips;
  int full_seg = (rows + (full > 1 ? full : 1) - 1) / (full > 1 ? full : 1);
  full_seg = full_seg > min_seg ? full_seg : min_seg;
  full_seg = full_seg < rows ? full_seg : rows;
  const int full_nseg = (rows + full_seg - 1) / full_seg;
  if (rule != 0) return full_nseg;
  int best = 1;
  double best_cost = 0.0;
  for (int k = 1; k <= resident; ++k) {
    int nseg = (int)(((long long)k * 256) / strips);
    if (nseg < 1) continue;
    const int most = rows / batch > 1 ? rows / batch : 1;
    nseg = nseg < most ? nseg : most;
    int seg = (rows + nseg - 1) / nseg;
    seg = ((seg + batch - 1) / batch) * batch;
    nseg = (rows + seg - 1) / seg;
    const int busiest = (int)(((long long)nseg * strips + 255) / 256);
    const double cost = busiest * (double)(seg + warm + 20) / smrf_cu_rate(busiest);
    if (best_cost == 0.0 || cost < best_cost) {
      best_cost = cost;
      best = nseg;
    }
  }
  if (rounds > 1 && best_cost > 0.0) {
    // A launch that asks for several rounds (the chained kernels: three) against the best single round.  Several rounds keep
    // every CU at `resident` workgroups until the end - a single round ends with the 3 % per residency class its youngest
    // workgroups are behind (profiles/r05_segment_balance.md section 1) - but march more warm-up rows.  Measured, windows
    // 1..10: one round wins below ~10^8 cells (8193^2 -6 %, 4096^2 -6.5 %, 1024^2 -17 %), three win at 16384^2 (+1.9 %).
    const int seg = ((full_seg + batch - 1) / batch) * batch;
    const int nseg = (rows + seg - 1) / seg;
    const double per_cu = (double)nseg * strips / 256.0;               // workgroups a CU runs one after the other, resident at a time
    const double many = (per_cu > resident ? per_cu : resident) * (double)(seg + warm + 20) / smrf_cu_rate(resident);
    const int k1 = (int)(((long long)best * strips + 255) / 256);
    if (many < best_cost * (1.0 + 0.03 * (k1 - 1))) return nseg;
  }
  return best;
}
