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
// round as before.  Rules 1, 2: one full round of `rounds x resident x 256 / strips` segments, rounded to nearest (rounds
// 1-4; a second, nearly empty round whenever that rounds up) or down, with segments of at least `min_seg` rows.
inline int smrf_pick_nseg(int rows, int strips, int resident, int rounds, int warm, int batch, int min_seg, int rule) {
  if (rule != 0 || rounds != 1) {
    const int nseg = (rounds * resident * 256 + (rule == 1 ? strips / 2 : 0)) / strips;
    int seg = (rows + (nseg > 1 ? nseg : 1) - 1) / (nseg > 1 ? nseg : 1);
    seg = seg > min_seg ? seg : min_seg;
    seg = seg < rows ? seg : rows;
    return (rows + seg - 1) / seg;
  }
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
    const double x = busiest <= 1 ? 1.0 : busiest == 2 ? 1.57 : busiest == 3 ? 2.1 : 2.37 + 0.1 * (busiest - 4);
    const double cost = busiest * (double)(seg + warm + 20) / x;
    if (best_cost == 0.0 || cost < best_cost) {
      best_cost = cost;
      best = nseg;
    }
  }
  return best;
}
