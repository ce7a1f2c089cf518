// Experiment library for the two-role ring kernels (morph_ring2.h): NOT part of libsmrf_hip.so.
//   hipcc -std=c++20 -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -Iinclude -Ineilpy_amd/csrc \
//         [-DSMRF_RING2_NP=3] tools/experiments/ring2_exp.hip -o tools/experiments/libring2_exp.so
// One entry: a whole-raster erosion (mask == NULL) or dilation + flag step with the two-role kernel of the radius;
// tools/ring2_probe.py compares it with the product kernels (bit equality and time).
#include <cstdarg>
#include <cstdio>

#include "morph_ring2.h"

// smrf_fail / smrf_set_error live in the product library's core.hip; the experiment reports through stderr
int smrf_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vfprintf(stderr, fmt, ap);
  va_end(ap);
  fputc('\n', stderr);
  return code;
}

namespace {
template <int R>
int run(const DiskArgs<float>& a, bool dilate, hipStream_t s) {
  return dilate ? smrf::ring2_launch<float, R, true>(a, s) : smrf::ring2_launch<float, R, false>(a, s);
}
}  // namespace

extern "C" __attribute__((visibility("default"))) int smrf_exp_ring2_f32(const float* in, const float* last, float* out,
                                                                         uint8_t* mask, double thr, int rows, int cols,
                                                                         int radius, int dilate, void* stream) {
  DiskArgs<float> a{};
  a.in = in; a.out = out; a.last = last; a.mask = mask; a.when = nullptr; a.thr = thr; a.widx = 0;
  a.img_rows = rows; a.cols = cols; a.ld = cols; a.in_row0 = 0; a.in_rows = rows; a.out_row0 = 0; a.out_rows = rows;
  a.radius = radius; a.nan_aware = 0; a.seg = 0;
  hipStream_t s = (hipStream_t)stream;
  switch (radius) {
    case 20: return run<20>(a, dilate, s);
    case 25: return run<25>(a, dilate, s);
    case 32: return run<32>(a, dilate, s);
    case 36: return run<36>(a, dilate, s);
    case 40: return run<40>(a, dilate, s);
    case 44: return run<44>(a, dilate, s);
    case 50: return run<50>(a, dilate, s);
    case 60: return run<60>(a, dilate, s);
    default: fprintf(stderr, "ring2 experiment: radius %d not instantiated\n", radius); return -1;
  }
}
