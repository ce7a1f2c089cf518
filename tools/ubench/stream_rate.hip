// Micro-benchmark (developer tool, round 4): what an HBM-streaming kernel of the LSQR solvers' shape can reach on this
// box - the ceiling `smrf_springs_lsqr_f64`'s vector kernels are priced against.  Plain copies at 8 and 16
// bytes per lane, with ordinary and non-temporal accesses, and a "five planes in, four planes out" pass (the traffic of
// xwav_kernel: reads v, w, x, uh, uv, writes x, w, uh, uv), each with a grid that covers the plane once and with the
// solvers' grid (2048 blocks, rows strided over gridDim.y).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_rate.hip -o tools/ubench/stream_rate && tools/ubench/stream_rate [out.md]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

typedef double d2 __attribute__((ext_vector_type(2)));

template <typename V, bool NT>
__global__ __launch_bounds__(256) void copy_flat(const V* __restrict__ in, V* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
    V v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

// the solvers' walk: blockIdx.x picks 256 lanes' worth of columns, blockIdx.y strides over the rows
template <typename V, bool NT>
__global__ __launch_bounds__(256) void copy_rows(const V* __restrict__ in, V* __restrict__ out, int rows, int colsv) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= colsv) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const long long i = (long long)r * colsv + c;
    V v = NT ? __builtin_nontemporal_load(in + i) : in[i];
    if (NT) __builtin_nontemporal_store(v, out + i);
    else out[i] = v;
  }
}

template <typename V, bool NT>
__global__ __launch_bounds__(256) void pass54_rows(V* __restrict__ x, V* __restrict__ w, const V* __restrict__ v, V* __restrict__ uh,
                                                   V* __restrict__ uv, int rows, int colsv, double t) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= colsv) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const long long i = (long long)r * colsv + c;
    V a, b, cc, d, e;
    if (NT) { a = __builtin_nontemporal_load(x + i); b = __builtin_nontemporal_load(w + i); cc = __builtin_nontemporal_load(v + i);
              d = __builtin_nontemporal_load(uh + i); e = __builtin_nontemporal_load(uv + i); }
    else { a = x[i]; b = w[i]; cc = v[i]; d = uh[i]; e = uv[i]; }
    a = a + t * b; b = cc + t * b; d = cc - t * d; e = cc - t * e;
    if (NT) { __builtin_nontemporal_store(a, x + i); __builtin_nontemporal_store(b, w + i); __builtin_nontemporal_store(d, uh + i);
              __builtin_nontemporal_store(e, uv + i); }
    else { x[i] = a; w[i] = b; uh[i] = d; uv[i] = e; }
  }
}

// atu's traffic: three planes in (v, uh, uv), one out (v, in place)
template <typename V, bool NT>
__global__ __launch_bounds__(256) void pass31_rows(V* __restrict__ v, const V* __restrict__ uh, const V* __restrict__ uv, int rows,
                                                   int colsv, double t) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= colsv) return;
  for (int r = blockIdx.y; r < rows; r += gridDim.y) {
    const long long i = (long long)r * colsv + c;
    V a, b, cc;
    if (NT) { a = __builtin_nontemporal_load(v + i); b = __builtin_nontemporal_load(uh + i); cc = __builtin_nontemporal_load(uv + i); }
    else { a = v[i]; b = uh[i]; cc = uv[i]; }
    a = (b + cc) - t * a;
    if (NT) __builtin_nontemporal_store(a, v + i);
    else v[i] = a;
  }
}

template <typename F>
double time_ms(F&& launch, int reps = 7) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  launch();
  (void)hipDeviceSynchronize();
  std::vector<float> ts;
  for (int i = 0; i < reps; ++i) {
    (void)hipEventRecord(e0);
    launch();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ts.push_back(ms);
  }
  std::sort(ts.begin(), ts.end());
  return ts[ts.size() / 2];
}

int main(int argc, char** argv) {
  const int rows = 8192, cols = 8192;
  const long long n = (long long)rows * cols;
  double* p[5];
  for (auto& q : p) { (void)hipMalloc(&q, n * sizeof(double)); (void)hipMemset(q, 0, n * sizeof(double)); }
  FILE* md = argc > 1 ? fopen(argv[1], "w") : nullptr;
  auto report = [&](const char* name, double ms, double bytes) {
    printf("%-58s %7.3f ms  %6.0f GB/s\n", name, ms, bytes / ms / 1e6);
    if (md) fprintf(md, "| %s | %.3f | %.0f |\n", name, ms, bytes / ms / 1e6);
  };
  if (md) fprintf(md, "| kernel (8192^2 float64 planes) | ms | GB/s |\n|---|---|---|\n");
  const double cb = 2.0 * n * 8;
  report("hipMemcpyAsync device to device", time_ms([&] { (void)hipMemcpyAsync(p[1], p[0], n * 8, hipMemcpyDeviceToDevice, 0); }), cb);
  for (int blocks : {2048, 8192, 65536}) {
    char nm[128];
    snprintf(nm, sizeof nm, "copy, 8 B / lane, grid-stride, %d blocks", blocks);
    report(nm, time_ms([&] { hipLaunchKernelGGL((copy_flat<double, false>), dim3(blocks), dim3(256), 0, 0, p[0], p[1], n); }), cb);
    snprintf(nm, sizeof nm, "copy, 16 B / lane, grid-stride, %d blocks", blocks);
    report(nm, time_ms([&] { hipLaunchKernelGGL((copy_flat<d2, false>), dim3(blocks), dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], n / 2); }), cb);
    snprintf(nm, sizeof nm, "copy, 16 B / lane, non-temporal, grid-stride, %d blocks", blocks);
    report(nm, time_ms([&] { hipLaunchKernelGGL((copy_flat<d2, true>), dim3(blocks), dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], n / 2); }), cb);
  }
  {
    const dim3 g8(cols / 256, 2048 / (cols / 256)), g16(cols / 512, 2048 / (cols / 512));
    report("copy, 8 B / lane, solver walk (32 x 64 blocks)", time_ms([&] { hipLaunchKernelGGL((copy_rows<double, false>), g8, dim3(256), 0, 0, p[0], p[1], rows, cols); }), cb);
    report("copy, 8 B / lane, non-temporal, solver walk", time_ms([&] { hipLaunchKernelGGL((copy_rows<double, true>), g8, dim3(256), 0, 0, p[0], p[1], rows, cols); }), cb);
    report("copy, 16 B / lane, solver walk (16 x 128 blocks)", time_ms([&] { hipLaunchKernelGGL((copy_rows<d2, false>), g16, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], rows, cols / 2); }), cb);
    report("copy, 16 B / lane, non-temporal, solver walk", time_ms([&] { hipLaunchKernelGGL((copy_rows<d2, true>), g16, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], rows, cols / 2); }), cb);
    const double pb = 9.0 * n * 8;
    report("5 planes in, 4 out, 8 B / lane, solver walk", time_ms([&] { hipLaunchKernelGGL((pass54_rows<double, false>), g8, dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], rows, cols, 0.5); }), pb);
    report("5 planes in, 4 out, 8 B / lane, non-temporal", time_ms([&] { hipLaunchKernelGGL((pass54_rows<double, true>), g8, dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], rows, cols, 0.5); }), pb);
    report("5 planes in, 4 out, 16 B / lane, solver walk", time_ms([&] { hipLaunchKernelGGL((pass54_rows<d2, false>), g16, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], (d2*)p[2], (d2*)p[3], (d2*)p[4], rows, cols / 2, 0.5); }), pb);
    report("5 planes in, 4 out, 16 B / lane, non-temporal", time_ms([&] { hipLaunchKernelGGL((pass54_rows<d2, true>), g16, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], (d2*)p[2], (d2*)p[3], (d2*)p[4], rows, cols / 2, 0.5); }), pb);
    const double qb = 4.0 * n * 8;
    report("3 planes in, 1 out (in place), 8 B / lane, solver walk", time_ms([&] { hipLaunchKernelGGL((pass31_rows<double, false>), g8, dim3(256), 0, 0, p[0], p[1], p[2], rows, cols, 0.5); }), qb);
    report("3 planes in, 1 out (in place), 8 B / lane, non-temporal", time_ms([&] { hipLaunchKernelGGL((pass31_rows<double, true>), g8, dim3(256), 0, 0, p[0], p[1], p[2], rows, cols, 0.5); }), qb);
    report("3 planes in, 1 out (in place), 16 B / lane, solver walk", time_ms([&] { hipLaunchKernelGGL((pass31_rows<d2, false>), g16, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], (d2*)p[2], rows, cols / 2, 0.5); }), qb);
    for (int gy : {4096 / 16, 8192 / 16}) {
      const dim3 g(cols / 512, gy);
      char nm[128];
      snprintf(nm, sizeof nm, "5 planes in, 4 out, 16 B / lane, %d blocks", 16 * gy);
      report(nm, time_ms([&] { hipLaunchKernelGGL((pass54_rows<d2, false>), g, dim3(256), 0, 0, (d2*)p[0], (d2*)p[1], (d2*)p[2], (d2*)p[3], (d2*)p[4], rows, cols / 2, 0.5); }), pb);
    }
  }
  if (md) fclose(md);
  return 0;
}
