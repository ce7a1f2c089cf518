// Micro-benchmark (developer tool, round 2): do VALU min/max and LDS reads overlap on a gfx950 SIMD, or do their
// times add?  One loop body = NL ds_read_b64 (conflict-free, consecutive lanes) + one s_waitcnt + NV v_min3_f32 on
// 8 independent accumulators whose other operands are the values just read - the shape of the ring kernel's lookup
// groups (morph_ring.h).  PIPE = 1 issues the next body's reads before this body's VALU (software pipelining inside
// one wave, two register sets).  256-thread blocks, 256 * w blocks = w waves per SIMD, all resident.
// Prints wall-clock core cycles per loop body per SIMD next to the VALU-only and LDS-only bodies.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench/mix_rate.hip -o tools/ubench/mix_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <utility>
#include <vector>

struct Stamp { unsigned long long cyc, real; };

template <int OFF>
__device__ __forceinline__ float2 rd(unsigned addr) {
  float2 v;
  asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
__device__ __forceinline__ void wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ float min3(float a, float b, float c) {
  float r;
  asm volatile("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}

template <int NL, int... I>
__device__ __forceinline__ void issue(float2 (&r)[14], unsigned addr, std::integer_sequence<int, I...>) {
  ((r[I] = rd<I * 1024 + 64>(addr)), ...);
}
template <int NV, int NL, int... J>
__device__ __forceinline__ void valu(float (&acc)[8], const float2 (&r)[14], std::integer_sequence<int, J...>) {
  constexpr int M = NL > 0 ? NL : 14;
  ((acc[J % 8] = min3(acc[J % 8], r[J % M].x, r[(J + 3) % M].y)), ...);
}

template <int NV, int NL, int PIPE>
__global__ __launch_bounds__(256) void mix(float* out, int iters, Stamp* st) {
  extern __shared__ float2 lds[];
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = float2{(float)i, (float)(i ^ 5)};
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + threadIdx.x);
  float acc[8];
  float2 r[2][14];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = 1e30f - threadIdx.x - i;
#pragma unroll
  for (int i = 0; i < 14; ++i) r[0][i] = r[1][i] = float2{(float)i + out[threadIdx.x & 63], 3.f * i};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  if constexpr (PIPE == 0) {
    for (int it = 0; it < iters; ++it) {
      if constexpr (NL > 0) { issue<NL>(r[0], addr, std::make_integer_sequence<int, NL>{}); wait0(); }
      if constexpr (NV > 0) valu<NV, NL>(acc, r[0], std::make_integer_sequence<int, NV>{});
    }
  } else {
    issue<NL>(r[0], addr, std::make_integer_sequence<int, NL>{});
    for (int it = 0; it < iters; it += 2) {
      wait0();
      issue<NL>(r[1], addr, std::make_integer_sequence<int, NL>{});
      valu<NV, NL>(acc, r[0], std::make_integer_sequence<int, NV>{});
      wait0();
      issue<NL>(r[0], addr, std::make_integer_sequence<int, NL>{});
      valu<NV, NL>(acc, r[1], std::make_integer_sequence<int, NV>{});
    }
    wait0();
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i];
#pragma unroll
  for (int i = 0; i < 14; ++i) s += r[0][i].x + r[1][i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, q1 - q0};
}

typedef void (*kern_t)(float*, int, Stamp*);
struct Entry { const char* name; kern_t k; int nv, nl; };

int main(int argc, char** argv) {
  const Entry entries[] = {
      {"VALU only: 32 v_min3", mix<32, 0, 0>, 32, 0},
      {"LDS only: 14 ds_read_b64", mix<0, 14, 0>, 0, 14},
      {"14 reads, wait, 32 v_min3", mix<32, 14, 0>, 32, 14},
      {"same, next reads issued before the VALU", mix<32, 14, 1>, 32, 14},
      {"VALU only: 16 v_min3", mix<16, 0, 0>, 16, 0},
      {"14 reads, wait, 16 v_min3", mix<16, 14, 0>, 16, 14},
      {"same, next reads issued before the VALU", mix<16, 14, 1>, 16, 14},
      {"LDS only: 7 ds_read_b64", mix<0, 7, 0>, 0, 7},
      {"7 reads, wait, 32 v_min3", mix<32, 7, 0>, 32, 7},
      {"same, next reads issued before the VALU", mix<32, 7, 1>, 32, 7},
  };
  float* out;
  Stamp* st;
  const int wmax = 8, iters = 20000;
  if (hipMalloc(&out, 256 * wmax * 256 * 4) != hipSuccess || hipMalloc(&st, 256 * wmax * 4 * sizeof(Stamp)) != hipSuccess) return 1;
  (void)hipMemset(out, 0, 256 * wmax * 256 * 4);
  FILE* md = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (md)
    fprintf(md, "| loop body | cycles/body/SIMD, 1 wave/SIMD | 2 waves | 4 waves | 8 waves | clock GHz |\n|---|---|---|---|---|---|\n");
  for (const Entry& e : entries) {
    double res[4], ghz = 0;
    int j = 0;
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 2048 * 8, 0, out, 100, st);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 2048 * 8, 0, out, iters, st);
      (void)hipEventRecord(e1);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", e.name); return 1; }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> h(blocks * 4);
      (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
      double cyc = 0, real = 0;
      for (auto& s : h) { cyc += (double)s.cyc; real += (double)s.real; }
      ghz = cyc / (real * 10.0);
      res[j++] = (ms * 1e-3) * (ghz * 1e9) / ((double)iters * wps);       // core cycles per body per SIMD (wps waves share it)
    }
    printf("%-44s %7.1f %7.1f %7.1f %7.1f  cycles/body/SIMD at 1/2/4/8 waves per SIMD (clock %.2f GHz)\n", e.name, res[0],
           res[1], res[2], res[3], ghz);
    fflush(stdout);
    if (md) fprintf(md, "| %s | %.1f | %.1f | %.1f | %.1f | %.2f |\n", e.name, res[0], res[1], res[2], res[3], ghz);
  }
  if (md) fclose(md);
  return 0;
}
