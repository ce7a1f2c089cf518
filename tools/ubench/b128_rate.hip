// Micro-benchmark (developer tool, round 2): ds_read_b128 at addresses that are only 8-byte aligned - does gfx950 return the
// right data, and at what rate next to ds_read_b64?  Lane l reads 16 bytes at byte address 8 * l + 8 * k (odd lanes are
// misaligned for 16 bytes).  Prints whether the loaded values are right and core cycles per read per SIMD at 2/4/8 waves.
//   hipcc --offload-arch=gfx950 -O3 -std=c++20 tools/ubench/b128_rate.hip -o tools/ubench/b128_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long cyc, real; };
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int OFF> __device__ __forceinline__ f4 rd128(unsigned a) { f4 v; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }
template <int OFF> __device__ __forceinline__ f2 rd64(unsigned a) { f2 v; asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "n"(OFF)); return v; }
__device__ __forceinline__ void wait0() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

__global__ void check(int* bad) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds) + 8u * threadIdx.x;   // 8-byte aligned only
  f4 v = rd128<8>(a);
  wait0();
  const float e = 2.0f * threadIdx.x + 2.0f;
  if (v.x != e || v.y != e + 1 || v.z != e + 2 || v.w != e + 3) atomicAdd(bad, 1);
}

template <int W>   // W = 16: b128 reads, 8: b64 reads; 12 reads per body, stride 8 bytes per lane
__global__ __launch_bounds__(256) void rate(float* out, int iters, Stamp* st) {
  extern __shared__ float lds[];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = (float)i;
  __syncthreads();
  const unsigned a = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds) + 8u * threadIdx.x;
  float s = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (W == 16) {
      f4 r0 = rd128<8>(a), r1 = rd128<520>(a), r2 = rd128<1032>(a), r3 = rd128<1544>(a), r4 = rd128<2056>(a), r5 = rd128<2568>(a),
         r6 = rd128<3080>(a), r7 = rd128<3592>(a), r8 = rd128<4104>(a), r9 = rd128<4616>(a), ra = rd128<5128>(a), rb = rd128<5640>(a);
      wait0();
      s += r0.x + r1.y + r2.z + r3.w + r4.x + r5.y + r6.z + r7.w + r8.x + r9.y + ra.z + rb.w;
    } else {
      f2 r0 = rd64<8>(a), r1 = rd64<520>(a), r2 = rd64<1032>(a), r3 = rd64<1544>(a), r4 = rd64<2056>(a), r5 = rd64<2568>(a),
         r6 = rd64<3080>(a), r7 = rd64<3592>(a), r8 = rd64<4104>(a), r9 = rd64<4616>(a), ra = rd64<5128>(a), rb = rd64<5640>(a);
      wait0();
      s += r0.x + r1.y + r2.x + r3.y + r4.x + r5.y + r6.x + r7.y + r8.x + r9.y + ra.x + rb.y;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, q1 - q0};
}

int main() {
  int* bad;
  (void)hipMalloc(&bad, 4);
  (void)hipMemset(bad, 0, 4);
  hipLaunchKernelGGL(check, dim3(1), dim3(256), 16384, 0, bad);
  int h = -1;
  (void)hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("ds_read_b128 at 8-byte aligned addresses: %s (%d lanes wrong), %s\n", h == 0 ? "correct data" : "WRONG DATA", h,
         hipGetErrorString(hipGetLastError()));
  float* out;
  Stamp* st;
  (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  (void)hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  const int iters = 20000;
  for (int w : {16, 8}) {
    for (int wps : {1, 2, 4, 8}) {
      const int blocks = 256 * wps;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      auto k = w == 16 ? rate<16> : rate<8>;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, out, 100, st);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 16384, 0, out, iters, st);
      (void)hipEventRecord(e1);
      if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> hs(blocks * 4);
      (void)hipMemcpy(hs.data(), st, hs.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
      double cyc = 0, real = 0;
      for (auto& s : hs) { cyc += (double)s.cyc; real += (double)s.real; }
      const double ghz = cyc / (real * 10.0);
      const double per_read = (ms * 1e-3) * (ghz * 1e9) / ((double)iters * 12 * wps);
      printf("%s  %d waves/SIMD: %.2f cycles per read per SIMD = %.0f B/clk/CU  (12 reads in flight per wave: %.0f cycles per body per wave)\n",
             w == 16 ? "ds_read_b128 (8-byte aligned)" : "ds_read_b64                  ", wps, per_read, 64.0 * w * 4 / per_read,
             per_read * 12 * wps);
      fflush(stdout);
    }
  }
  return 0;
}
