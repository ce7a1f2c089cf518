// Micro-benchmark (developer tool): sustained issue rate of the ring kernels' instruction kinds
// on gfx950 as a function of waves per SIMD.  hipcc --offload-arch=gfx950 -O3 issue_rate.hip -o issue_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ void k(float* out, int iters, unsigned long long* cyc) {
  __shared__ float2 lds[1024];
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = out[threadIdx.x & 63], c = out[(threadIdx.x + 7) & 63];
  lds[threadIdx.x & 1023] = make_float2(b, c);
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + (threadIdx.x & 255));
  float2 r0, r1, r2, r3;
  r0 = r1 = r2 = r3 = make_float2(0, 0);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == 0) {   // 8 independent v_min3_f32
      asm volatile("v_min3_f32 %0, %0, %8, %9\nv_min3_f32 %1, %1, %8, %9\nv_min3_f32 %2, %2, %8, %9\nv_min3_f32 %3, %3, %8, %9\n"
                   "v_min3_f32 %4, %4, %8, %9\nv_min3_f32 %5, %5, %8, %9\nv_min3_f32 %6, %6, %8, %9\nv_min3_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (KIND == 1) {   // 8 independent v_min_f32
      asm volatile("v_min_f32 %0, %0, %8\nv_min_f32 %1, %1, %8\nv_min_f32 %2, %2, %8\nv_min_f32 %3, %3, %8\n"
                   "v_min_f32 %4, %4, %8\nv_min_f32 %5, %5, %8\nv_min_f32 %6, %6, %8\nv_min_f32 %7, %7, %8"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (KIND == 2) {   // 4 ds_read_b64 + wait
      asm volatile("ds_read_b64 %0, %4 offset:0\nds_read_b64 %1, %4 offset:2048\nds_read_b64 %2, %4 offset:4096\nds_read_b64 %3, %4 offset:6144\n"
                   "s_waitcnt lgkmcnt(0)" : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
      a0 += r0.x + r1.x + r2.x + r3.x;
    } else if constexpr (KIND == 3) {   // mix: 4 ds_read_b64 then 8 v_min3 then wait (like one lookup group + ring slots)
      asm volatile("ds_read_b64 %0, %4 offset:0\nds_read_b64 %1, %4 offset:2048\nds_read_b64 %2, %4 offset:4096\nds_read_b64 %3, %4 offset:6144"
                   : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3) : "v"(addr));
      asm volatile("v_min3_f32 %0, %0, %8, %9\nv_min3_f32 %1, %1, %8, %9\nv_min3_f32 %2, %2, %8, %9\nv_min3_f32 %3, %3, %8, %9\n"
                   "v_min3_f32 %4, %4, %8, %9\nv_min3_f32 %5, %5, %8, %9\nv_min3_f32 %6, %6, %8, %9\nv_min3_f32 %7, %7, %8, %9"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      a0 += r0.x + r1.x + r2.x + r3.x;
    } else if constexpr (KIND == 4) {   // 8 s_add (SALU)
      int s = i;
      asm volatile("s_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1\n"
                   "s_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1\ns_add_i32 %0, %0, 1" : "+s"(s));
      a0 += s;
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
int run(const char* name, int per_iter) {
  float* out; unsigned long long* cyc;
  CHECK(hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float)));
  CHECK(hipMalloc(&cyc, 8));
  for (int wps : {1, 2, 3, 4, 8}) {                 // waves per SIMD: blocks of 256 threads = 1 wave per SIMD
    const int iters = 200000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, 100, cyc);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h; CHECK(hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost));
    printf("%-26s waves/SIMD %d: %.2f ticks/inst/wave  tick = %.2f ns  -> %.2f ns per inst per wave, %.3f G inst/s/SIMD\n",
           name, wps, (double)h / iters / per_iter, ms * 1e6 / (double)h, ms * 1e6 / iters / per_iter,
           (double)iters * per_iter * wps / (ms * 1e6));
  }
  return 0;
}

int main() {
  run<0>("v_min3_f32 x8", 8);
  run<1>("v_min_f32 x8", 8);
  run<2>("ds_read_b64 x4 + wait", 4);
  run<3>("4 ds_read_b64 + 8 v_min3", 12);
  run<4>("s_add_i32 x8", 8);
  return 0;
}
