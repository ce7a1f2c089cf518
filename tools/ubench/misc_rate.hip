// Micro-benchmark (developer tool, round 3): issue rate per SIMD of the non-min/max VALU instructions found in the ring
// kernels' prefetch / epilogue code: 64-bit address arithmetic (v_lshl_add_u64), v_mov_b32, the flag step's
// v_cvt_f64_f32 + v_cmp_lt_f64, v_cndmask.  Same method as op_rate.hip (8 independent chains, 128 instructions per loop
// body, 256-thread blocks, 256 * w blocks, wall clock x in-kernel clock).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/misc_rate.hip -o tools/ubench/misc_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long cyc, real; };

#define KERNEL(NAME, BODY, DECL, SINK)                                                                           \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, Stamp* st) {                            \
    DECL                                                                                                         \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();          \
    for (int i = 0; i < iters; ++i) { BODY }                                                                     \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();          \
    out[blockIdx.x * 256 + threadIdx.x] = SINK;                                                                  \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, q1 - q0};            \
  }

#define DECL64 unsigned long long a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, \
                                   a6 = a0 + 6, a7 = a0 + 7, b = out[threadIdx.x & 63];
#define DECL32 unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
                        a7 = a0 + 7, b = out[threadIdx.x & 63];
#define REGS8 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b)

KERNEL(k_lshl_add_u64,
       asm volatile(".rept 16\nv_lshl_add_u64 %0, %0, 2, %8\nv_lshl_add_u64 %1, %1, 2, %8\nv_lshl_add_u64 %2, %2, 2, %8\n"
                    "v_lshl_add_u64 %3, %3, 2, %8\nv_lshl_add_u64 %4, %4, 2, %8\nv_lshl_add_u64 %5, %5, 2, %8\n"
                    "v_lshl_add_u64 %6, %6, 2, %8\nv_lshl_add_u64 %7, %7, 2, %8\n.endr" REGS8);,
       DECL64, (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_mov_b32,
       asm volatile(".rept 16\nv_mov_b32 %0, %8\nv_mov_b32 %1, %8\nv_mov_b32 %2, %8\nv_mov_b32 %3, %8\nv_mov_b32 %4, %8\n"
                    "v_mov_b32 %5, %8\nv_mov_b32 %6, %8\nv_mov_b32 %7, %8\n.endr" REGS8);,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_cvt_f64_f32,
       asm volatile(".rept 16\nv_cvt_f64_f32 %0, %8\nv_cvt_f64_f32 %1, %8\nv_cvt_f64_f32 %2, %8\nv_cvt_f64_f32 %3, %8\n"
                    "v_cvt_f64_f32 %4, %8\nv_cvt_f64_f32 %5, %8\nv_cvt_f64_f32 %6, %8\nv_cvt_f64_f32 %7, %8\n.endr"
                    : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"((unsigned)b));,
       DECL64, (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_cmp_lt_f64,
       asm volatile(".rept 16\nv_cmp_lt_f64 vcc, %0, %8\nv_cmp_lt_f64 vcc, %1, %8\nv_cmp_lt_f64 vcc, %2, %8\n"
                    "v_cmp_lt_f64 vcc, %3, %8\nv_cmp_lt_f64 vcc, %4, %8\nv_cmp_lt_f64 vcc, %5, %8\nv_cmp_lt_f64 vcc, %6, %8\n"
                    "v_cmp_lt_f64 vcc, %7, %8\n.endr" REGS8 : "vcc");,
       DECL64, (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_cmp_lt_f32,
       asm volatile(".rept 16\nv_cmp_lt_f32 vcc, %0, %8\nv_cmp_lt_f32 vcc, %1, %8\nv_cmp_lt_f32 vcc, %2, %8\n"
                    "v_cmp_lt_f32 vcc, %3, %8\nv_cmp_lt_f32 vcc, %4, %8\nv_cmp_lt_f32 vcc, %5, %8\nv_cmp_lt_f32 vcc, %6, %8\n"
                    "v_cmp_lt_f32 vcc, %7, %8\n.endr" REGS8 : "vcc");,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_cndmask,
       asm volatile(".rept 16\nv_cndmask_b32 %0, %0, %8, vcc\nv_cndmask_b32 %1, %1, %8, vcc\nv_cndmask_b32 %2, %2, %8, vcc\n"
                    "v_cndmask_b32 %3, %3, %8, vcc\nv_cndmask_b32 %4, %4, %8, vcc\nv_cndmask_b32 %5, %5, %8, vcc\n"
                    "v_cndmask_b32 %6, %6, %8, vcc\nv_cndmask_b32 %7, %7, %8, vcc\n.endr" REGS8 : "vcc");,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_sub_f32,
       asm volatile(".rept 16\nv_sub_f32 %0, %0, %8\nv_sub_f32 %1, %1, %8\nv_sub_f32 %2, %2, %8\nv_sub_f32 %3, %3, %8\n"
                    "v_sub_f32 %4, %4, %8\nv_sub_f32 %5, %5, %8\nv_sub_f32 %6, %6, %8\nv_sub_f32 %7, %7, %8\n.endr" REGS8);,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)
KERNEL(k_min3_f32,
       asm volatile(".rept 16\nv_min3_f32 %0, %0, %8, %8\nv_min3_f32 %1, %1, %8, %8\nv_min3_f32 %2, %2, %8, %8\n"
                    "v_min3_f32 %3, %3, %8, %8\nv_min3_f32 %4, %4, %8, %8\nv_min3_f32 %5, %5, %8, %8\nv_min3_f32 %6, %6, %8, %8\n"
                    "v_min3_f32 %7, %7, %8, %8\n.endr" REGS8);,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

// round 4: the 64-bit and packed moves (does turning the in-place ring back pair by pair save issue time?)
KERNEL(k_mov_b64,
       asm volatile(".rept 16\nv_mov_b64 %0, %8\nv_mov_b64 %1, %8\nv_mov_b64 %2, %8\nv_mov_b64 %3, %8\nv_mov_b64 %4, %8\n"
                    "v_mov_b64 %5, %8\nv_mov_b64 %6, %8\nv_mov_b64 %7, %8\n.endr" REGS8);,
       DECL64, (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_pk_mov_b32,
       asm volatile(".rept 16\nv_pk_mov_b32 %0, %8, %8\nv_pk_mov_b32 %1, %8, %8\nv_pk_mov_b32 %2, %8, %8\nv_pk_mov_b32 %3, %8, %8\n"
                    "v_pk_mov_b32 %4, %8, %8\nv_pk_mov_b32 %5, %8, %8\nv_pk_mov_b32 %6, %8, %8\nv_pk_mov_b32 %7, %8, %8\n.endr" REGS8);,
       DECL64, (unsigned)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7))
KERNEL(k_min_f32_dpp,
       asm volatile(".rept 16\nv_min_f32_dpp %0, %8, %0 row_shr:1 row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %1, %8, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    "v_min_f32_dpp %2, %8, %2 row_shr:1 row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %3, %8, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    "v_min_f32_dpp %4, %8, %4 row_shr:1 row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %5, %8, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                    "v_min_f32_dpp %6, %8, %6 row_shr:1 row_mask:0xf bank_mask:0xf\nv_min_f32_dpp %7, %8, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n.endr" REGS8);,
       DECL32, a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7)

typedef void (*kern_t)(unsigned*, int, Stamp*);
struct Entry { const char* name; kern_t k; };

int main(int argc, char** argv) {
  const Entry entries[] = {{"v_min3_f32", k_min3_f32}, {"v_lshl_add_u64", k_lshl_add_u64}, {"v_mov_b32", k_mov_b32},
                           {"v_sub_f32", k_sub_f32}, {"v_cvt_f64_f32", k_cvt_f64_f32}, {"v_cmp_lt_f64", k_cmp_lt_f64},
                           {"v_cmp_lt_f32", k_cmp_lt_f32}, {"v_cndmask_b32", k_cndmask}, {"v_mov_b64", k_mov_b64},
                           {"v_pk_mov_b32", k_pk_mov_b32}, {"v_min_f32_dpp row_shr:1", k_min_f32_dpp}};
  const int iters = 4000;
  unsigned* out;
  Stamp* st;
  hipMalloc(&out, 256 * 8 * 256 * 4 + 1024);
  hipMalloc(&st, 256 * 8 * 4 * sizeof(Stamp));
  hipMemset(out, 0, 256 * 8 * 256 * 4 + 1024);
  FILE* md = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (md) fprintf(md, "| opcode | cyc/inst/SIMD, 2 waves/SIMD | 4 waves | 8 waves | clock GHz |\n|---|---|---|---|---|\n");
  for (const Entry& e : entries) {
    double cyc[3] = {0, 0, 0}, ghz = 0;
    int wi = 0;
    for (int w : {2, 4, 8}) {
      const int blocks = 256 * w;
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 10, st);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, st);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> h(blocks * 4);
      hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
      double c = 0, r = 0;
      for (auto& s : h) { c += s.cyc; r += s.real; }
      ghz = (c / r) * 0.1;                                 // s_memrealtime ticks at 100 MHz
      const double insts_per_simd = (double)iters * 128.0 * w;
      cyc[wi++] = ms * 1e-3 * ghz * 1e9 / insts_per_simd;
    }
    printf("%-16s %6.2f %6.2f %6.2f cycles/inst/SIMD at 2 / 4 / 8 waves per SIMD (%.2f GHz)\n", e.name, cyc[0], cyc[1], cyc[2], ghz);
    if (md) fprintf(md, "| `%s` | %.2f | %.2f | %.2f | %.2f |\n", e.name, cyc[0], cyc[1], cyc[2], ghz);
  }
  if (md) fclose(md);
  return 0;
}
