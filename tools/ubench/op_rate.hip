// Micro-benchmark (developer tool, round 2): wall-clock issue rate per SIMD of single VALU opcodes on
// gfx950, 8 independent accumulators, 128 instructions per loop body (.rept), 256-thread blocks,
// 256 * w blocks (w = 2, 4, 8 waves per SIMD when all are resident).  Prints cycles per
// wave-instruction per SIMD at the clock measured in-kernel (s_memtime / s_memrealtime).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/op_rate.hip -o tools/ubench/op_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

struct Stamp { unsigned long long cyc, real; };

#define OP3(NAME, MNEMONIC)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, Stamp* st) {                            \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
             a7 = a0 + 7, b = out[threadIdx.x & 63], c = out[(threadIdx.x + 7) & 63];                          \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();          \
    for (int i = 0; i < iters; ++i)                                                                              \
      asm volatile(".rept 16\n" MNEMONIC " %0, %0, %8, %9\n" MNEMONIC " %1, %1, %8, %9\n" MNEMONIC              \
                   " %2, %2, %8, %9\n" MNEMONIC " %3, %3, %8, %9\n" MNEMONIC " %4, %4, %8, %9\n" MNEMONIC       \
                   " %5, %5, %8, %9\n" MNEMONIC " %6, %6, %8, %9\n" MNEMONIC " %7, %7, %8, %9\n.endr"           \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                   : "v"(b), "v"(c));                                                                            \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();          \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, q1 - q0};            \
  }
#define OP2(NAME, MNEMONIC)                                                                                     \
  __global__ __launch_bounds__(256) void NAME(unsigned* out, int iters, Stamp* st) {                            \
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6,    \
             a7 = a0 + 7, b = out[threadIdx.x & 63];                                                            \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();          \
    for (int i = 0; i < iters; ++i)                                                                              \
      asm volatile(".rept 16\n" MNEMONIC " %0, %0, %8\n" MNEMONIC " %1, %1, %8\n" MNEMONIC " %2, %2, %8\n"      \
                   MNEMONIC " %3, %3, %8\n" MNEMONIC " %4, %4, %8\n" MNEMONIC " %5, %5, %8\n" MNEMONIC          \
                   " %6, %6, %8\n" MNEMONIC " %7, %7, %8\n.endr"                                                 \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
                   : "v"(b));                                                                                    \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();          \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                \
    if ((threadIdx.x & 63) == 0) st[blockIdx.x * 4 + (threadIdx.x >> 6)] = Stamp{t1 - t0, q1 - q0};            \
  }

OP3(k_fma_f32, "v_fma_f32")
OP3(k_min3_f32, "v_min3_f32")
OP3(k_max3_f32, "v_max3_f32")
OP3(k_med3_f32, "v_med3_f32")
OP3(k_min3_i32, "v_min3_i32")
OP3(k_min3_u32, "v_min3_u32")
OP3(k_max3_i32, "v_max3_i32")
OP3(k_minimum3_f32, "v_minimum3_f32")
OP3(k_maximum3_f32, "v_maximum3_f32")
OP3(k_pk_minimum3_f16, "v_pk_minimum3_f16")
OP3(k_min3_f16, "v_min3_f16")
OP3(k_min3_i16, "v_min3_i16")
OP3(k_bfi, "v_bfi_b32")
OP3(k_and_or, "v_and_or_b32")
OP3(k_add3, "v_add3_u32")
OP3(k_mad_u32_u24, "v_mad_u32_u24")
OP3(k_perm, "v_perm_b32")
OP2(k_add_f32, "v_add_f32")
OP2(k_mul_f32, "v_mul_f32")
OP2(k_min_f32, "v_min_f32")
OP2(k_max_f32, "v_max_f32")
OP2(k_min_i32, "v_min_i32")
OP2(k_min_u32, "v_min_u32")
OP2(k_max_u32, "v_max_u32")
OP2(k_add_u32, "v_add_u32")
OP2(k_and_b32, "v_and_b32")
OP2(k_xor_b32, "v_xor_b32")
OP2(k_pk_min_f16, "v_pk_min_f16")
OP2(k_pk_min_i16, "v_pk_min_i16")
OP2(k_lshlrev, "v_lshlrev_b32")
OP2(k_min_f16, "v_min_f16")

typedef void (*kern_t)(unsigned*, int, Stamp*);
struct Entry { const char* name; kern_t k; };

int main(int argc, char** argv) {
  const Entry entries[] = {
      {"v_fma_f32", k_fma_f32}, {"v_add_f32", k_add_f32}, {"v_mul_f32", k_mul_f32},
      {"v_min_f32", k_min_f32}, {"v_max_f32", k_max_f32}, {"v_min3_f32", k_min3_f32}, {"v_max3_f32", k_max3_f32},
      {"v_med3_f32", k_med3_f32}, {"v_minimum3_f32", k_minimum3_f32}, {"v_maximum3_f32", k_maximum3_f32},
      {"v_min_i32", k_min_i32}, {"v_min_u32", k_min_u32}, {"v_max_u32", k_max_u32},
      {"v_min3_i32", k_min3_i32}, {"v_min3_u32", k_min3_u32}, {"v_max3_i32", k_max3_i32},
      {"v_min_f16", k_min_f16}, {"v_min3_f16", k_min3_f16}, {"v_min3_i16", k_min3_i16},
      {"v_pk_min_f16", k_pk_min_f16}, {"v_pk_min_i16", k_pk_min_i16}, {"v_pk_minimum3_f16", k_pk_minimum3_f16},
      {"v_add_u32", k_add_u32}, {"v_and_b32", k_and_b32}, {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshlrev},
      {"v_bfi_b32", k_bfi}, {"v_and_or_b32", k_and_or}, {"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad_u32_u24},
      {"v_perm_b32", k_perm},
  };
  unsigned* out;
  Stamp* st;
  const int wmax = 8, iters = 4000;
  if (hipMalloc(&out, 256 * wmax * 256 * 4) != hipSuccess || hipMalloc(&st, 256 * wmax * 4 * sizeof(Stamp)) != hipSuccess) return 1;
  (void)hipMemset(out, 0, 256 * wmax * 256 * 4);
  FILE* md = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (md) fprintf(md, "| opcode | cyc/inst/SIMD, 2 waves/SIMD | 4 waves | 8 waves | clock GHz |\n|---|---|---|---|---|\n");
  for (const Entry& e : entries) {
    double res[3], ghz = 0;
    int j = 0;
    for (int wps : {2, 4, 8}) {
      const int blocks = 256 * wps;
      hipEvent_t e0, e1;
      (void)hipEventCreate(&e0);
      (void)hipEventCreate(&e1);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, 100, st);
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, out, iters, st);
      (void)hipEventRecord(e1);
      if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", e.name); return 1; }
      float ms = 0;
      (void)hipEventElapsedTime(&ms, e0, e1);
      std::vector<Stamp> h(blocks * 4);
      (void)hipMemcpy(h.data(), st, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
      double cyc = 0, real = 0;
      for (auto& s : h) { cyc += (double)s.cyc; real += (double)s.real; }
      ghz = cyc / (real * 10.0);
      const double inst_per_simd = (double)iters * 128 * wps;              // every SIMD runs wps waves of the grid
      res[j++] = (ms * 1e-3) * (ghz * 1e9) / inst_per_simd;                // wall time in core cycles per instruction
    }
    printf("%-20s  %5.2f  %5.2f  %5.2f  cyc/inst/SIMD at 2/4/8 waves per SIMD (clock %.2f GHz)\n", e.name, res[0], res[1], res[2], ghz);
    fflush(stdout);
    if (md) fprintf(md, "| `%s` | %.2f | %.2f | %.2f | %.2f |\n", e.name, res[0], res[1], res[2], ghz);
  }
  if (md) fclose(md);
  return 0;
}
