// Micro-benchmark (developer tool, round 2): what one gfx950 SIMD sustains of the ring kernels'
// instruction kinds, as a function of resident waves per SIMD, in CYCLES PER WAVE-INSTRUCTION.
//
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/issue_rate.hip -o /tmp/issue_rate && /tmp/issue_rate
//   hipcc --offload-arch=gfx950 -O3 --offload-device-only -S tools/ubench/issue_rate.hip -o issue_rate.s   (the loop's ISA)
//
// Every loop body is >= 128 instructions of the measured kind(s), emitted with .rept from inline asm,
// so that the 3 scalar loop-control instructions are < 2.5 % of the stream.  Each wave stamps
// s_memtime (shader clock) and s_memrealtime (constant 100 MHz) around its loop: cycles come from the
// first, the core clock the loop ran at from the ratio of the two.  Blocks are 256 threads = one wave
// on each of the CU's 4 SIMDs; a grid of 256 * w blocks puts w waves on every SIMD (no LDS/VGPR limit
// in the way), which the per-CU wave census the kernel writes confirms.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <type_traits>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

enum Kind { FMA = 0, MIN2, MIN3, PKFMA, MIN3_DEP, LDS_B64, MIX_LDS_MIN3, MIN3_SALU, MIN3_2CHAIN, NKIND };
static const char* kNames[NKIND] = {
    "v_fma_f32 (8 independent accumulators)",
    "v_min_f32 (8 independent)",
    "v_min3_f32 (8 independent)",
    "v_pk_fma_f32 (8 independent, 2 fp32 lanes each)",
    "v_min3_f32 (ONE dependent chain)",
    "ds_read_b64 (16 in flight, counted waits)",
    "2 v_min3_f32 : 1 ds_read_b64 (the consume mix; counts VALU+LDS)",
    "2 v_min3_f32 : 1 s_add_u32 (counts VALU only)",
    "v_min3_f32 (two dependent chains)",
};
// instructions counted per loop iteration (the .rept factor times the body)
static const int kPerIter[NKIND] = {128, 128, 128, 128, 128, 128, 192, 128, 128};

struct Stamp { unsigned long long cyc, real; unsigned hwid; };

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, Stamp* stamps) {
  __shared__ float2 lds[2048];
  float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b = out[threadIdx.x & 63], c = out[(threadIdx.x + 7) & 63];
  float2 pa = make_float2(a0, a1), pb = make_float2(a2, a3), pc = make_float2(a4, a5), pd = make_float2(a6, a7);
  float2 pe = pa, pf = pb, pg = pc, ph = pd, pm = make_float2(b, c);
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = make_float2(b + i, c);
  __syncthreads();
  unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) void*)(lds + threadIdx.x);
  float2 r0 = {0, 0}, r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, r6 = r0, r7 = r0;
  int s0 = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long q0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if constexpr (KIND == FMA) {
      asm volatile(".rept 16\n"
                   "v_fma_f32 %0, %0, %8, %9\nv_fma_f32 %1, %1, %8, %9\nv_fma_f32 %2, %2, %8, %9\nv_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\nv_fma_f32 %5, %5, %8, %9\nv_fma_f32 %6, %6, %8, %9\nv_fma_f32 %7, %7, %8, %9\n"
                   ".endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (KIND == MIN2) {
      asm volatile(".rept 16\n"
                   "v_min_f32 %0, %0, %8\nv_min_f32 %1, %1, %8\nv_min_f32 %2, %2, %8\nv_min_f32 %3, %3, %8\n"
                   "v_min_f32 %4, %4, %8\nv_min_f32 %5, %5, %8\nv_min_f32 %6, %6, %8\nv_min_f32 %7, %7, %8\n"
                   ".endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
    } else if constexpr (KIND == MIN3) {
      asm volatile(".rept 16\n"
                   "v_min3_f32 %0, %0, %8, %9\nv_min3_f32 %1, %1, %8, %9\nv_min3_f32 %2, %2, %8, %9\nv_min3_f32 %3, %3, %8, %9\n"
                   "v_min3_f32 %4, %4, %8, %9\nv_min3_f32 %5, %5, %8, %9\nv_min3_f32 %6, %6, %8, %9\nv_min3_f32 %7, %7, %8, %9\n"
                   ".endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
    } else if constexpr (KIND == PKFMA) {
      asm volatile(".rept 16\n"
                   "v_pk_fma_f32 %0, %0, %8, %8\nv_pk_fma_f32 %1, %1, %8, %8\nv_pk_fma_f32 %2, %2, %8, %8\nv_pk_fma_f32 %3, %3, %8, %8\n"
                   "v_pk_fma_f32 %4, %4, %8, %8\nv_pk_fma_f32 %5, %5, %8, %8\nv_pk_fma_f32 %6, %6, %8, %8\nv_pk_fma_f32 %7, %7, %8, %8\n"
                   ".endr"
                   : "+v"(pa), "+v"(pb), "+v"(pc), "+v"(pd), "+v"(pe), "+v"(pf), "+v"(pg), "+v"(ph) : "v"(pm));
    } else if constexpr (KIND == MIN3_DEP) {
      asm volatile(".rept 128\nv_min3_f32 %0, %0, %1, %2\n.endr" : "+v"(a0) : "v"(b), "v"(c));
    } else if constexpr (KIND == MIN3_2CHAIN) {
      asm volatile(".rept 64\nv_min3_f32 %0, %0, %2, %3\nv_min3_f32 %1, %1, %2, %3\n.endr" : "+v"(a0), "+v"(a1) : "v"(b), "v"(c));
    } else if constexpr (KIND == LDS_B64) {
      // 8 reads, then 16 x { wait for the older 8 of 16 outstanding... } kept simple: issue 8, wait until 8 left, issue 8 more
      asm volatile("ds_read_b64 %0, %8 offset:0\nds_read_b64 %1, %8 offset:2048\nds_read_b64 %2, %8 offset:4096\nds_read_b64 %3, %8 offset:6144\n"
                   "ds_read_b64 %4, %8 offset:8192\nds_read_b64 %5, %8 offset:10240\nds_read_b64 %6, %8 offset:12288\nds_read_b64 %7, %8 offset:14336\n"
                   ".rept 15\n"
                   "s_waitcnt lgkmcnt(4)\n"
                   "ds_read_b64 %0, %8 offset:0\nds_read_b64 %1, %8 offset:2048\nds_read_b64 %2, %8 offset:4096\nds_read_b64 %3, %8 offset:6144\n"
                   "s_waitcnt lgkmcnt(4)\n"
                   "ds_read_b64 %4, %8 offset:8192\nds_read_b64 %5, %8 offset:10240\nds_read_b64 %6, %8 offset:12288\nds_read_b64 %7, %8 offset:14336\n"
                   ".endr\n"
                   "s_waitcnt lgkmcnt(0)"
                   : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(addr));
    } else if constexpr (KIND == MIX_LDS_MIN3) {
      // per step: 4 reads (next group) | wait for the previous 4 | 8 min3 on the previous group's values.
      // Separate asm statements (volatile: kept in order) so that the compiler names the halves of the 64-bit reads.
      auto rd4 = [](unsigned ad, float2& x0, float2& x1, float2& x2, float2& x3, auto off) {
        constexpr int O = decltype(off)::value;
        asm volatile("ds_read_b64 %0, %4 offset:%5\nds_read_b64 %1, %4 offset:%6\nds_read_b64 %2, %4 offset:%7\nds_read_b64 %3, %4 offset:%8"
                     : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3) : "v"(ad), "n"(O), "n"(O + 2048), "n"(O + 4096), "n"(O + 6144));
      };
      auto mn8 = [](float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float& a6, float& a7,
                    const float2& x0, const float2& x1, const float2& x2, const float2& x3) {
        asm volatile("s_waitcnt lgkmcnt(4)\n"
                     "v_min3_f32 %0, %0, %8, %9\nv_min3_f32 %1, %1, %10, %11\nv_min3_f32 %2, %2, %12, %13\nv_min3_f32 %3, %3, %14, %15\n"
                     "v_min3_f32 %4, %4, %8, %10\nv_min3_f32 %5, %5, %9, %11\nv_min3_f32 %6, %6, %12, %14\nv_min3_f32 %7, %7, %13, %15"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(x0.x), "v"(x0.y), "v"(x1.x), "v"(x1.y), "v"(x2.x), "v"(x2.y), "v"(x3.x), "v"(x3.y));
      };
      rd4(addr, r0, r1, r2, r3, std::integral_constant<int, 0>{});
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        rd4(addr, r4, r5, r6, r7, std::integral_constant<int, 8192>{});
        mn8(a0, a1, a2, a3, a4, a5, a6, a7, r0, r1, r2, r3);
        rd4(addr, r0, r1, r2, r3, std::integral_constant<int, 0>{});
        mn8(a0, a1, a2, a3, a4, a5, a6, a7, r4, r5, r6, r7);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else if constexpr (KIND == MIN3_SALU) {
      asm volatile(".rept 16\n"
                   "v_min3_f32 %0, %0, %9, %10\nv_min3_f32 %1, %1, %9, %10\ns_add_u32 %8, %8, 1\nv_min3_f32 %2, %2, %9, %10\nv_min3_f32 %3, %3, %9, %10\ns_add_u32 %8, %8, 1\n"
                   "v_min3_f32 %4, %4, %9, %10\nv_min3_f32 %5, %5, %9, %10\ns_add_u32 %8, %8, 1\nv_min3_f32 %6, %6, %9, %10\nv_min3_f32 %7, %7, %9, %10\ns_add_u32 %8, %8, 1\n"
                   ".endr"
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0) : "v"(b), "v"(c) : "scc");
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long q1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + pa.x + pb.y + pc.x + pd.y + pe.x + pf.y + pg.x + ph.y +
                                        r0.x + r1.x + r2.x + r3.x + r4.x + r5.x + r6.x + r7.x + r0.y + (float)s0;
  if ((threadIdx.x & 63) == 0) {
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    Stamp st{t1 - t0, q1 - q0, hwid};
    stamps[blockIdx.x * 4 + (threadIdx.x >> 6)] = st;
  }
}

template <int KIND>
int run(FILE* md) {
  float* out;
  Stamp* stamps;
  const int wmax = 8;
  CHECK(hipMalloc(&out, 256 * wmax * 256 * sizeof(float)));
  CHECK(hipMemset(out, 0, 256 * wmax * 256 * sizeof(float)));
  CHECK(hipMalloc(&stamps, 256 * wmax * 4 * sizeof(Stamp)));
  const int iters = 20000;
  printf("%s\n", kNames[KIND]);
  if (md) fprintf(md, "| %s |", kNames[KIND]);
  for (int wps : {1, 2, 3, 4, 8}) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 200, stamps);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, stamps);
    hipEventRecord(e1);
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<Stamp> h(blocks * 4);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    double cyc = 0, real = 0;
    for (auto& s : h) { cyc += (double)s.cyc; real += (double)s.real; }
    cyc /= h.size();
    real /= h.size();
    // census: waves per (XCC, SE, CU, SIMD) from HW_ID (gfx9 layout: simd [5:4], cu [11:8], sh [12], se [15:13]; XCC id not in HW_ID)
    const double n_inst = (double)iters * kPerIter[KIND];
    const double cyc_per_inst_wave = cyc / n_inst;                // one wave's view
    const double cyc_per_inst_simd = cyc / (n_inst * wps);        // the SIMD's view (w waves share it)
    const double ghz = cyc / (real * 10.0);                       // 100 MHz realtime -> 10 ns per tick
    const double rate = n_inst * wps / (ms * 1e6);                // G inst/s/SIMD from the wall clock
    fflush(stdout);
    printf("  waves/SIMD %d: %6.2f cyc/inst seen by a wave, %5.2f cyc/inst per SIMD, clock %.2f GHz, wall %.3f ms, %.3f G inst/s/SIMD\n", wps,
           cyc_per_inst_wave, cyc_per_inst_simd, ghz, ms, rate);
    if (md) fprintf(md, " %.2f (%.2f GHz) |", cyc_per_inst_simd, ghz);
  }
  if (md) fprintf(md, "\n");
  hipFree(out);
  hipFree(stamps);
  return 0;
}

int main(int argc, char** argv) {
  FILE* md = argc > 1 ? fopen(argv[1], "w") : nullptr;
  if (md) {
    fprintf(md, "| stream (>=128 instructions per loop body, loop control < 2.5 %%) | 1 wave/SIMD | 2 | 3 | 4 | 8 |\n|---|---|---|---|---|---|\n");
  }
  run<FMA>(md);
  run<MIN2>(md);
  run<MIN3>(md);
  run<PKFMA>(md);
  run<MIN3_DEP>(md);
  run<MIN3_2CHAIN>(md);
  run<LDS_B64>(md);
  run<MIX_LDS_MIN3>(md);
  run<MIN3_SALU>(md);
  if (md) fclose(md);
  return 0;
}
