// Issue/latency micro-benchmark for gfx950: cycles (s_memtime) per instruction of a LONE wavefront (one wave per SIMD)
// for dependent / independent fp64 and fp32 chains, exec-mask round trips and taken branches -- the ingredients of
// k_cs_march's instruction stream. Build: hipcc --offload-arch=gfx950 -O3 -o issue_latency issue_latency.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 256
template <int MODE, int REPS = REP>
__global__ void k(double* out, unsigned long long* cyc, double seed, int nloop) {
  double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  float b0 = (float)a0;
  const double c = seed * 0.5;
  int flag = threadIdx.x & 1;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < nloop; ++it) {
    if (__builtin_readcyclecounter() - t0 > 400000000ull) break;   // watchdog
#pragma unroll
    for (int r = 0; r < REPS / 8; ++r) {
      if (MODE == 0) {          // 8 dependent v_add_f64
        asm volatile("v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n"
                     "v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n v_add_f64 %0, %0, %1\n" : "+v"(a0) : "v"(c));
      } else if (MODE == 1) {   // 8 independent v_add_f64
        asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n"
                     "v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 2) {   // 8 dependent v_fma_f64
        asm volatile("v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n"
                     "v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n v_fma_f64 %0, %0, %1, %1\n" : "+v"(a0) : "v"(c));
      } else if (MODE == 3) {   // 8 dependent v_add_f32
        asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n"
                     "v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n" : "+v"(b0) : "v"((float)c));
      } else if (MODE == 4) {   // 8 independent v_mul_f64
        asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                     "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
      } else if (MODE == 5) {   // 4 x { v_cmp -> s_and_saveexec -> v_add_f64 -> s_or exec }: 16 instructions
        asm volatile(
            "v_cmp_eq_u32 vcc, 1, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
            "v_cmp_eq_u32 vcc, 1, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
            "v_cmp_eq_u32 vcc, 1, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
            "v_cmp_eq_u32 vcc, 1, %1\n s_and_saveexec_b64 s[20:21], vcc\n v_add_f64 %0, %0, %2\n s_or_b64 exec, exec, s[20:21]\n"
            : "+v"(a0) : "v"(flag), "v"(c) : "vcc", "scc", "s20", "s21");
      } else if (MODE == 6) {   // 8 independent v_cndmask_b32 (1-pass ops)
        asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n"
                     "v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n v_mov_b32 %0, %1\n" : "+v"(b0) : "v"(flag));
      } else if (MODE == 7) {   // 8 SALU
        asm volatile("s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n"
                     "s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n s_add_u32 s20, s20, 1\n" ::: "s20", "scc");
      } else if (MODE == 8) {   // 4 x { v_add_f64 ; taken branch over one instruction }: 8 instructions issued
        asm volatile(
            "v_add_f64 %0, %0, %1\n s_branch 1f\n v_add_f64 %0, %0, %1\n 1:\n"
            "v_add_f64 %0, %0, %1\n s_branch 2f\n v_add_f64 %0, %0, %1\n 2:\n"
            "v_add_f64 %0, %0, %1\n s_branch 3f\n v_add_f64 %0, %0, %1\n 3:\n"
            "v_add_f64 %0, %0, %1\n s_branch 4f\n v_add_f64 %0, %0, %1\n 4:\n" : "+v"(a0) : "v"(c));
      } else if (MODE == 9) {   // alternate VALU f64 / SALU, independent: 8 instructions
        asm volatile(
            "v_add_f64 %0, %0, %4\n s_add_u32 s20, s20, 1\n v_add_f64 %1, %1, %4\n s_add_u32 s21, s21, 1\n"
            "v_add_f64 %2, %2, %4\n s_add_u32 s20, s20, 1\n v_add_f64 %3, %3, %4\n s_add_u32 s21, s21, 1\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c) : "s20", "s21", "scc");
      } else if (MODE == 11) {  // 8 independent v_add_f32 in the 8-byte VOP3 encoding
        asm volatile("v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n"
                     "v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n v_add_f32_e64 %0, %0, %1\n" : "+v"(b0) : "v"((float)c));
      } else if (MODE == 12) {  // 8 x s_mov_b32 with a 32-bit literal (8 bytes each)
        asm volatile("s_mov_b32 s20, 0x12345678\n s_mov_b32 s21, 0x12345679\n s_mov_b32 s20, 0x1234567a\n s_mov_b32 s21, 0x1234567b\n"
                     "s_mov_b32 s20, 0x1234567c\n s_mov_b32 s21, 0x1234567d\n s_mov_b32 s20, 0x1234567e\n s_mov_b32 s21, 0x1234567f\n" ::: "s20", "s21");
      } else if (MODE == 13) {  // 4 x { v_add_f32 (4 bytes), s_mov_b32 literal (8 bytes) }: two pipes, 48 bytes per 8 instructions
        asm volatile("v_add_f32 %0, %0, %1\n s_mov_b32 s20, 0x12345678\n v_add_f32 %0, %0, %1\n s_mov_b32 s21, 0x12345679\n"
                     "v_add_f32 %0, %0, %1\n s_mov_b32 s20, 0x1234567a\n v_add_f32 %0, %0, %1\n s_mov_b32 s21, 0x1234567b\n" : "+v"(b0) : "v"((float)c) : "s20", "s21");
      } else if (MODE == 14) {  // 4 x { v_add_f32_e64 (8 bytes), s_mov_b32 literal (8 bytes) }: 64 bytes per 8 instructions
        asm volatile("v_add_f32_e64 %0, %0, %1\n s_mov_b32 s20, 0x12345678\n v_add_f32_e64 %0, %0, %1\n s_mov_b32 s21, 0x12345679\n"
                     "v_add_f32_e64 %0, %0, %1\n s_mov_b32 s20, 0x1234567a\n v_add_f32_e64 %0, %0, %1\n s_mov_b32 s21, 0x1234567b\n" : "+v"(b0) : "v"((float)c) : "s20", "s21");
      } else if (MODE == 10) {  // v_cmp (VALU writes vcc) -> v_cndmask (reads vcc) dependent pairs: 8 instructions
        asm volatile(
            "v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n"
            "v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f64 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n"
            : "+v"(a0) : "v"(c), "v"(b0), "v"(flag) : "vcc");
      }
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE, int REPS = REP>
void run(const char* name, int waves_per_simd) {
  const int blocks = 256, nloop = 200 * REP / REPS;
  double* out; unsigned long long* cyc;
  hipMalloc(&out, sizeof(double) * blocks * 1024); hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
  const int threads = 256 * waves_per_simd;   // one workgroup per CU: waves_per_simd waves on each SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, REPS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0, nloop);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, REPS>), dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0, nloop);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  const double ninstr = (double)nloop * REPS * (MODE == 5 ? 2 : 1);
  fflush(stdout);
  printf("%-44s waves/SIMD %d: %7.2f memtime-ticks/instr, %7.2f ns/instr (event time %.3f ms)\n", name, waves_per_simd,
         mean / ninstr, 1e6 * ms / ninstr, ms);
  fflush(stdout);
  hipFree(out); hipFree(cyc);
}

int main() {
  // code footprint: the same instruction mix as a 2 KB, 8 KB, 40 KB and 80 KB loop body (the instruction cache is 64 KB per two CUs)
  for (int w = 1; w <= 3; ++w) {
    run<14, 256>("8-byte mix, 2 KB loop", w);
    run<14, 1024>("8-byte mix, 8 KB loop", w);
    run<14, 5120>("8-byte mix, 40 KB loop", w);
    run<14, 10240>("8-byte mix, 80 KB loop", w);
    run<13, 5120>("6-byte mix, 30 KB loop", w);
    run<3, 5120>("v_add_f32 4 bytes, 20 KB loop", w);
    run<0, 5120>("v_add_f64 8 bytes, 40 KB loop", w);
  }
  // instruction-fetch bandwidth: the same arithmetic in 4-byte and 8-byte encodings, 1-3 waves per SIMD
  for (int w = 1; w <= 3; ++w) {
    run<3>("v_add_f32 (VOP2, 4 bytes)", w);
    run<11>("v_add_f32_e64 (VOP3, 8 bytes)", w);
    run<12>("s_mov_b32 literal (8 bytes)", w);
    run<13>("v_add_f32 + s_mov literal (6 bytes avg)", w);
    run<14>("v_add_f32_e64 + s_mov literal (8 bytes avg)", w);
    run<0>("dependent v_add_f64 (8 bytes)", w);
    run<9>("alternating v_add_f64 / s_add_u32 (6 bytes avg)", w);
  }
  for (int w = 1; w <= 2; ++w) {
    run<0>("dependent v_add_f64", w);
    run<1>("independent v_add_f64 (8 chains)", w);
    run<2>("dependent v_fma_f64", w);
    run<4>("independent v_mul_f64 (8 chains)", w);
    run<3>("dependent v_add_f32", w);
    run<6>("v_mov_b32", w);
    run<7>("dependent s_add_u32", w);
    run<9>("alternating v_add_f64 / s_add_u32", w);
    run<10>("v_cmp_f64 -> v_cndmask pairs", w);
    run<5>("v_cmp/saveexec/v_add_f64/s_or (per instr)", w);
    run<8>("v_add_f64 + taken s_branch (per issued instr)", w);
  }
  return 0;
}
