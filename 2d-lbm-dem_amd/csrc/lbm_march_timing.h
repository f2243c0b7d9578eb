// lbm_march_timing.h -- EXPERIMENT BUILD ONLY (make AB=1 ABFLAGS=-DMARCH_TIMING=k): per-phase cycle timers of k_cs_march's
// iteration. The product build defines the three markers MT_DECL / MT(i) / MT_FLUSH as nothing (lbm_fused.hip).
#pragma once

namespace {
// MARCH_TIMING = k: only phase k is timed (one accumulator and one pending stamp: the kernel has no registers to spare --
// with all eight phases timed at once it spills and runs twice as long); phase k lies between boundary
// MT(k == 0 ? 7 : k - 1) and boundary MT(k); a stamp is only consumed at the end of its phase, so it adds no s_waitcnt
// of its own in between. scripts/march_timing.py reads the sums.
__device__ unsigned long long g_march_t[16];
#define MT_FROM (MARCH_TIMING == 0 ? 7 : MARCH_TIMING - 1)
#define MT_DECL unsigned long long mt_acc_ = 0, mt_from_ = __builtin_readcyclecounter(); const unsigned long long mt_start_ = mt_from_;
#define MT(i) { if ((i) == MT_FROM) { __builtin_amdgcn_sched_barrier(0); mt_from_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } \
                if ((i) == MARCH_TIMING) { __builtin_amdgcn_sched_barrier(0); mt_acc_ += __builtin_readcyclecounter() - mt_from_; __builtin_amdgcn_sched_barrier(0); } }
#define MT_FLUSH if (lane == 0) { atomicAdd(&g_march_t[MARCH_TIMING], mt_acc_); atomicAdd(&g_march_t[8], __builtin_readcyclecounter() - mt_start_); atomicAdd(&g_march_t[9], 1ull); }
extern "C" __attribute__((visibility("default"))) int lbmdem_ab_march_timing(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_t), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_march_t), z, sizeof z) == hipSuccess ? 0 : -1;
}
}  // namespace
