// lbm_march_timing.h -- EXPERIMENT BUILD ONLY (make AB=1 ABFLAGS=-DMARCH_TIMING=k): per-phase cycle timers of k_cs_march's
// iteration. The product build defines the three markers MT_DECL / MT(i) / MT_FLUSH as nothing (lbm_fused.hip).
// make AB=1 ABFLAGS=-DMARCH_TRACE: instead, every wavefront logs the constant-rate clock (100 MHz, the same on all XCDs) at
// the head of each of its rows into a buffer the caller provides (scripts/march_trace.py): where and when rows are slow.
#pragma once

#ifdef MARCH_TRACE
namespace {
// per wavefront `stride` words: [0] start (ticks, low 32 bits), [1] window << 20 | XCC_ID << 16 | HW_ID bits, [2] first row, [3] rows logged,
// [4 + k] ticks since the start at the head of the k-th iteration, [4 + rows] at the end
__device__ unsigned* g_trace_buf;
__device__ unsigned g_trace_waves, g_trace_stride;
#define MT_DECL \
  unsigned* tr_p_ = (g_trace_buf && (unsigned)w < g_trace_waves) ? g_trace_buf + (size_t)w * g_trace_stride : nullptr; \
  const unsigned long long tr_t0_ = wall_clock64();                                                                    \
  if (tr_p_ && lane == 0) {                                                                                             \
    unsigned hw_, xcc_;                                                                                                 \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));                                                   \
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));                                                 \
    tr_p_[0] = (unsigned)tr_t0_; tr_p_[1] = (unsigned)strip << 20 | (xcc_ & 15u) << 16 | (hw_ & 0xFFFFu); tr_p_[2] = (unsigned)xs; tr_p_[3] = 0u; \
  }
#define MT(i) { if ((i) == 7 && tr_p_ && lane == 0 && (unsigned)(x - xs) + 5u < g_trace_stride) \
                  tr_p_[4 + (x - xs)] = (unsigned)(wall_clock64() - tr_t0_); }
#define MT_FLUSH if (tr_p_ && lane == 0) { const unsigned n_ = (unsigned)(xe - xs); tr_p_[3] = n_; \
                   if (n_ + 5u <= g_trace_stride) tr_p_[4 + n_] = (unsigned)(wall_clock64() - tr_t0_); }
extern "C" __attribute__((visibility("default"))) int lbmdem_ab_march_trace(unsigned* dev, unsigned waves, unsigned stride) {
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_buf), &dev, sizeof dev) != hipSuccess) return -1;
  if (hipMemcpyToSymbol(HIP_SYMBOL(g_trace_waves), &waves, sizeof waves) != hipSuccess) return -1;
  return hipMemcpyToSymbol(HIP_SYMBOL(g_trace_stride), &stride, sizeof stride) == hipSuccess ? 0 : -1;
}
}  // namespace
#else

namespace {
// MARCH_TIMING = k: only phase k is timed (one accumulator and one pending stamp: the kernel has no registers to spare --
// with all eight phases timed at once it spills and runs twice as long); phase k lies between boundary
// MT(k == 0 ? 7 : k - 1) and boundary MT(k); a stamp is only consumed at the end of its phase, so it adds no s_waitcnt
// of its own in between. scripts/march_timing.py reads the sums.
__device__ unsigned long long g_march_t[16];
// boundaries in program order: 7 (loop head), 0, 1, 2, 3, 4, 10 (links compacted), 11 (links evaluated), 5, 6
#define MT_FROM (MARCH_TIMING == 0 ? 7 : MARCH_TIMING == 10 ? 4 : MARCH_TIMING == 11 ? 10 : MARCH_TIMING == 5 ? 11 : MARCH_TIMING - 1)
#define MT_DECL unsigned long long mt_acc_ = 0, mt_from_ = __builtin_readcyclecounter(); const unsigned long long mt_start_ = mt_from_;
#define MT(i) { if ((i) == MT_FROM || ((i) == 12 && MARCH_TIMING == 5)) {   /* (12: before the links' loop, which a row without links skips) */ __builtin_amdgcn_sched_barrier(0); mt_from_ = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); } \
                if ((i) == MARCH_TIMING) { __builtin_amdgcn_sched_barrier(0); mt_acc_ += __builtin_readcyclecounter() - mt_from_; __builtin_amdgcn_sched_barrier(0); } }
#define MT_FLUSH if (lane == 0) { atomicAdd(&g_march_t[MARCH_TIMING], mt_acc_); atomicAdd(&g_march_t[8], __builtin_readcyclecounter() - mt_start_); atomicAdd(&g_march_t[9], 1ull); }
extern "C" __attribute__((visibility("default"))) int lbmdem_ab_march_timing(unsigned long long* out) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_march_t), sizeof(unsigned long long) * 16) != hipSuccess) return -1;
  unsigned long long z[16] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_march_t), z, sizeof z) == hipSuccess ? 0 : -1;
}
}  // namespace
#endif  // MARCH_TRACE
