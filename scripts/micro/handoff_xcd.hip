// Micro-benchmark: latency of handing a tagged 16-byte slot from one workgroup to another INSIDE a launch, same XCD
// (blocks 0 and 8) and across XCDs (blocks 0 and 1), by the cache policy of the store and of the polling load.
// aux bits of the raw buffer builtins on gfx950: 1 = sc0, 2 = nt, 16 = sc1. Question behind it (k_dem_chain): which
// load flavour is served by the XCD's own L2 when the producer wrote the line with a plain store?
// Ping-pong: A stores tag i, B polls for it and answers with tag i, A polls for the answer: one iteration = 2 hops.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SAUX, int LAUX, bool ATOMIC>
__global__ __launch_bounds__(64) void k_pingpong(void* buf, unsigned bytes, int iters, int peer, long long* out, int* lost) {
  const int b = blockIdx.x;
  if (b != 0 && b != peer) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(buf, 0, (int)bytes, 0x00020000);
  const unsigned mine = (b == 0 ? 0u : 4096u) + threadIdx.x * 16u;     // 64 lanes x 16 B = 8 lines per side
  const unsigned theirs = (b == 0 ? 4096u : 0u) + threadIdx.x * 16u;
  unsigned long long* their64 = (unsigned long long*)((char*)buf + theirs);
  const long long t0 = wall_clock64();
  for (int i = 1; i <= iters; ++i) {
    if (b == 0) { u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine, 0, SAUX); }
    unsigned spins = 0;
    for (;;) {
      unsigned tag;
      if (ATOMIC) tag = (unsigned)__hip_atomic_fetch_or(their64, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, theirs, 0, LAUX); tag = v.x; }
      if (__all(tag == (unsigned)i)) break;
      if (++spins > (1u << 20)) { *lost = 1; return; }
    }
    if (b != 0) { u32x4 v = {(unsigned)i, (unsigned)i, (unsigned)i, (unsigned)i}; __builtin_amdgcn_raw_buffer_store_b128(v, rs, mine, 0, SAUX); }
  }
  if (b == 0 && threadIdx.x == 0) *out = wall_clock64() - t0;
}

template <int SAUX, int LAUX, bool ATOMIC>
int run(const char* name, void* buf, long long* out, int* lost) {
  const int iters = 2000;
  for (int peer : {8, 1}) {
    double best = 1e30;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemset(buf, 0, 8192)); CK(hipMemset(lost, 0, 4));
      hipLaunchKernelGGL((k_pingpong<SAUX, LAUX, ATOMIC>), dim3(16), dim3(64), 0, 0, buf, 8192u, iters, peer, out, lost);
      CK(hipDeviceSynchronize());
      long long t; int l; CK(hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&l, lost, 4, hipMemcpyDeviceToHost));
      if (l) { best = -1; break; }
      const double us = t * 0.01 / iters / 2;
      if (us < best) best = us;
    }
    if (best < 0) printf("%-34s %s: LOST (never seen)\n", name, peer == 8 ? "same XCD " : "cross XCD");
    else printf("%-34s %s: %.3f us per hop\n", name, peer == 8 ? "same XCD " : "cross XCD", best);
  }
  return 0;
}

int main() {
  void* buf; long long* out; int* lost;
  CK(hipMalloc(&buf, 8192)); CK(hipMalloc(&out, 8)); CK(hipMalloc(&lost, 4));
  run<0, 16, false>("store plain, load sc1", buf, out, lost);
  run<0, 2, false>("store plain, load nt", buf, out, lost);
  run<0, 17, false>("store plain, load sc0 sc1", buf, out, lost);
  run<0, 1, false>("store plain, load sc0", buf, out, lost);
  run<0, 0, true>("store plain, atomic or 0", buf, out, lost);
  run<16, 16, false>("store sc1, load sc1", buf, out, lost);
  run<16, 2, false>("store sc1, load nt", buf, out, lost);
  run<17, 17, false>("store sc0 sc1, load sc0 sc1", buf, out, lost);
  run<2, 2, false>("store nt, load nt", buf, out, lost);
  run<2, 16, false>("store nt, load sc1", buf, out, lost);
  run<16, 0, true>("store sc1, atomic or 0", buf, out, lost);
  return 0;
}
