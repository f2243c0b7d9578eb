// Micro-benchmark: what does a grid-wide barrier inside one cooperative kernel cost on this GPU, next to the
// ~4.4 us of a dependent kernel launch? (Question behind it: would the 12 DEM sub-steps between two fluid steps be
// cheaper as ONE persistent kernel with a barrier per sub-step?)
// Each "sub-step": every thread reads one double written by ANOTHER workgroup in the previous sub-step (so the
// barrier has to carry real cross-XCD visibility), adds 1, writes it to the other buffer; then the barrier.
// Spins are bounded: a lost barrier sets a flag instead of hanging the box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned* counter, unsigned target, int* lost) {
  __syncthreads();
  bool ok = true;
  if (threadIdx.x == 0) {
    __atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE);   // agent scope by default for global atomics in HIP
    long spins = 0;
    while (__atomic_load_n(counter, __ATOMIC_ACQUIRE) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > (1L << 22)) { *lost = 1; ok = false; break; }
    }
  }
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(256) void k_steps(double* a, double* b, int n, int steps, unsigned* counter, int* lost) {
  const int nb = gridDim.x;
  for (int s = 0; s < steps; ++s) {
    const double* in = (s & 1) ? b : a;
    double* out = (s & 1) ? a : b;
    // element handled: this block's slot; source: the slot of the block "opposite" in the grid
    const int src_blk = (blockIdx.x + nb / 2 + 1) % nb;
    for (int k = threadIdx.x; k < n / nb; k += blockDim.x)
      out[(long)blockIdx.x * (n / nb) + k] = in[(long)src_blk * (n / nb) + k] + 1.0;
    if (!grid_barrier(counter, (unsigned)(s + 1) * nb, lost)) return;
  }
}

__global__ __launch_bounds__(256) void k_one(const double* in, double* out, int n) {
  const int nb = gridDim.x;
  const int src_blk = (blockIdx.x + nb / 2 + 1) % nb;
  for (int k = threadIdx.x; k < n / nb; k += blockDim.x)
    out[(long)blockIdx.x * (n / nb) + k] = in[(long)src_blk * (n / nb) + k] + 1.0;
}

int main() {
  const int steps = 120;
  for (int nb : {64, 128, 256, 512, 782, 1024}) {
    const int per = 64;           // doubles per workgroup and sub-step: a DEM tile's worth of state
    const int n = nb * per;
    double *a, *b; unsigned* counter; int* lost;
    CK(hipMalloc(&a, sizeof(double) * n)); CK(hipMalloc(&b, sizeof(double) * n));
    CK(hipMalloc(&counter, 4)); CK(hipMalloc(&lost, 4));
    CK(hipMemset(a, 0, sizeof(double) * n)); CK(hipMemset(b, 0, sizeof(double) * n));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    int maxb = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&maxb, k_steps, 256, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    if ((long)maxb * prop.multiProcessorCount < nb) { printf("nb %d does not fit (%d x %d)\n", nb, maxb, prop.multiProcessorCount); continue; }
    float best_c = 1e9f, best_l = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      CK(hipMemset(counter, 0, 4)); CK(hipMemset(lost, 0, 4));
      int st = steps; int nn = n;
      void* args[] = {&a, &b, &nn, &st, &counter, &lost};
      CK(hipEventRecord(e0));
      CK(hipLaunchCooperativeKernel((const void*)k_steps, dim3(nb), dim3(256), args, 0, nullptr));
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_c) best_c = ms;
      CK(hipEventRecord(e0));
      for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(k_one, dim3(nb), dim3(256), 0, 0, (s & 1) ? b : a, (s & 1) ? a : b, n);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best_l) best_l = ms;
    }
    int hl = 0; CK(hipMemcpy(&hl, lost, 4, hipMemcpyDeviceToHost));
    std::vector<double> h(n); CK(hipMemcpy(h.data(), a, sizeof(double) * n, hipMemcpyDeviceToHost));
    // both variants ran `steps` increments per repetition on the same buffers: 10 runs x 120
    bool good = true; for (int k = 0; k < n; ++k) if (h[k] != 10.0 * steps) { good = false; break; }
    printf("workgroups %4d: persistent + barrier %.2f us/sub-step, separate launches %.2f us/sub-step, lost %d, values %s (%.0f)\n",
           nb, 1e3 * best_c / steps, 1e3 * best_l / steps, hl, good ? "ok" : "WRONG", h[0]);
    (void)hipFree(a); (void)hipFree(b); (void)hipFree(counter); (void)hipFree(lost);
  }
  return 0;
}
