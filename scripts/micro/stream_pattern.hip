// Micro-benchmark: how fast can 9 fp64 planes be read and written with the access patterns considered
// for the fused collide+stream kernel? No arithmetic beyond a dependency-keeping add.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// A: one wave = 64-wide window (62 writers, offset -1), marches LX rows, prefetch 1 row ahead
// LAYOUT 0: nine planes f[q][x][y]; 1: tiles of 16 y-nodes, f[x][y/16][q][y%16] (the nine 128-byte lines of a tile
// are contiguous: a wave-row touches 2 DRAM regions instead of 18)
template <int LAYOUT>
__device__ __forceinline__ long fidx(int q, int x, int y, int ly, long plane) {
  if (LAYOUT == 0) return q * plane + (long)x * ly + y;
  return ((((long)x * (ly >> 4) + (y >> 4)) * 9 + q) << 4) + (y & 15);
}

template <int LX, int USE, int OFF, int WORK = 0, int MINW = 1, int LAYOUT = 0, int SYNC = 0>
__global__ __launch_bounds__(256, MINW) void k_march(const double* __restrict__ fin, double* __restrict__ fout, int lx, int ly,
                                               long plane, int nstrips, int nwork, int remap) {
  const int lane = threadIdx.x & 63;
  int blk = blockIdx.x;
  if (remap) { const int per = gridDim.x >> 3; blk = (blk & 7) * per + (blk >> 3); }
  const int w = blk * 4 + (threadIdx.x >> 6);
  if (w >= nwork) return;
  const int strip = w % nstrips, seg = w / nstrips;
  const int y = strip * USE - OFF + lane;
  const bool yin = y >= 0 && y < ly;
  const bool writer = lane >= OFF && lane < OFF + USE && yin;
  const int xs = seg * LX, xe = min(xs + LX, lx);
  double cur[9], nxt[9];
  for (int q = 0; q < 9; ++q) cur[q] = yin ? fin[fidx<LAYOUT>(q, xs, y, ly, plane)] : 0.0;
  for (int x = xs; x < xe; ++x) {
    const int xn = x + 1 < lx ? x + 1 : x;
    for (int q = 0; q < 9; ++q) nxt[q] = yin ? fin[fidx<LAYOUT>(q, xn, y, ly, plane)] : 0.0;
    // synthetic VALU load: WORK x 9 dependent-per-q fp64 FMAs (independent across q)
    for (int k = 0; k < WORK; ++k)
      for (int q = 0; q < 9; ++q) cur[q] = cur[q] * 1.0000001 + 0.5;
    if (SYNC) {  // the four waves of a workgroup exchange their seam columns through LDS once per row
      __shared__ double seam[4][2][3];
      const int wv = threadIdx.x >> 6;
      if (lane == 0) { seam[wv][0][0] = cur[1]; seam[wv][0][1] = cur[2]; seam[wv][0][2] = cur[3]; }
      if (lane == 63) { seam[wv][1][0] = cur[5]; seam[wv][1][1] = cur[6]; seam[wv][1][2] = cur[7]; }
      __syncthreads();
      if (lane == 0 && wv > 0) cur[0] += seam[wv - 1][1][0] * 1e-30;
      if (lane == 63 && wv < 3) cur[0] += seam[wv + 1][0][0] * 1e-30;
      __syncthreads();
    }
    if (writer) for (int q = 0; q < 9; ++q) fout[fidx<LAYOUT>(q, x, y, ly, plane)] = cur[q] + 1.0;
    for (int q = 0; q < 9; ++q) cur[q] = nxt[q];
  }
}

// B: plain grid-stride copy, 16 B per lane
__global__ void k_copy16(const double2* __restrict__ in, double2* __restrict__ out, long n2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += (long)gridDim.x * blockDim.x) {
    double2 v = in[i]; v.x += 1.0; v.y += 1.0; out[i] = v;
  }
}
// C: plain grid-stride copy, 8 B per lane
__global__ void k_copy8(const double* __restrict__ in, double* __restrict__ out, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i] + 1.0;
}

template <class F> float timeit(F f, int reps = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main() {
  const int lx = 4096, ly = 4096; const long plane = (long)lx * ly; const long n = 9 * plane;
  double *a, *b; CK(hipMalloc(&a, n * 8)); CK(hipMalloc(&b, n * 8)); CK(hipMemset(a, 0, n * 8)); CK(hipMemset(b, 0, n * 8));
  const double gb = 2.0 * n * 8 / 1e9;
  auto rep = [&](const char* name, float ms) { printf("%-48s %.3f ms  %.0f GB/s\n", name, ms, gb / ms * 1e3); };
  rep("copy 16B/lane grid-stride 2048x256", timeit([&] { hipLaunchKernelGGL(k_copy16, dim3(2048), dim3(256), 0, 0, (double2*)a, (double2*)b, n / 2); }));
  rep("copy 8B/lane grid-stride 2048x256", timeit([&] { hipLaunchKernelGGL(k_copy8, dim3(2048), dim3(256), 0, 0, a, b, n); }));
  rep("copy 8B/lane grid-stride 8192x256", timeit([&] { hipLaunchKernelGGL(k_copy8, dim3(8192), dim3(256), 0, 0, a, b, n); }));
#define MARCH(LX, USE, OFF, REMAP, NAME) { const int ns = (ly + USE - 1) / USE, nseg = (lx + LX - 1) / LX, nw = ns * nseg; int g = (nw + 3) / 4; if (REMAP) g = (g + 7) / 8 * 8; \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_march<LX, USE, OFF>), dim3(g), dim3(256), 0, 0, a, b, lx, ly, plane, ns, nw, REMAP); })); }
  MARCH(16, 62, 1, 1, "march LX16 62/64 off-1 remap");
  MARCH(16, 62, 1, 0, "march LX16 62/64 off-1 noremap");
  MARCH(16, 64, 0, 1, "march LX16 64/64 aligned remap");
  MARCH(16, 64, 0, 0, "march LX16 64/64 aligned noremap");
  MARCH(64, 64, 0, 1, "march LX64 64/64 aligned remap");
  MARCH(8, 64, 0, 1, "march LX8 64/64 aligned remap");
  MARCH(32, 62, 1, 1, "march LX32 62/64 off-1 remap");
#define MARCHW(LX, WORK, MINW, NAME) { const int ns = (ly + 61) / 62, nseg = (lx + LX - 1) / LX, nw = ns * nseg; int g = ((nw + 3) / 4 + 7) / 8 * 8; \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_march<LX, 62, 1, WORK, MINW>), dim3(g), dim3(256), 0, 0, a, b, lx, ly, plane, ns, nw, 1); })); }
  MARCHW(16, 28, 8, "march LX16 62/64 +252 FMA/row  8 waves/SIMD");
  MARCHW(16, 56, 8, "march LX16 62/64 +504 FMA/row  8 waves/SIMD");
  MARCHW(16, 112, 8, "march LX16 62/64 +1008 FMA/row 8 waves/SIMD");
  MARCHW(16, 56, 2, "march LX16 62/64 +504 FMA/row  (minw 2)");
  MARCHW(16, 112, 2, "march LX16 62/64 +1008 FMA/row (minw 2)");
  // layout comparison at the occupancy of the real kernel: SHM bytes of dynamic LDS per workgroup limit the CU to 2 workgroups
#define MARCHL(LX, LAYOUT, SHM, NAME) { const int ns = (ly + 61) / 62, nseg = (lx + LX - 1) / LX, nw = ns * nseg; int g = ((nw + 3) / 4 + 7) / 8 * 8; \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_march<LX, 62, 1, 0, 1, LAYOUT>), dim3(g), dim3(256), SHM, 0, a, b, lx, ly, plane, ns, nw, 1); })); }
  MARCHL(32, 0, 0, "march LX32 62/64 planes            full occupancy");
  MARCHL(32, 1, 0, "march LX32 62/64 16-node tiles     full occupancy");
  MARCHL(32, 0, 76800, "march LX32 62/64 planes            2 workgroups/CU");
  MARCHL(32, 1, 76800, "march LX32 62/64 16-node tiles     2 workgroups/CU");
  MARCHL(16, 0, 76800, "march LX16 62/64 planes            2 workgroups/CU");
  MARCHL(16, 1, 76800, "march LX16 62/64 16-node tiles     2 workgroups/CU");
  // producing lanes per window, tile layout, the real kernel's occupancy: 62 (round 1-4), 60 (stores on 32-byte sector boundaries), 56 (64-byte)
#define MARCHU(LX, USE, OFF, SHM, NAME) { const int ns = (ly + USE - 1) / USE, nseg = (lx + LX - 1) / LX, nw = ns * nseg; int g = ((nw + 3) / 4 + 7) / 8 * 8; \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_march<LX, USE, OFF, 0, 1, 1>), dim3(g), dim3(256), SHM, 0, a, b, lx, ly, plane, ns, nw, 1); })); }
  MARCHU(32, 62, 1, 76800, "march LX32 62/64 off-1 tiles       2 workgroups/CU");
  MARCHU(32, 60, 2, 76800, "march LX32 60/64 off-2 tiles       2 workgroups/CU");
  MARCHU(32, 56, 4, 76800, "march LX32 56/64 off-4 tiles       2 workgroups/CU");
  MARCHU(32, 62, 1, 76800, "march LX32 62/64 off-1 tiles       2 workgroups/CU (again)");
  MARCHU(32, 60, 2, 76800, "march LX32 60/64 off-2 tiles       2 workgroups/CU (again)");
  // aligned 64-lane windows (idea: seam columns through LDS), with and without a workgroup barrier per row
#define MARCHS(LX, SYNC, SHM, NAME) { const int ns = ly / 64, nseg = (lx + LX - 1) / LX, nw = ns * nseg; int g = ((nw + 3) / 4 + 7) / 8 * 8; \
    rep(NAME, timeit([&] { hipLaunchKernelGGL((k_march<LX, 64, 0, 0, 1, 1, SYNC>), dim3(g), dim3(256), SHM, 0, a, b, lx, ly, plane, ns, nw, 1); })); }
  MARCHS(32, 0, 76800, "march LX32 64/64 tiles aligned          2 workgroups/CU");
  MARCHS(32, 1, 76800, "march LX32 64/64 tiles aligned + 2 barriers/row  2 wg/CU");
  MARCHS(16, 0, 76800, "march LX16 64/64 tiles aligned          2 workgroups/CU");
  MARCHS(16, 1, 76800, "march LX16 64/64 tiles aligned + 2 barriers/row  2 wg/CU");
  return 0;
}
