// Hardware probe for two gfx950 primitives k_cs_march3 relies on:
//  (1) full-wave DPP shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1) across all 64 lanes,
//  (2) global_load_lds_dwordx4: LDS image = M0 base + 16 * lane, per-lane global source, partial EXEC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_dn1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__global__ void k_dpp(int* up, int* dn, int* upo, int* dno) {
  const int l = threadIdx.x;
  up[l] = dpp_up1(l * 3 + 1);
  dn[l] = dpp_dn1(l * 3 + 1);
  upo[l] = __builtin_amdgcn_update_dpp(-7, l * 3 + 1, 0x138, 0xf, 0xf, false);   // end lane takes `old`
  dno[l] = __builtin_amdgcn_update_dpp(-9, l * 3 + 1, 0x130, 0xf, 0xf, false);
}
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k_dma(const double* rec, const int* ids, double* out, int T) {
  __shared__ double2 img[2][4 * 64];
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int k = l; k < 256; k += 64) img[wv][k] = make_double2(-1.0, -1.0);
  __syncthreads();
  const unsigned base = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)&img[wv][0]);
  if (l < T) {
    const char* src = reinterpret_cast<const char*>(rec) + (long)ids[wv * 64 + l] * 64;
    lds_dma16(src, base);
    lds_dma16(src + 16, base + 1024);
    lds_dma16(src + 32, base + 2048);
    lds_dma16(src + 48, base + 3072);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int p = 0; p < 4; ++p) {
    out[((wv * 64 + l) * 4 + p) * 2 + 0] = img[wv][p * 64 + l].x;
    out[((wv * 64 + l) * 4 + p) * 2 + 1] = img[wv][p * 64 + l].y;
  }
}
int main() {
  int *d; hipMalloc(&d, 4 * 64 * 4);
  hipLaunchKernelGGL(k_dpp, dim3(1), dim3(64), 0, 0, d, d + 64, d + 128, d + 192);
  int h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int eu = l == 0 ? 1 : (l - 1) * 3 + 1, ed = l == 63 ? 63 * 3 + 1 : (l + 1) * 3 + 1;
    const int euo = l == 0 ? -7 : (l - 1) * 3 + 1, edo = l == 63 ? -9 : (l + 1) * 3 + 1;
    if (h[l] != eu || h[64 + l] != ed || h[128 + l] != euo || h[192 + l] != edo) { ++bad; printf("dpp lane %d: %d %d %d %d\n", l, h[l], h[64 + l], h[128 + l], h[192 + l]); }
  }
  printf("dpp wave shifts: %s\n", bad ? "MISMATCH" : "ok");
  const int n = 1000;
  double* hrec = new double[n * 8];
  for (int i = 0; i < n * 8; ++i) hrec[i] = i * 0.5;
  int hid[128]; for (int i = 0; i < 128; ++i) hid[i] = (i * 37 + 11) % n;
  double *drec, *dout; int* dids;
  hipMalloc(&drec, n * 64); hipMalloc(&dout, 128 * 8 * 8); hipMalloc(&dids, 128 * 4);
  hipMemcpy(drec, hrec, n * 64, hipMemcpyHostToDevice); hipMemcpy(dids, hid, 128 * 4, hipMemcpyHostToDevice);
  for (int T : {64, 20, 1, 0}) {
    hipLaunchKernelGGL(k_dma, dim3(1), dim3(128), 0, 0, drec, dids, dout, T);
    double ho[128 * 8]; hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
    int b2 = 0;
    for (int t = 0; t < 128; ++t) for (int j = 0; j < 8; ++j) {
      const double e = (t & 63) < T ? hrec[hid[t] * 8 + j] : -1.0;
      if (ho[t * 8 + j] != e) { if (b2 < 5) printf("dma T=%d slot %d field %d: got %g want %g\n", T, t, j, ho[t * 8 + j], e); ++b2; }
    }
    printf("lds dma, %d active lanes: %s\n", T, b2 ? "MISMATCH" : "ok");
  }
  return 0;
}
