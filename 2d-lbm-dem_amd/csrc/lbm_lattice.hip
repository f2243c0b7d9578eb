// lbm_lattice.hip -- everything else that touches the populations: host layout <-> device layout, initial density,
// macroscopic fields, the total density (tree and the reference's serial chain), halo rows, the VTK fields.

#include "lbm_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// layout conversion, diagnostics, halo packing
// ---------------------------------------------------------------------------------------------

// host AoS rows [nxl][ly][9] (reference layout, main.c:56) -> device planes. One thread per
// (node, q) element read coalesced from the AoS side through LDS-free index math; init-time only.
__global__ void k_aos_to_soa(const real* __restrict__ aos, real* __restrict__ f, LatticeView L) {
  const long total = (long)L.nxl * L.ly * 9;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k % 9);
    const long node = k / 9;
    const int y = (int)(node % L.ly), xl = (int)(node / L.ly);
    f[fidx(q, (long)xl * L.sy + y)] = aos[k];
  }
}
__global__ void k_soa_to_aos(const real* __restrict__ f, real* __restrict__ aos, LatticeView L, int xl0,
                             int nrows) {
  const long total = (long)nrows * L.ly * 9;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k % 9);
    const long node = k / 9;
    const int y = (int)(node % L.ly), xr = (int)(node / L.ly);
    aos[k] = f[fidx(q, (long)(xl0 + xr) * L.sy + y)];
  }
}

// init_density (main.c:716-724): f = w[q] everywhere
__global__ void k_fill_equilibrium(real* __restrict__ f, LatticeView L) {
  const long total = 9 * L.plane;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)((k % (9 * LBMDEM_TILE_Y)) / LBMDEM_TILE_Y);  // f[tile][q][TILE_Y]
    f[k] = q == 0 ? 4. / 9 : ((q & 1) ? 1. / 36 : 1. / 9);
  }
}

// rho, rho*u sums in the order write_vtk forms them (main.c:315-319)
__global__ void k_macro(const real* __restrict__ f, LatticeView L, int xl0, int nrows,
                        real* __restrict__ rho, real* __restrict__ ux, real* __restrict__ uy) {
  const long total = (long)nrows * L.ly;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int y = (int)(k % L.ly), xr = (int)(k / L.ly);
    const long node = (long)(xl0 + xr) * L.sy + y;
    real s = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const real v = f[fidx(q, node)];
      s += v;
      sx += v * EXq(q);
      sy += v * EYq(q);
    }
    rho[k] = s; ux[k] = sx; uy[k] = sy;
  }
}

// Total mass, per-block partial sums over the owned rows (check_density, main.c:1249-1261).
// Summation order differs from the reference's serial sweep; compared with a tolerance.
__global__ void k_density_partial(const real* __restrict__ f, LatticeView L, double* __restrict__ partial) {
  __shared__ double red[256];
  const long rows = L.xo1 - L.xo0;
  const long total = rows * L.ly;
  double s = 0.0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int y = (int)(k % L.ly), xr = (int)(k / L.ly);
    const long node = (long)(L.xo0 + xr) * L.sy + y;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += f[fidx(q, node)];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- the reference's SERIAL total density (main.c:1249-1273: sum = sum + f[x][y][q], x outer, y, q inner) -----------
// A serial floating-point sum is a chain, but while the running sum s stays inside one binade [2^k, 2^(k+1)) every
// addition of a positive a rounds to the same quantum u = 2^(k-52): s is a multiple of u, so RN(s + a) = s + RN_u(a),
// where RN_u(a) -- a rounded to a multiple of u -- does not depend on s unless a / u falls exactly half-way between two
// integers (then the tie goes to the even multiple: depends on s). Hence for one lattice row whose additions all
// happen in binade k and which holds no tie, no non-positive and no over-large value, the chain adds exactly
// (sum of the integers n = RN(a / u)) * u -- and integer sums associate. One workgroup per row forms that integer
// sum and the flags; the host walks the rows with the exact running sum and replays a row element by element (in
// the reference's order) whenever the shortcut does not apply: the rows where the sum crosses a power of two (~13 of
// 4096 at 4096^2), tie rows (~1), and whatever the first pass could not classify.
__global__ void k_density_rowsum(const real* __restrict__ f, LatticeView L, double* __restrict__ rowsum) {
  __shared__ double red[256];
  const long row = L.xo0 + blockIdx.x;
  double s = 0.0;
  for (int y = threadIdx.x; y < L.ly; y += blockDim.x) {
    const long node = row * L.sy + y;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += f[fidx(q, node)];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) rowsum[blockIdx.x] = red[0];
}

// quanta[row] = sum over the row of RN(a / u), u = 2^(kexp[row] - (p - 1)) the quantum of a running sum in binade kexp[row]
// (p = LBMDEM_REAL_MANT significand bits of `real`); flags[row] != 0: the shortcut does not apply
__global__ void k_density_rowquanta(const real* __restrict__ f, LatticeView L, const int* __restrict__ kexp,
                                    unsigned long long* __restrict__ quanta, int* __restrict__ flags) {
  __shared__ unsigned long long red[256];
  __shared__ int bad;
  if (threadIdx.x == 0) bad = 0;
  __syncthreads();
  const long row = L.xo0 + blockIdx.x;
  const int k = kexp[blockIdx.x];
  const double top = ldexp(1.0, k + 1);   // a >= 2^(k+1) would leave the binade on its own
  unsigned long long n = 0;
  int mybad = 0;
  for (int y = threadIdx.x; y < L.ly; y += blockDim.x) {
    const long node = row * L.sy + y;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const double a = f[fidx(q, node)];   // (a float is a double: the quantum arithmetic below is exact in double for either type)
      if (!(a > 0.0) || !(a < top)) { mybad = 1; continue; }   // also NaN
      const double t = ldexp(a, (LBMDEM_REAL_MANT - 1) - k);   // a / u, exact (a power-of-two scaling; a tiny a may underflow to 0: n = 0, right)
      const double fl = floor(t);
      if (t - fl == 0.5) mybad = 1;        // a tie: the rounding depends on the running sum
      n += (unsigned long long)rint(t);
    }
  }
  if (mybad) bad = 1;
  red[threadIdx.x] = n;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) { quanta[blockIdx.x] = red[0]; flags[blockIdx.x] = bad; }
}

// halo rows <-> contiguous buffer [9][nrows][ly]; blockIdx.y = side: rows from xl0a (low) / xl0b (high), a null buffer
// skips the side
__global__ void k_halo_pack(const real* __restrict__ f, LatticeView L, int xl0a, int xl0b, int nrows,
                            real* __restrict__ bufa, real* __restrict__ bufb) {
  const int xl0 = blockIdx.y ? xl0b : xl0a;
  real* __restrict__ buf = blockIdx.y ? bufb : bufa;
  if (!buf) return;
  const long per = (long)nrows * L.ly, total = 9 * per;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k / per);
    const long r = k % per;
    const int y = (int)(r % L.ly), xr = (int)(r / L.ly);
    buf[k] = f[fidx(q, (long)(xl0 + xr) * L.sy + y)];
  }
}
__global__ void k_halo_unpack(real* __restrict__ f, LatticeView L, int xl0a, int xl0b, int nrows,
                              const real* __restrict__ bufa, const real* __restrict__ bufb) {
  const int xl0 = blockIdx.y ? xl0b : xl0a;
  const real* __restrict__ buf = blockIdx.y ? bufb : bufa;
  if (!buf) return;
  const long per = (long)nrows * L.ly, total = 9 * per;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k / per);
    const long r = k % per;
    const int y = (int)(r % L.ly), xr = (int)(r / L.ly);
    f[fidx(q, (long)(xl0 + xr) * L.sy + y)] = buf[k];
  }
}

// The five fields write_vtk builds per node (main.c:284-323), float32, in [y][x] order (x fastest), for
// the owned rows (x offset = first owned row). Fluid sums are accumulated in FLOAT with a real
// intermediate per addition, exactly as `float += real` does in the reference.
__global__ void k_vtk_fields(const real* __restrict__ f, const int* __restrict__ obst, LatticeView L,
                             const real* __restrict__ gp, const real* __restrict__ v1,
                             const real* __restrict__ v2, const real* __restrict__ a1,
                             const real* __restrict__ a2, real rho_moy, float* __restrict__ grain_pressure,
                             float* __restrict__ grain_velocity, float* __restrict__ grain_acceleration,
                             float* __restrict__ fluid_pressure, float* __restrict__ fluid_velocity) {
  const int nx = L.xo1 - L.xo0;
  const long total = (long)nx * L.ly;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int xr = (int)(k % nx), y = (int)(k / nx);  // output index = y * nx + xr
    const long node = (long)(L.xo0 + xr) * L.sy + y;
    const int i = obst[node];
    float gpr = -1.f, gv0 = 0.f, gv1 = 0.f, ga0 = 0.f, ga1 = 0.f, fp = 0.f, fv0 = 0.f, fv1 = 0.f;
    if (i >= 0 && i < L.n) {
      gpr = (float)gp[i];
      gv0 = (float)v1[i]; gv1 = (float)v2[i];
      ga0 = (float)a1[i]; ga1 = (float)a2[i];
    } else {
#pragma unroll
      for (int q = 0; q < 9; ++q) {
        const real v = f[fidx(q, node)];
        fp = (float)((real)fp + v);
        fv0 = (float)((real)fv0 + v * EXq(q));
        fv1 = (float)((real)fv1 + v * EYq(q));
      }
      fp = (float)((1. / 3.) * rho_moy * ((real)fp - 1.));
    }
    grain_pressure[k] = gpr;
    grain_velocity[3 * k] = gv0; grain_velocity[3 * k + 1] = gv1; grain_velocity[3 * k + 2] = 0.f;
    grain_acceleration[3 * k] = ga0; grain_acceleration[3 * k + 1] = ga1; grain_acceleration[3 * k + 2] = 0.f;
    fluid_pressure[k] = fp;
    fluid_velocity[3 * k] = fv0; fluid_velocity[3 * k + 1] = fv1; fluid_velocity[3 * k + 2] = 0.f;
  }
}

}  // namespace

void launch_aos_to_soa(const real* aos_rows, real* f, const LatticeView& L, hipStream_t st) {
  hipLaunchKernelGGL(k_aos_to_soa, dim3(grid_for((long)L.nxl * L.ly * 9)), dim3(256), 0, st, aos_rows, f, L);
}
void launch_soa_to_aos(const real* f, real* aos_rows, const LatticeView& L, int xl0, int nrows,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_soa_to_aos, dim3(grid_for((long)nrows * L.ly * 9)), dim3(256), 0, st, f, aos_rows, L,
                     xl0, nrows);
}
void launch_fill_equilibrium(real* f, const LatticeView& L, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_equilibrium, dim3(grid_for(9 * L.plane)), dim3(256), 0, st, f, L);
}
void launch_macro(const real* f, const LatticeView& L, int xl0, int nrows, real* rho, real* ux,
                  real* uy, hipStream_t st) {
  hipLaunchKernelGGL(k_macro, dim3(grid_for((long)nrows * L.ly)), dim3(256), 0, st, f, L, xl0, nrows, rho,
                     ux, uy);
}
void launch_density_partial(const real* f, const LatticeView& L, double* partial, int nblocks,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_density_partial, dim3(nblocks), dim3(256), 0, st, f, L, partial);
}
void launch_density_rowsum(const real* f, const LatticeView& L, double* rowsum, hipStream_t st) {
  hipLaunchKernelGGL(k_density_rowsum, dim3(L.xo1 - L.xo0), dim3(256), 0, st, f, L, rowsum);
}
void launch_density_rowquanta(const real* f, const LatticeView& L, const int* kexp, unsigned long long* quanta,
                              int* flags, hipStream_t st) {
  hipLaunchKernelGGL(k_density_rowquanta, dim3(L.xo1 - L.xo0), dim3(256), 0, st, f, L, kexp, quanta, flags);
}
void launch_halo_pack(const real* f, const LatticeView& L, int xl0_lo, int xl0_hi, int nrows, real* buf_lo,
                      real* buf_hi, hipStream_t st) {
  hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(9L * nrows * L.ly), 2), dim3(256), 0, st, f, L, xl0_lo, xl0_hi, nrows,
                     buf_lo, buf_hi);
}
void launch_vtk_fields(const real* f, const int* obst, const LatticeView& L, const real* gp,
                       const real* v1, const real* v2, const real* a1, const real* a2,
                       real rho_moy, float* grain_pressure, float* grain_velocity,
                       float* grain_acceleration, float* fluid_pressure, float* fluid_velocity,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_vtk_fields, dim3(grid_for((long)(L.xo1 - L.xo0) * L.ly)), dim3(256), 0, st, f, obst, L, gp,
                     v1, v2, a1, a2, rho_moy, grain_pressure, grain_velocity, grain_acceleration,
                     fluid_pressure, fluid_velocity);
}

void launch_halo_unpack(real* f, const LatticeView& L, int xl0_lo, int xl0_hi, int nrows, const real* buf_lo,
                        const real* buf_hi, hipStream_t st) {
  hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(9L * nrows * L.ly), 2), dim3(256), 0, st, f, L, xl0_lo, xl0_hi, nrows,
                     buf_lo, buf_hi);
}

// A plain copy, grid-stride, in several shapes (bytes per lane and iteration x workgroups): what this GPU's memory system
// moves when nothing else is asked of it (lbmdem_measure_copy takes the best shape; bench.py reports the fused kernel's
// traffic rate against it, so that lines measured on different boxes can be compared). The shapes are those of
// scripts/micro/stream_pattern.hip: 8 bytes per lane from 2 048 workgroups was the best there (5.6 TB/s), 16 bytes from 8 192 --
// the only shape until round 6 -- some 15 % behind it.
template <class T>
__global__ __launch_bounds__(256) void k_plain_copy(const T* __restrict__ src, T* __restrict__ dst, long n) {
  const long stride = (long)gridDim.x * blockDim.x;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += stride) dst[k] = src[k];
}

int plain_copy_shapes() { return 4; }

void launch_plain_copy(const void* src, void* dst, size_t bytes, hipStream_t st, int shape) {
  switch (shape) {
    case 1: hipLaunchKernelGGL(k_plain_copy<double>, dim3(2048), dim3(256), 0, st, (const double*)src, (double*)dst, (long)(bytes / 8)); break;
    case 2: hipLaunchKernelGGL(k_plain_copy<double2>, dim3(2048), dim3(256), 0, st, (const double2*)src, (double2*)dst, (long)(bytes / 16)); break;
    case 3: hipLaunchKernelGGL(k_plain_copy<double>, dim3(4096), dim3(256), 0, st, (const double*)src, (double*)dst, (long)(bytes / 8)); break;
    default: hipLaunchKernelGGL(k_plain_copy<double2>, dim3(256 * 32), dim3(256), 0, st, (const double2*)src, (double2*)dst, (long)(bytes / 16)); break;
  }
}
