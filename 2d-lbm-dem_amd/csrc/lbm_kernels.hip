// lbm_kernels.hip -- D2Q9 fluid kernels of the MI355X LBM-DEM stepper (gfx950, wave64).
//
// What the reference does in five in-place sweeps of an AoS lattice per fluid step
// (reinit_obst_density main.c:966-986, then collision_streaming main.c:1071-1243: collide,
// edge bounce-back copies, grain interpolated bounce-back, swap, stream) is done here in ONE
// two-lattice pass over SoA planes:
//
//   k_collide_stream: a workgroup stages the post-collision state of a (TX+2)x(TY+2) tile in LDS
//   (solid nodes hold their re-initialised equilibrium instead), then every node of the inner
//   TXxTY tile PULLS its nine populations from the staged neighbours, evaluating the lattice-edge
//   bounce-back and the grain interpolated bounce-back (Bouzidi, moving wall) on the fly. The wall
//   distance delta (main.c:1054-1058) and the `act` flag (main.c:1039-1052) are recomputed from the
//   obstacle map and the grain centres instead of being stored (the reference spends most of
//   obst_construction clearing a 9-real-per-node delta array).
//
// Bit parity with the reference's serial loops is a design constraint: expression association is
// kept, the file is compiled with -ffp-contract=off, and the one order-dependent read of the
// reference's in-place IBB loop (a node two links out that the x-outer/y-inner scan has already
// rewritten) is reproduced by recomputing that node's new value (see pull_one()).
//
// The equivalence swap+stream == pull, f_new[P][q] = f*[P - e_q][q] (or f*[P][opp q] when P - e_q is
// off the array), is derived in SURVEY.md "Notes" and verified by tests against the oracle.

#include "lbmdem_internal.h"

#include <stdlib.h>

namespace {

// D2Q9 direction table (main.c:70-71) and weights (main.c:53-54)
__host__ __device__ constexpr int EXq(int q) {
  return (q == 1 || q == 2 || q == 3) ? -1 : ((q == 5 || q == 6 || q == 7) ? 1 : 0);
}
__host__ __device__ constexpr int EYq(int q) {
  return (q == 1 || q == 7 || q == 8) ? 1 : ((q == 3 || q == 4 || q == 5) ? -1 : 0);
}
__host__ __device__ constexpr int OPPq(int q) { return q == 0 ? 0 : (q <= 4 ? q + 4 : q - 4); }
__host__ __device__ constexpr real Wq(int q) {   // real _w[Q] = {4. / 9, 1. / 36, 1. / 9, ...}: the double quotient, rounded to real
  return (real)(q == 0 ? 4. / 9 : ((q & 1) ? 1. / 36 : 1. / 9));
}

// Populations are stored in tiles of 16 consecutive y: f[x][y / 16][q][y % 16]. The nine 128-byte cache lines
// of a tile are contiguous, so a wave's row (or a grain's footprint) touches one DRAM region per lattice
// instead of nine planes 128 MB apart: the bare marching pattern runs 6.5 % faster than on nine planes
// (scripts/micro/stream_pattern.hip, 0.500 vs 0.534 ms at the kernel's occupancy). `node` = xl * sy + y as for
// the obstacle map; sy is a multiple of 16, so node >> 4 is the tile and node & 15 the position in it.
#ifndef LBMDEM_F_TILES
#define LBMDEM_F_TILES 1   // 0: nine planes f[q][x][y] (A/B builds only: scripts/ab_layout.sh)
#endif
#if LBMDEM_F_TILES
// (LBMDEM_TILE_Y = 16 nodes for double, 32 for float: one 128-byte line per tile and direction either way)
__device__ __forceinline__ long fbase(long node) {
  return (node / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (node % LBMDEM_TILE_Y);
}
// the same from the local row and the column (sy is a multiple of the tile)
__device__ __forceinline__ long fbase_xy(const LatticeView& L, int xl, int y) {
  return (long)xl * L.sy * 9 + (y / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (y % LBMDEM_TILE_Y);
}
#define F_QSTRIDE(L) ((long)LBMDEM_TILE_Y)
#else
__device__ __forceinline__ long fbase(long node) { return node; }
__device__ __forceinline__ long fbase_xy(const LatticeView& L, int xl, int y) { return (long)xl * L.sy + y; }
#define F_QSTRIDE(L) ((L).plane)
#endif
#define fidx(q, node) (fbase(node) + (q) * F_QSTRIDE(L))   // needs the LatticeView `L` in scope

// the fluid-side record of one grain
struct GP { real x1, x2, v1, v2, v3, xc, yc, r2; };

__device__ __forceinline__ GP load_gp(const GrainFluidView& G, int i) {
  const real2* p = reinterpret_cast<const real2*>(G.pk + (long)i * 8);
  const real2 a = p[0], b = p[1], c = p[2], d = p[3];
  return GP{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// Lowest index of the grains covering node `node` whose highest-index cover is `top` (GrainFluidView::mincov).
__device__ __forceinline__ int min_cover(const GrainFluidView& G, long node, int top) {
  if (!G.mincov) return top;
  const unsigned m = G.mincov[node];
  return (m >> 20) == (G.epoch & 0xFFFu) ? (int)(0xFFFFFu - (m & 0xFFFFFu)) : top;
}

// rigid-body velocity of a grain at global node (x, y): main.c:974-975,1172-1173
__device__ __forceinline__ real wall_ux(const LatticeView& L, const GP& g, int y) {
  return g.v1 - (y * L.dx + L.Mby - g.x2) * g.v3;
}
__device__ __forceinline__ real wall_uy(const LatticeView& L, const GP& g, int x) {
  return g.v2 + (x * L.dx + L.Mgx - g.x1) * g.v3;
}

// Correctly rounded a / b from y = RN(1 / b) (Markstein's theorem; Muller et al., Handbook of Floating-
// Point Arithmetic, section 4.7): q = RN(a * y) is a faithful quotient, r = a - b * q is exact in an fma,
// and RN(q + r * y) = RN(a / b) -- for every a, provided the significand of b is not all ones and
// nothing under- or overflows (the callers check b; |a / b| here is a velocity ratio or a moment, far
// from 1e-290). r == 0 means q is already the exact quotient (this also keeps the sign of a zero).
// One division costs ~30 fp64 instructions on gfx950; this costs 3 + a select. Round 1 measured -18 % VALU instructions
// and -1 % time with it in the fused kernel; by round 3 (DPP shifts, branch-free classification) the guards, the selects
// and the skipped fallback cost more than the divisions: true divisions are 0-4.7 % faster (six interleaved pairs on one
// GPU, never slower), so the product uses them and this form only exists with -DLBMDEM_EXACT_RECIP.
#ifdef LBMDEM_EXACT_RECIP
__device__ __forceinline__ real exact_div(real a, real b, real y) {
  const real q = a * y;
  const real r = __builtin_fma(-b, q, a);     // (the float overload for float operands: exact there too)
  return r == (real)0.0 ? q : (real)__builtin_fma(r, y, q);
}
#ifdef LBMDEM_SINGLE_PRECISION
__device__ __forceinline__ bool significand_all_ones(real v) { return (__float_as_int(v) & 0x7FFFFF) == 0x7FFFFF; }
// numerators for which exact_div cannot underflow (divisors here are O(1) .. O(1e8))
__device__ __forceinline__ bool div_safe(real a) { return a == 0.0f || (fabsf(a) > 1e-20f && fabsf(a) < 1e20f); }
#else
__device__ __forceinline__ bool significand_all_ones(real v) {
  return (__double_as_longlong(v) & 0xFFFFFFFFFFFFFll) == 0xFFFFFFFFFFFFFll;
}
// numerators for which exact_div cannot underflow (divisors here are O(1) .. O(1e8))
__device__ __forceinline__ bool div_safe(real a) { return a == 0.0 || (fabs(a) > 1e-200 && fabs(a) < 1e200); }
#endif
#endif  // LBMDEM_EXACT_RECIP

// main.c:1082-1116, in registers
__device__ __forceinline__ void mrt_collide(const LatticeView& L, real (&f)[9]) {
  const real a = 1. / 36;
  const real f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4], f5 = f[5], f6 = f[6],
               f7 = f[7], f8 = f[8];
  const real rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
  const real e = -4 * f0 + 2 * f1 - f2 + 2 * f3 - f4 + 2 * f5 - f6 + 2 * f7 - f8;
  const real eps = 4 * f0 + f1 - 2 * f2 + f3 - 2 * f4 + f5 - 2 * f6 + f7 - 2 * f8;
  const real j_x = f5 + f6 + f7 - f1 - f2 - f3;
  const real q_x = -f1 + 2 * f2 - f3 + f5 - 2 * f6 + f7;
  const real j_y = f1 + f8 + f7 - f3 - f4 - f5;
  const real q_y = f1 - f3 + 2 * f4 - f5 + f7 - 2 * f8;
  const real p_xx = f2 - f4 + f6 - f8;
  const real p_xy = -f1 + f3 - f5 + f7;

  const real j_x2 = j_x * j_x;
  const real j_y2 = j_y * j_y;

  // three quotients by the same rho: one true division (the reciprocal) + three exact_div
  real d1, d2, d3;  // 3 * (j_x2 + j_y2) / rho, (j_x2 - j_y2) / rho, j_x * j_y / rho
  const real n1 = 3 * (j_x2 + j_y2), n2 = j_x2 - j_y2, n3 = j_x * j_y;
#ifdef LBMDEM_EXACT_RECIP   /* experiment builds: the round-1 form (one reciprocal + three exact_div, guarded) */
  if (L.recip_ok && !significand_all_ones(rho) && rho > 1e-8 && rho < 1e8 && div_safe(n1) && div_safe(n2) &&
      div_safe(n3)) {
    const real y = 1.0 / rho;
    d1 = exact_div(n1, rho, y);
    d2 = exact_div(n2, rho, y);
    d3 = exact_div(n3, rho, y);
  } else
#endif
  {
    d1 = n1 / rho;
    d2 = n2 / rho;
    d3 = n3 / rho;
  }
  const real eO = e - L.s2 * (e + 2 * rho - d1);
  const real epsO = eps - L.s3 * (eps - rho + d1);
  const real q_xO = q_x - L.s5 * (q_x + j_x);
  const real q_yO = q_y - L.s7 * (q_y + j_y);
  const real p_xxO = p_xx - L.s8 * (p_xx - d2);
  const real p_xyO = p_xy - L.s9 * (p_xy - d3);

  f[0] = a * (4 * rho - 4 * eO + 4 * epsO);
  f[2] = a * (4 * rho - eO - 2 * epsO - 6 * j_x + 6 * q_xO + 9 * p_xxO);
  f[4] = a * (4 * rho - eO - 2 * epsO - 6 * j_y + 6 * q_yO - 9 * p_xxO);
  f[6] = a * (4 * rho - eO - 2 * epsO + 6 * j_x - 6 * q_xO + 9 * p_xxO);
  f[8] = a * (4 * rho - eO - 2 * epsO + 6 * j_y - 6 * q_yO - 9 * p_xxO);
  f[1] = a * (4 * rho + 2 * eO + epsO - 6 * j_x - 3 * q_xO + 6 * j_y + 3 * q_yO - 9 * p_xyO);
  f[3] = a * (4 * rho + 2 * eO + epsO - 6 * j_x - 3 * q_xO - 6 * j_y - 3 * q_yO + 9 * p_xyO);
  f[5] = a * (4 * rho + 2 * eO + epsO + 6 * j_x + 3 * q_xO - 6 * j_y - 3 * q_yO - 9 * p_xyO);
  f[7] = a * (4 * rho + 2 * eO + epsO + 6 * j_x + 3 * q_xO + 6 * j_y + 3 * q_yO + 9 * p_xyO);
}

// equilibrium at rho = 1 and the grain's rigid-body velocity: main.c:974-981.
// The reference evaluates eu = (ex*ux + ey*uy)/c for all nine directions. Opposite directions have
// exactly negated numerators (negation and IEEE rounding commute), so eu[q+4] == -eu[q] bit for bit;
// products with ex, ey in {0, +-1} are exact (a 0*u term only decides the sign of a zero sum, which
// 1. + 3*eu and eu*eu then erase). Hence four divisions instead of nine, same bits.
__device__ __forceinline__ void grain_equilibrium_u(const LatticeView& L, real ux, real uy, real (&f)[9]) {
  real u_squ, e1, e2, e3, e4;
#ifdef LBMDEM_EXACT_RECIP
  if (L.recip_ok && div_safe(ux * ux) && div_safe(uy * uy)) {  // divisions by the run constants c and c*c
    u_squ = exact_div(ux * ux + uy * uy, L.cc, L.rcc);
    e1 = exact_div(-ux + uy, L.c, L.rc);
    e2 = exact_div(-ux, L.c, L.rc);
    e3 = exact_div(-ux + (-uy), L.c, L.rc);
    e4 = exact_div(-uy, L.c, L.rc);
  } else
#endif
  {
    u_squ = (ux * ux + uy * uy) / L.cc;   // L.cc = c * c in `real` arithmetic (host), main.c:976
    e1 = (-ux + uy) / L.c;     // q = 1: (-1, 1)
    e2 = (-ux) / L.c;          // q = 2: (-1, 0)
    e3 = (-ux + (-uy)) / L.c;  // q = 3: (-1,-1)
    e4 = (-uy) / L.c;          // q = 4: ( 0,-1)
  }
  // main.c:980: w * (1. + 3 * eu + 4.5 * eu * eu - 1.5 * u_squ) -- the literals 1., 4.5, 1.5 make the bracket a double
  // sum in either build (3 * eu is an int times a real: a real); the product with w is rounded to real once
  const double k = 1.5 * u_squ;
  f[0] = Wq(0) * (1. + 0.0 - k);          // eu = 0: 1. + 3*0 + 4.5*0*0 == 1.
  f[1] = Wq(1) * (1. + 3 * e1 + 4.5 * e1 * e1 - k);
  f[5] = Wq(5) * (1. + 3 * (-e1) + 4.5 * e1 * e1 - k);
  f[2] = Wq(2) * (1. + 3 * e2 + 4.5 * e2 * e2 - k);
  f[6] = Wq(6) * (1. + 3 * (-e2) + 4.5 * e2 * e2 - k);
  f[3] = Wq(3) * (1. + 3 * e3 + 4.5 * e3 * e3 - k);
  f[7] = Wq(7) * (1. + 3 * (-e3) + 4.5 * e3 * e3 - k);
  f[4] = Wq(4) * (1. + 3 * e4 + 4.5 * e4 * e4 - k);
  f[8] = Wq(8) * (1. + 3 * (-e4) + 4.5 * e4 * e4 - k);
}
__device__ __forceinline__ void grain_equilibrium(const LatticeView& L, const GP& g, int x, int y,
                                                  real (&f)[9]) {
  grain_equilibrium_u(L, wall_ux(L, g, y), wall_uy(L, g, x), f);
}

// wall distance along link q from solid node (x, y) of a disc (xc, yc, r2): main.c:1054-1058
template <int q>
__device__ __forceinline__ real link_delta(int x, int y, real xc, real yc, real r2) {
  constexpr int ex = EXq(q), ey = EYq(q);
  const real aa = (real)(ex < 0 ? -ex : ex) + (real)(ey < 0 ? -ey : ey);
  const real bb = (x + ex - xc) * ex + (y + ey - yc) * ey;
  const real cc = (x + ex - xc) * (x + ex - xc) + (y + ey - yc) * (y + ey - yc) - r2;
  // main.c:1058: fabs() and sqrt() are <math.h>'s double functions: everything right of `bb -` is double in either build
  return (real)((bb - sqrt(fabs((double)(bb * bb - aa * cc)))) / aa);
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// obst = -1 in the interior, nbgrains on the four lattice edges (main.c:669-683, 997-999): obst_fill_range (lbmdem_internal.h)
__global__ void k_obst_fill(int* __restrict__ obst, LatticeView L, int row0, int row1) {
  obst_fill_range(obst, L, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, row0, row1);
}

// Rasterise the reduced discs (main.c:1016-1032). One wavefront per grain: every lane derives the grain's lattice
// geometry (main.c:1009-1013: centre, squared reduced radius, unreduced radius in nodes), lane 0 stores it -- also as
// one packed 64-byte record {x1, x2, v1, v2, v3, xc, yc, r2} for the fluid kernels -- and the lanes sweep the bounding
// box two rows at a time, y fastest (coalesced, no integer divisions). Overlaps resolve to the highest grain index,
// which is what the reference's ascending serial paint produces (main.c:1028) -> atomicMax; the value it returns
// tells a painter that the node lies under several discs: both grains are flagged as overlapping (the force kernel
// derives the footprint of unflagged grains from the disc test alone) and the lowest index covering the node is
// recorded in `mincov` (every painter records itself and the owner it found, so the lowest cover ends up there
// whatever the order of the painters).
// (Measured alternative: plain stores, then a second launch that re-reads the nodes and settles overlaps with atomics
// only where they occur -- 37 + 30 us against 52 us: the kernel is bound by its 50 000 short waves, not by atomics.)
__global__ void k_obst_paint(int* __restrict__ obst, LatticeView L, int n, const real* __restrict__ x1,
                             const real* __restrict__ x2, const real* __restrict__ r,
                             const real* __restrict__ rLB, const real* __restrict__ v1,
                             const real* __restrict__ v2, const real* __restrict__ v3,
                             real* __restrict__ oxc, real* __restrict__ oyc, real* __restrict__ or2,
                             real* __restrict__ orbl0, real* __restrict__ pk,
                             unsigned char* __restrict__ touched, const unsigned char* __restrict__ mask,
                             unsigned* __restrict__ mincov, unsigned epoch, const int* __restrict__ list,
                             const int* __restrict__ list_count, int list_cap, const int* __restrict__ voff,
                             const int* __restrict__ vnbr) {
  const int lane = threadIdx.x & 63;
  int i = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (list) {                    // strip decomposition: only the grains that can reach this rank's rows
    if (i >= *list_count || i >= list_cap) return;   // (an overflowing list is flagged by its producer)
    i = list[i];
  }
  if (i >= n) return;
  if (mask && !mask[i]) return;
  const real gx1 = x1[i], gx2 = x2[i];
  const real xc = (gx1 - L.Mgx) / L.dx, yc = (gx2 - L.Mby) / L.dx, r2 = rLB[i] * rLB[i], rbl0 = r[i] / L.dx;
  if (lane == 0) {
    oxc[i] = xc; oyc[i] = yc; or2[i] = r2; orbl0[i] = rbl0;
    real* o = pk + (long)i * 8;
    o[0] = gx1; o[1] = gx2; o[2] = v1[i]; o[3] = v2[i]; o[4] = v3[i]; o[5] = xc; o[6] = yc; o[7] = r2;
  }
  const real R2 = rbl0 * rbl0;
  int xi = (int)(xc - rbl0), xf = (int)(xc + rbl0);
  if (xi < 1) xi = 1;
  if (xf >= L.lx - 1) xf = L.lx - 2;
  int yi = (int)(yc - rbl0), yf = (int)(yc + rbl0);
  if (yi < 1) yi = 1;
  if (yf >= L.ly - 1) yf = L.ly - 2;
  // restrict to the local slab
  if (xi < L.gx0) xi = L.gx0;
  if (xf > L.gx0 + L.nxl - 1) xf = L.gx0 + L.nxl - 1;
  if (xi > xf || yi > yf) return;
  const int ny = yf - yi + 1;
  auto in_disc = [&](int x, int y) {
    const real d2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
    return d2 <= R2 && d2 <= r2;
  };
  // Round 3: reduced discs of a physical packing do not overlap, so nearly all of the ~10 M returning atomicMax per
  // step had nobody to arbitrate with (they were what set this kernel's time: profiles/r02_l_sq_counters.txt). A disc that
  // is at least 1.5 nodes clear of every partner in the grain's Verlet list (symmetric: every pair within distVerlet of
  // touching at the last rebuild, which is far more than reduced discs need to meet) shares no node with another disc:
  // plain stores, no flags. Everything else -- and every grain while there is no list (voff == null: before the first
  // rebuild, strips with distributed grains) -- keeps the atomic path.
  bool alone = false;
  if (voff) {
    const int k0 = voff[i], k1 = voff[i + 1];
    bool near = false;
    const real ri = rLB[i];
    for (int k = k0 + lane; k < k1; k += 64) {
      const int j = vnbr[k];
      const real ddx = (x1[j] - gx1) / L.dx, ddy = (x2[j] - gx2) / L.dx, rr = ri + rLB[j] + 1.5;
      near |= !(ddx * ddx + ddy * ddy >= rr * rr);   // also true for a NaN
    }
    alone = !__any(near);
  }
  if (alone) {
    if (ny <= 32) {
      const int y = yi + (lane & 31);
      if ((lane & 31) < ny)
        for (int x = xi + (lane >> 5); x <= xf; x += 2)
          if (in_disc(x, y)) obst[(long)(x - L.gx0) * L.sy + y] = i;
    } else {
      const int total = (xf - xi + 1) * ny;
      for (int k = lane; k < total; k += 64) {
        const int x = xi + k / ny, y = yi + k % ny;
        if (in_disc(x, y)) obst[(long)(x - L.gx0) * L.sy + y] = i;
      }
    }
    return;
  }
  auto overlap = [&](long node, int old) {   // the node was somebody else's: rare
    if (old >= 0 && old < n && old != i) {
      touched[i] = 1; touched[old] = 1;
      if (mincov) {
        atomicMax(&mincov[node], (epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)i));
        atomicMax(&mincov[node], (epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)old));
      }
    }
  };
  constexpr int SWEEPS = 12;   // two rows of the box per sweep, y fastest: no integer divisions
  if (ny <= 32 && xf - xi + 1 <= 2 * SWEEPS) {
    // all atomics of the wave are issued before the first returned value is looked at: one round trip, not twelve
    const int y = yi + (lane & 31);
    const bool col = (lane & 31) < ny;
    int old[SWEEPS];
#pragma unroll
    for (int s_ = 0; s_ < SWEEPS; ++s_) {
      const int x = xi + 2 * s_ + (lane >> 5);
      old[s_] = -1;
      if (col && x <= xf && in_disc(x, y)) old[s_] = atomicMax(&obst[(long)(x - L.gx0) * L.sy + y], i);
    }
#pragma unroll
    for (int s_ = 0; s_ < SWEEPS; ++s_) {
      const int x = xi + 2 * s_ + (lane >> 5);
      if (old[s_] >= 0) overlap((long)(x - L.gx0) * L.sy + y, old[s_]);
    }
  } else {
    const int total = (xf - xi + 1) * ny;
    for (int k = lane; k < total; k += 64) {
      const int x = xi + k / ny, y = yi + k % ny;
      if (in_disc(x, y)) {
        const long node = (long)(x - L.gx0) * L.sy + y;
        overlap(node, atomicMax(&obst[node], i));
      }
    }
  }
}

__global__ void k_fill_u64(unsigned long long* __restrict__ p, long count, unsigned long long v) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) p[k] = v;
}

// ---------------------------------------------------------------------------------------------
// the fused fluid kernel
// ---------------------------------------------------------------------------------------------

template <int TX, int TY>
struct Tile {
  static constexpr int RX = TX + 2, RY = TY + 2;  // staged populations: halo 1
  static constexpr int OX = TX + 4, OY = TY + 4;  // staged obstacle ids: halo 2 (act of halo-1 nodes)
  real* sF;  // [9][RX][RY]
  int* sO;     // [OX][OY]
  __device__ __forceinline__ real& F(int q, int tx, int ty) const {
    return sF[(q * RX + (tx + 1)) * RY + (ty + 1)];
  }
  __device__ __forceinline__ int O(int tx, int ty) const { return sO[(tx + 2) * OY + (ty + 2)]; }
  // A solid node is "active" when one of its 8 neighbours was fluid at the moment its owning grain
  // was painted (main.c:1039-1052). Grains are painted in ascending index, so besides the
  // neighbours that are fluid in the final map this also counts neighbours now covered by a
  // HIGHER-index grain that do not lie inside the owner's own disc (they were still fluid when the
  // owner was painted). Only reachable when reduced discs of different grains touch or overlap.
  // A neighbour additionally covered by a third, LOWER-index disc was not fluid then: decided with the
  // rasteriser's record of the lowest index covering a multiply covered node (min_cover).
  __device__ __forceinline__ bool active(const LatticeView& L, const GrainFluidView& G, int tx, int ty,
                                         int gx, int gy) const {
    const int oS = O(tx, ty);
    bool higher = false;
#pragma unroll
    for (int q = 1; q < 9; ++q) {
      const int o = O(tx + EXq(q), ty + EYq(q));
      if (o == -1) return true;
      higher |= (o > oS && o != L.n);
    }
    if (!higher) return false;
    const real xc = G.xc[oS], yc = G.yc[oS], r2 = G.r2[oS], rb = G.rbl0[oS];
    const real R2 = rb * rb;
#pragma unroll
    for (int q = 1; q < 9; ++q) {
      const int o = O(tx + EXq(q), ty + EYq(q));
      if (o > oS && o != L.n) {
        const int x = gx + EXq(q), y = gy + EYq(q);
        const real d2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
        if (!(d2 <= R2 && d2 <= r2) && min_cover(G, (long)(x - L.gx0) * L.sy + y, o) > oS) return true;
      }
    }
    return false;
  }
};

// Interpolated bounce-back (Bouzidi, moving wall) at solid node S = (sx, sy) of grain i for link q
// towards the fluid node N = S + e_q: main.c:1166-1185 / 1198-1217.
struct IbbLink { real d, uw; };

template <int q>
__device__ __forceinline__ IbbLink ibb_link(const LatticeView& L, const GP& g, int sx, int sy) {
  constexpr int ex = EXq(q), ey = EYq(q);
  IbbLink k;
  k.d = link_delta<q>(sx, sy, g.xc, g.yc, g.r2);
  k.uw = ex * wall_ux(L, g, sy) + ey * wall_uy(L, g, sx);
  return k;
}
__device__ __forceinline__ bool ibb_far(const IbbLink& k) { return k.d >= 0.5; }
__device__ __forceinline__ bool ibb_near(const IbbLink& k) { return k.d > 0. && k.d < 0.5; }
// delta >= 1/2: fN_opp = f*[N][opp q], fN_q = f*[N][q]
template <int q>
__device__ __forceinline__ real ibb_far_value(const LatticeView& L, const IbbLink& k, real fN_opp,
                                                real fN_q) {
  return fN_opp / (2 * k.d) + (2 * k.d - 1) * fN_q / (2 * k.d) + 3 * (Wq(q) / L.c) * k.uw / k.d;
}
// 0 < delta < 1/2: f2 = the population read two links out, f[N + e_q][opp q]
template <int q>
__device__ __forceinline__ real ibb_near_value(const LatticeView& L, const IbbLink& k, real fN_opp,
                                                 real f2) {
  return 2 * k.d * fN_opp + (1 - 2 * k.d) * f2 + 6 * (Wq(q) / L.c) * k.uw;
}

// f_new[P][q] for one direction, P = (gx, gy) global. The node's surroundings come from a context:
//   C.own(q)     f*[P][q]              (f* = post-collision for fluid nodes, pre-IBB value otherwise)
//   C.in(d)      f*[P + e_d][opp d]    the population of the neighbour in direction d that points at P
//   C.o_own(), C.o_nb(d)               obstacle ids of P and of that neighbour
//   C.act_nb(d)                        `act` flag of that neighbour (asked only for interior solid ones)
//   C.gp_nb(d)                         fluid-side record of the grain that owns that neighbour
// so f*[S][q] with S = P - e_q is C.in(opp q) and f*[P + e_q][opp q] is C.in(q).
// EDGE = false is the specialisation for nodes at least two rows/columns away from every lattice edge
// (S, P and P + e_q are then all interior): the edge logic disappears.
// Everything of a pull except the interpolated bounce-back itself. Returns true when (P, q) IS an
// interpolated-bounce-back link (P fluid, source S an interior grain node) -- `out` is then not set.
template <int q, bool EDGE = true, class Ctx>
__device__ __forceinline__ bool pull_classify(const Ctx& C, const LatticeView& L, int gx, int gy, real& out) {
  constexpr int ex = EXq(q), ey = EYq(q), qo = OPPq(q);
  const int sxg = gx - ex, syg = gy - ey;  // source node S = P - e_q
  if (EDGE && (sxg < 0 || sxg >= L.lx || syg < 0 || syg >= L.ly)) {  // array edge: main.c:1237
    out = C.own(qo);
    return false;
  }
  const bool s_interior = !EDGE || (sxg >= 1 && sxg <= L.lx - 2 && syg >= 1 && syg <= L.ly - 2);
  if (!s_interior) {
    // S is a lattice-edge wall node. Its slot q was overwritten by the edge copies
    // (main.c:1123-1145) with f*[P][opp q] when P is interior, and -- because the y-edge loop runs
    // before the x-edge loop -- also when S sits on a y edge and P on an x edge; otherwise it
    // still holds its old value.
    const bool s_yedge = (syg == 0 || syg == L.ly - 1) && sxg >= 1 && sxg <= L.lx - 2;
    const bool copied = gy >= 1 && gy <= L.ly - 2 && ((gx >= 1 && gx <= L.lx - 2) || s_yedge);
    out = copied ? C.own(qo) : C.in(qo);
    // EXTENSION (off unless lbmdem_set_lid): the top plate's lid terms the reference has commented out,
    // f[x][ly-1][3] = ... - uw_h/6, f[x][ly-1][5] = ... + uw_h/6 (main.c:1129-1130)
    if ((q == 3 || q == 5) && L.lid6 != 0.0 && copied && syg == L.ly - 1 && sxg >= 1 && sxg <= L.lx - 2)
      out = q == 3 ? out - L.lid6 : out + L.lid6;
    // ... which the side-wall copies that run afterwards (main.c:1134,1138) hand on to the two wall nodes next to the
    // top corners: f[0][ly-2][7] = f[1][ly-1][3], f[lx-1][ly-2][1] = f[lx-2][ly-1][5]
    if (q == 7 && L.lid6 != 0.0 && sxg == 0 && syg == L.ly - 2) out = out - L.lid6;
    if (q == 1 && L.lid6 != 0.0 && sxg == L.lx - 1 && syg == L.ly - 2) out = out + L.lid6;
    return false;
  }
  const int oS = C.o_nb(qo);
  if (oS == -1) {  // plain streaming from a fluid node
    out = C.in(qo);
    return false;
  }
  if (C.o_own() != -1) {  // solid -> non-fluid link: active solid nodes reset the slot to w (main.c:1161-1162)
    out = C.act_nb(qo) ? Wq(q) : C.in(qo);
    return false;
  }
  return true;
}

// Is the node two links out, NN = P + e_q, an interior node? (only asked for bounce-back links)
template <int q, bool EDGE>
__device__ __forceinline__ bool nn_interior(const LatticeView& L, int gx, int gy) {
  const int nxg = gx + EXq(q), nyg = gy + EYq(q);
  return !EDGE || (nxg >= 1 && nxg <= L.lx - 2 && nyg >= 1 && nyg <= L.ly - 2);
}

// The interpolated bounce-back value of link (P, q): P fluid, S = P - e_q an (active) node of a grain.
template <int q, bool EDGE = true, class Ctx>
__device__ __forceinline__ real ibb_eval(const Ctx& C, const LatticeView& L, int gx, int gy) {
  constexpr int ex = EXq(q), ey = EYq(q), qo = OPPq(q);
  const int sxg = gx - ex, syg = gy - ey;
  const IbbLink k = ibb_link<q>(L, C.gp_nb(qo), sxg, syg);
  if (ibb_far(k)) return ibb_far_value<q>(L, k, C.own(qo), C.own(q));
  if (!ibb_near(k)) return C.in(qo);  // neither branch fires: slot keeps its value

  // 0 < delta < 1/2: the reference reads f[NN][opp q], NN = P + e_q, *in place* (main.c:1181,1213).
  real f2;
  const int nxg = gx + ex, nyg = gy + ey;
  if (!nn_interior<q, EDGE>(L, gx, gy)) {
    f2 = C.own(q);  // edge wall node: its slot opp q was set by the edge copy to f*[P][q]
  } else {
    const int oN = C.o_nb(q);
    f2 = C.in(q);  // fluid: post-collision; solid: value before the IBB loop
    // NN is solid and precedes S in the reference's x-outer/y-inner scan (e_q lexicographically
    // negative, q = 1..4): S reads the value NN's own IBB update has already produced. That update
    // saw S's slot q in its pre-loop state (S comes later), so the chain ends here.
    if (oN != -1 && q <= 4) {
      const IbbLink kn = ibb_link<qo>(L, C.gp_nb(q), nxg, nyg);
      if (ibb_far(kn)) f2 = ibb_far_value<qo>(L, kn, C.own(q), C.own(qo));
      else if (ibb_near(kn)) f2 = ibb_near_value<qo>(L, kn, C.own(q), C.in(qo));
    }
  }
  return ibb_near_value<q>(L, k, C.own(qo), f2);
}

template <int q, bool EDGE = true, class Ctx>
__device__ __forceinline__ real pull_one(const Ctx& C, const LatticeView& L, const GrainFluidView& G,
                                           int gx, int gy) {
  real out;
  if (!pull_classify<q, EDGE>(C, L, gx, gy, out)) return out;
  return ibb_eval<q, EDGE>(C, L, gx, gy);
}

// The same bounce-back value with the direction as a RUN-TIME argument, for the compacted evaluation of
// the marching kernel (one lane per link, any direction). Formula for formula the arithmetic of
// ibb_eval<q>: (x + ex - xc) * ex etc. are the same IEEE operations whether ex is a template constant
// or a variable; the division by aa in {1, 2} is exact either way.
struct RtLink {
  int q;                 // 1..8
  int gx, gy;            // P
  real own_qo, own_q;  // f*[P][opp q], f*[P][q]
  real in_q, in_qo;    // f*[P + e_q][opp q], f*[P - e_q][q]
  bool nn_int, hazard;   // NN interior; NN solid and q <= 4 (its own update precedes S's)
};
__device__ __forceinline__ real link_delta_rt(int x, int y, int ex, int ey, real xc, real yc, real r2) {
  const int aai = (ex < 0 ? -ex : ex) + (ey < 0 ? -ey : ey);
  const real aa = (real)aai;
  const real bb = (x + ex - xc) * ex + (y + ey - yc) * ey;
  const real cc = (x + ex - xc) * (x + ex - xc) + (y + ey - yc) * (y + ey - yc) - r2;
  const double t = bb - sqrt(fabs((double)(bb * bb - aa * cc)));   // double in either build, see link_delta
  return (real)(aai == 2 ? t * 0.5 : t);  // == t / aa exactly
}
// wc_diag = Wq(1) / L.c, wc_axis = Wq(2) / L.c (what ibb_*_value<q> form as Wq(q) / L.c)
// Lattice line through node (x, y) parallel to e = (ex, ey), numbered relative to the grain centre (xc, yc):
// the index of a link's slot in the ForceSlots table. Producer (fused kernel) and consumer (force kernel) both
// call this with the same doubles.
__device__ __forceinline__ int slot_line(int x, int y, int ex, int ey, real xc, real yc) {
  return ey * (x - (int)xc) - ex * (y - (int)yc);
}

// delta >= 1/2 (main.c:1175-1177 / 1207-1209). (Round 3 measured the three quotients as one true reciprocal + three
// exact_div: the same bits, ~50 instructions fewer, 2-4 % SLOWER -- the guards and selects lengthen the pass's dependent
// chain; not kept.)
__device__ __forceinline__ real ibb_far_rt(const LatticeView& L, const IbbLink& k, real fN_opp, real fN_q, real wc) {
  return fN_opp / (2 * k.d) + (2 * k.d - 1) * fN_q / (2 * k.d) + 3 * wc * k.uw / k.d;
}

template <class RecFn>
__device__ __forceinline__ real ibb_eval_rt(const LatticeView& L, const RtLink& k, real wc_diag,
                                              real wc_axis, RecFn rec_of) {
  const int q = k.q;
  const int ex = (q >= 1 && q <= 3) ? -1 : ((q >= 5 && q <= 7) ? 1 : 0);
  const int ey = (q == 1 || q >= 7) ? 1 : ((q >= 3 && q <= 5) ? -1 : 0);
  const real wc = (q & 1) ? wc_diag : wc_axis;
  const int sx = k.gx - ex, sy = k.gy - ey;
  const GP g = rec_of(-ex, -ey);  // record of the grain that owns S = P - e_q
  IbbLink a;
  a.d = link_delta_rt(sx, sy, ex, ey, g.xc, g.yc, g.r2);
  a.uw = ex * wall_ux(L, g, sy) + ey * wall_uy(L, g, sx);
  if (a.d >= 0.5) return ibb_far_rt(L, a, k.own_qo, k.own_q, wc);
  if (!(a.d > 0. && a.d < 0.5)) return k.in_qo;
  real f2;
  if (!k.nn_int) {
    f2 = k.own_q;
  } else {
    f2 = k.in_q;
    if (k.hazard) {
      const int nx = k.gx + ex, ny = k.gy + ey;
      const GP gn = rec_of(ex, ey);  // record of the grain that owns NN = P + e_q
      IbbLink b;
      b.d = link_delta_rt(nx, ny, -ex, -ey, gn.xc, gn.yc, gn.r2);
      b.uw = (-ex) * wall_ux(L, gn, ny) + (-ey) * wall_uy(L, gn, nx);
      if (b.d >= 0.5) f2 = ibb_far_rt(L, b, k.own_q, k.own_qo, wc);
      else if (b.d > 0. && b.d < 0.5) f2 = 2 * b.d * k.own_q + (1 - 2 * b.d) * k.in_qo + 6 * wc * b.uw;
    }
  }
  return 2 * a.d * k.own_qo + (1 - 2 * a.d) * f2 + 6 * wc * a.uw;
}

// context over the LDS tile
template <int TX, int TY>
struct TileCtx {
  const Tile<TX, TY>& T;
  const LatticeView& L;
  const GrainFluidView& G;
  int px, py, gx, gy;
  __device__ __forceinline__ real own(int q) const { return T.F(q, px, py); }
  __device__ __forceinline__ real in(int d) const { return T.F(OPPq(d), px + EXq(d), py + EYq(d)); }
  __device__ __forceinline__ int o_own() const { return T.O(px, py); }
  __device__ __forceinline__ int o_nb(int d) const { return T.O(px + EXq(d), py + EYq(d)); }
  __device__ __forceinline__ bool act_nb(int d) const {
    return T.active(L, G, px + EXq(d), py + EYq(d), gx + EXq(d), gy + EYq(d));
  }
  __device__ __forceinline__ GP gp_nb(int d) const { return load_gp(G, o_nb(d)); }
};

template <int TX, int TY>
__global__ __launch_bounds__(256) void k_collide_stream(const real* __restrict__ fin,
                                                        real* __restrict__ fout,
                                                        const int* __restrict__ ob_old,
                                                        const int* __restrict__ ob_new, LatticeView L,
                                                        GrainFluidView G, int tiles_y, int ntiles,
                                                        int xcd_remap) {
  using TT = Tile<TX, TY>;
  __shared__ real sF[9 * TT::RX * TT::RY];
  __shared__ int sO[TT::OX * TT::OY];
  TT T{sF, sO};
  const int tid = threadIdx.x;
  // Tile index. Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8), each
  // with a private L2. With the remap, XCD k walks the k-th contiguous eighth of the tile sequence
  // (y fastest), so tiles that share halo rows/columns run on the same XCD close in time and the
  // halo re-reads hit that L2. Placement only affects speed.
  int t = blockIdx.x;
  if (xcd_remap) {
    const int per = gridDim.x >> 3;
    t = (t & 7) * per + (t >> 3);
  }
  if (t >= ntiles) return;
  const int ty0 = (t % tiles_y) * TY;           // global y of the tile origin
  const int txl0 = L.xo0 + (t / tiles_y) * TX;  // local row of the tile origin

  // obstacle ids, halo 2; positions off the lattice read as "wall" so they never look fluid
  for (int k = tid; k < TT::OX * TT::OY; k += 256) {
    const int ox = k / TT::OY, oy = k % TT::OY;
    const int xl = txl0 - 2 + ox, y = ty0 - 2 + oy;
    int v = L.n;
    if (xl >= 0 && xl < L.nxl && y >= 0 && y < L.ly) v = ob_new[(long)xl * L.sy + y];
    sO[k] = v;
  }
  __syncthreads();

  // phase 1: stage f* (before IBB) for the tile + halo 1
  for (int k = tid; k < TT::RX * TT::RY; k += 256) {
    const int rx = k / TT::RY, ry = k % TT::RY;
    const int xl = txl0 - 1 + rx, y = ty0 - 1 + ry;
    if (xl < 0 || xl >= L.nxl || y < 0 || y >= L.ly) continue;  // never read in phase 2
    const int gx = L.gx0 + xl;
    const long node = (long)xl * L.sy + y;
    const bool interior = gx >= 1 && gx <= L.lx - 2 && y >= 1 && y <= L.ly - 2;
    real f[9];
    // reinit_obst_density (main.c:966-986) acts on the PREVIOUS obstacle map with the current grain
    // state: nodes that were solid restart from the grain's equilibrium
    const int oo = interior ? ob_old[node] : -1;
    if (oo != -1) {
      grain_equilibrium(L, load_gp(G, oo), gx, y, f);
    } else {
#pragma unroll
      for (int q = 0; q < 9; ++q) f[q] = fin[fidx(q, node)];
    }
    if (interior && sO[(rx + 1) * TT::OY + (ry + 1)] == -1) mrt_collide(L, f);
#pragma unroll
    for (int q = 0; q < 9; ++q) sF[(q * TT::RX + rx) * TT::RY + ry] = f[q];
  }
  __syncthreads();

  // phase 2: pull
  for (int k = tid; k < TX * TY; k += 256) {
    const int px = k / TY, py = k % TY;
    const int xl = txl0 + px, gy = ty0 + py;
    if (xl >= L.xo1 || gy >= L.ly) continue;
    const int gx = L.gx0 + xl;
    const long node = (long)xl * L.sy + gy;
    const TileCtx<TX, TY> C{T, L, G, px, py, gx, gy};
    fout[fidx(0, node)] = T.F(0, px, py);
    fout[fidx(1, node)] = pull_one<1>(C, L, G, gx, gy);
    fout[fidx(2, node)] = pull_one<2>(C, L, G, gx, gy);
    fout[fidx(3, node)] = pull_one<3>(C, L, G, gx, gy);
    fout[fidx(4, node)] = pull_one<4>(C, L, G, gx, gy);
    fout[fidx(5, node)] = pull_one<5>(C, L, G, gx, gy);
    fout[fidx(6, node)] = pull_one<6>(C, L, G, gx, gy);
    fout[fidx(7, node)] = pull_one<7>(C, L, G, gx, gy);
    fout[fidx(8, node)] = pull_one<8>(C, L, G, gx, gy);
  }
}

// ---------------------------------------------------------------------------------------------
// the fused fluid kernel, register-resident "marching" form
// ---------------------------------------------------------------------------------------------
//
// One WAVEFRONT owns a window of 64 consecutive y (62 produce output, the two edge lanes only feed
// their neighbours) and walks along x over LX rows. Each lane keeps the f* of its column for three
// consecutive rows in registers; row x+1 is re-initialised/collided and rotated in while row x is
// produced. The six diagonal/vertical neighbours a pull needs live in the adjacent lanes and are
// fetched with cross-lane shuffles; the obstacle ids of the 3x3 neighbourhood are read directly (they
// are 4-byte, cached). No barriers: waves are independent, every load of a row is a full 512-byte
// coalesced request, and the loads of the next two rows are in flight while the current one is
// computed. LDS is used only wave-privately: a ring of grain records (RecRing) and the scratch of the
// compacted bounce-back evaluation. Redundant work: 2 of 64 lanes and 3 of LX+3 rows.
// DESIGN.md section 4 lists what was measured on the way (in-order vmcnt, no loads under branches, ...).

struct Ids3 { int m, c, p; };  // obstacle ids at (y-1, y, y+1) of one row

__device__ __forceinline__ Ids3 load_ids(const int* __restrict__ ob, const LatticeView& L, int xl, int y) {
  // branch-free: clamped addresses, then select. (Loads under branches make the compiler fall back to
  // s_waitcnt vmcnt(0) at every control-flow merge, which drains the prefetch pipeline.)
  const bool rok = xl >= 0 && xl < L.nxl;
  const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
  const int* row = ob + (long)xc * L.sy;
  const int ym = y - 1, yp = y + 1;
  const int cm = ym < 0 ? 0 : (ym >= L.ly ? L.ly - 1 : ym);
  const int cc = y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y);
  const int cp = yp < 0 ? 0 : (yp >= L.ly ? L.ly - 1 : yp);
  const int vm = row[cm], vc = row[cc], vp = row[cp];
  Ids3 r;  // off the lattice reads as "wall": never fluid
  r.m = (rok && cm == ym) ? vm : L.n;
  r.c = (rok && cc == y) ? vc : L.n;
  r.p = (rok && cp == yp) ? vp : L.n;
  return r;
}

// `act` of the centre node of a 3x3 block of ids (rows a = x-1, b = x, c = x+1); see Tile::active.
// `own_rec()` returns the record of the grain that owns the centre node (only called on the rare
// path where a neighbour belongs to a higher-index grain). Requires reductionR < 1 (then the paint
// test d2 <= R2 && d2 <= r2 reduces to d2 <= r2); the launcher routes other configurations to the
// LDS-tile kernel.
template <class RecFn>
__device__ __forceinline__ bool node_active(const LatticeView& L, const GrainFluidView& G, const Ids3& a, const Ids3& b,
                                            const Ids3& c, int gx, int gy, RecFn own_rec) {
  const int o = b.c;
  // neighbour ids in direction order 1..8: (-1,1) (-1,0) (-1,-1) (0,-1) (1,-1) (1,0) (1,1) (0,1)
  const int nb[9] = {0, a.p, a.c, a.m, b.m, c.m, c.c, c.p, b.p};
  bool higher = false, fluid = false;
#pragma unroll
  for (int q = 1; q < 9; ++q) {
    fluid |= nb[q] == -1;
    higher |= (nb[q] > o && nb[q] != L.n);
  }
  if (fluid || !higher) return fluid;
  const GP g = own_rec();
  unsigned cand = 0;   // neighbours of a higher-index grain outside the owner's own disc
#pragma unroll
  for (int q = 1; q < 9; ++q) {
    if (nb[q] > o && nb[q] != L.n) {
      const int x = gx + EXq(q), y = gy + EYq(q);
      const real d2 = (x - g.xc) * (x - g.xc) + (y - g.yc) * (y - g.yc);
      if (!(d2 <= g.r2)) cand |= 1u << q;
    }
  }
  if (cand == 0 || G.mincov == nullptr) return cand != 0;
  // ... and not covered by a third disc of lower index either: the rasteriser's lowest-cover record of multiply
  // covered nodes. Global loads, but only where discs overlap (never in a packing at reductionR = 0.85); a small
  // rolled loop so that the marching kernel's register allocation does not feel it.
  bool act = false;
#pragma unroll 1
  while (cand) {
    const int q = __ffs(cand) - 1;
    cand &= cand - 1;
    const int ex = (q >= 1 && q <= 3) ? -1 : ((q >= 5 && q <= 7) ? 1 : 0);
    const int ey = (q == 1 || q >= 7) ? 1 : ((q >= 3 && q <= 5) ? -1 : 0);
    const unsigned m = G.mincov[(long)(gx + ex - L.gx0) * L.sy + (gy + ey)];
    act |= (m >> 20) != (G.epoch & 0xFFFu) || (int)(0xFFFFFu - (m & 0xFFFFFu)) > o;
  }
  return act;
}

// Wave-private LDS ring of grain records: slot [row & 3][lane] holds the record of the grain that owns
// node (row, lane's column), written by that lane when the row was fetched. A bounce-back link at P
// reads the record of its solid neighbour from there: LDS waits use lgkmcnt and do not disturb the
// in-order vmcnt pipeline of the row prefetch, and no load sits inside a divergent path.
constexpr int REC_RING = 4;
struct RecRing {
  real2* base;  // this wave's [REC_RING][4][64] real2
  __device__ __forceinline__ void put(int row, int lane, const GP& g) const {
    real2* p = base + (row & (REC_RING - 1)) * 4 * 64 + lane;
    p[0] = make_real2(g.x1, g.x2);
    p[64] = make_real2(g.v1, g.v2);
    p[128] = make_real2(g.v3, g.xc);
    p[192] = make_real2(g.yc, g.r2);
  }
  // only the lattice-unit centre (xc, yc) of that record
  __device__ __forceinline__ void get_centre(int row, int lane, real& xc, real& yc) const {
    const real* p = reinterpret_cast<const real*>(base + (row & (REC_RING - 1)) * 4 * 64 + lane);
    xc = p[2 * 128 + 1];
    yc = p[2 * 192];
  }
  __device__ __forceinline__ GP get(int row, int lane) const {
    const real2* p = base + (row & (REC_RING - 1)) * 4 * 64 + lane;
    const real2 a = p[0], b = p[64], c = p[128], d = p[192];
    return GP{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
  }
};

struct RegCtx {
  real Fo[9];   // f*[P][q]
  real In[9];   // f*[P + e_d][opp d]
  int o0;
  int onb[9];
  unsigned act;   // bit d: act of the neighbour in direction d
  RecRing ring;
  int row, lane;  // local row of P and this lane
  __device__ __forceinline__ GP gp_nb(int d) const { return ring.get(row + EXq(d), lane + EYq(d)); }
  __device__ __forceinline__ real own(int q) const { return Fo[q]; }
  __device__ __forceinline__ real in(int d) const { return In[d]; }
  __device__ __forceinline__ int o_own() const { return o0; }
  __device__ __forceinline__ int o_nb(int d) const { return onb[d]; }
  __device__ __forceinline__ bool act_nb(int d) const { return (act >> d) & 1u; }
};

#ifdef LBMDEM_AB   /* k_cs_march3 and its helpers only exist in the experiment build */
// the same context for k_cs_march3 (no record ring: the classification never asks for a record)
struct RegCtx3 {
  real Fo[9];
  real In[9];
  int o0;
  int onb[9];
  unsigned act;
  __device__ __forceinline__ real own(int q) const { return Fo[q]; }
  __device__ __forceinline__ real in(int d) const { return In[d]; }
  __device__ __forceinline__ int o_own() const { return o0; }
  __device__ __forceinline__ int o_nb(int d) const { return onb[d]; }
  __device__ __forceinline__ bool act_nb(int d) const { return (act >> d) & 1u; }
};

// classify_store_row for k_cs_march3: ALL nine populations are stored, unconditionally -- a bounce-back link's slot
// gets a placeholder that the compacted pass at the end of the iteration overwrites (one wave's stores to one address
// keep their order). Nine stores per row whatever the row contains: the vector-memory queue between the record DMA and
// the pass that reads the records has a fixed length, so the wait for the records never covers the row prefetch.
template <bool EDGE>
__device__ __forceinline__ void classify_store_all(const RegCtx3& C, const LatticeView& L, int gx, int y,
                                                   real* __restrict__ fout, long node) {
  const long fb = fbase(node);
  fout[fb] = C.own(0);
#define LBM_CLASSIFY_ALL(Q)                                       \
  {                                                               \
    real o_;                                                    \
    if (pull_classify<Q, EDGE>(C, L, gx, y, o_)) o_ = C.own(Q);   \
    fout[fb + Q * F_QSTRIDE(L)] = o_;                             \
  }
  LBM_CLASSIFY_ALL(1) LBM_CLASSIFY_ALL(2) LBM_CLASSIFY_ALL(3) LBM_CLASSIFY_ALL(4)
  LBM_CLASSIFY_ALL(5) LBM_CLASSIFY_ALL(6) LBM_CLASSIFY_ALL(7) LBM_CLASSIFY_ALL(8)
#undef LBM_CLASSIFY_ALL
}

#endif  // LBMDEM_AB

// All nine pulls of a node except the interpolated bounce-back links: the others are stored right away,
// the bounce-back links are only flagged: bit q of `ibb` = link (P, q) needs ibb_eval; `nnm` / `hzm` =
// that link's NN is interior / is a solid node whose own update precedes S's (q <= 4).
template <bool EDGE, class Ctx>
__device__ __forceinline__ void classify_store_row(const Ctx& C, const LatticeView& L, int gx, int y,
                                                   real* __restrict__ fout, long fb, unsigned& ibb,
                                                   unsigned& nnm, unsigned& hzm) {
  fout[fb] = C.own(0);
#define LBM_CLASSIFY(Q)                                                   \
  {                                                                       \
    real o_;                                                            \
    if (pull_classify<Q, EDGE>(C, L, gx, y, o_)) {                        \
      ibb |= 1u << Q;                                                     \
      if (nn_interior<Q, EDGE>(L, gx, y)) {                               \
        nnm |= 1u << Q;                                                   \
        if (Q <= 4 && C.o_nb(Q) != -1) hzm |= 1u << Q;                    \
      }                                                                   \
    } else {                                                              \
      fout[fb + Q * F_QSTRIDE(L)] = o_;                                   \
    }                                                                     \
  }
  LBM_CLASSIFY(1) LBM_CLASSIFY(2) LBM_CLASSIFY(3) LBM_CLASSIFY(4)
  LBM_CLASSIFY(5) LBM_CLASSIFY(6) LBM_CLASSIFY(7) LBM_CLASSIFY(8)
#undef LBM_CLASSIFY
}

#ifndef MARCH_BRANCHY   /* -DMARCH_BRANCHY (experiment builds) restores the branchy form for deep rows too */
// classify_store_row for rows and lanes at least two nodes away from every lattice edge (S, P and NN all interior),
// without a branch: the eight pulls differ only in WHICH value they take -- the streamed population, or the weight w_q
// when both ends are solid and the source is an active node (main.c:1161-1162) -- and all of them are stored; the slot of
// a bounce-back link gets the streamed value as a placeholder, which the compacted pass that follows overwrites (the
// stores of one wavefront to one address keep their order; a link whose wall distance fires neither formula keeps exactly
// this value, main.c:1166-1217). ~13 instructions per direction instead of three nested divergent branches: 1.6 % of the
// kernel (A/B on one GPU, four interleaved pairs).
template <class Ctx>
__device__ __forceinline__ void classify_store_row_deep(const Ctx& C, const LatticeView& L, real* __restrict__ fout,
                                                        long fb, unsigned& ibb, unsigned& nnm, unsigned& hzm) {
  fout[fb] = C.own(0);
  const bool own_solid = C.o_own() != -1;
#define LBM_CLASSIFY_DEEP(Q)                                                        \
  {                                                                                 \
    const bool src_solid = C.o_nb(OPPq(Q)) != -1;                                   \
    const bool reset = src_solid & own_solid & C.act_nb(OPPq(Q));                   \
    const real in_ = C.in(OPPq(Q));                                                 \
    fout[fb + Q * F_QSTRIDE(L)] = reset ? Wq(Q) : in_;                              \
    const unsigned link = (src_solid & !own_solid) ? 1u << Q : 0u;                  \
    ibb |= link;                                                                    \
    if (Q <= 4) hzm |= C.o_nb(Q) != -1 ? link : 0u;                                 \
  }
  LBM_CLASSIFY_DEEP(1) LBM_CLASSIFY_DEEP(2) LBM_CLASSIFY_DEEP(3) LBM_CLASSIFY_DEEP(4)
  LBM_CLASSIFY_DEEP(5) LBM_CLASSIFY_DEEP(6) LBM_CLASSIFY_DEEP(7) LBM_CLASSIFY_DEEP(8)
#undef LBM_CLASSIFY_DEEP
  nnm = ibb;
}
#endif

// number of set bits of `m` below this lane
__device__ __forceinline__ unsigned mbcnt(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}
// inclusive prefix sum over the 64 lanes of a wave with DPP row shifts / row broadcasts (no LDS traffic)
__device__ __forceinline__ int wave_inclusive_scan(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);  // row_shr:1
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);  // row_shr:2
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);  // row_shr:4
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);  // row_shr:8
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 -> rows 1, 3
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 -> rows 2, 3
  return v;
}
// the neighbour lanes' values of a pull: full-wave DPP shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1, GFX9) -- no LDS
// round trip, no address registers; lane i <- lane i-1 / lane i+1, the end lane keeps its value. DPP reads nothing from
// a lane that is switched off: only call these in wave-uniform control flow.
__device__ __forceinline__ int dpp_up1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int dpp_dn1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
#ifdef LBMDEM_SINGLE_PRECISION
__device__ __forceinline__ real dpp_up1(real v) { return __int_as_float(dpp_up1(__float_as_int(v))); }
__device__ __forceinline__ real dpp_dn1(real v) { return __int_as_float(dpp_dn1(__float_as_int(v))); }
#else
__device__ __forceinline__ real dpp_up1(real v) {
  return __hiloint2double(dpp_up1(__double2hiint(v)), dpp_up1(__double2loint(v)));
}
__device__ __forceinline__ real dpp_dn1(real v) {
  return __hiloint2double(dpp_dn1(__double2hiint(v)), dpp_dn1(__double2loint(v)));
}
#endif
#ifdef MARCH_BPERMUTE   /* experiment builds: the round-2 form (ds_bpermute) */
__device__ __forceinline__ real shfl_up1(real v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ real shfl_dn1(real v) { return __shfl_down(v, 1, 64); }
__device__ __forceinline__ int shfl_up1(int v) { return __shfl_up(v, 1, 64); }
__device__ __forceinline__ int shfl_dn1(int v) { return __shfl_down(v, 1, 64); }
#else
__device__ __forceinline__ real shfl_up1(real v) { return dpp_up1(v); }
__device__ __forceinline__ real shfl_dn1(real v) { return dpp_dn1(v); }
__device__ __forceinline__ int shfl_up1(int v) { return dpp_up1(v); }
__device__ __forceinline__ int shfl_dn1(int v) { return dpp_dn1(v); }
#endif

#ifdef MARCH_TIMING   /* experiment builds: where a wavefront's cycles go (s_memtime between the phases of an iteration) */
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
#else
#define MT_DECL
#define MT(i)
#define MT_FLUSH
#endif

template <int LX, int MINW, int WW>
__global__ __launch_bounds__(256, MINW) void k_cs_march(const real* __restrict__ fin, real* __restrict__ fout,
                                                  const int* __restrict__ ob_old,
                                                  const int* __restrict__ ob_new, LatticeView L,
                                                  GrainFluidView G, ForceSlots S, int nstrips, int nwork,
                                                  int xcd_remap, int seg_rows, int seg_stride) {
  const int lane = threadIdx.x & 63;
  int blk = blockIdx.x;
  if (xcd_remap) {  // XCD k (blocks b % 8 == k) walks the k-th contiguous eighth of the work list
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  const int w = blk * 4 + (threadIdx.x >> 6);
  if (w >= nwork) return;  // whole wave
  const int strip = w % nstrips, seg = w / nstrips;
  // WW producing lanes in the middle of the window, (64 - WW) / 2 feeding lanes on either side
  constexpr int OFF = (64 - WW) / 2;
  const int y = strip * WW - OFF + lane;
  const bool yin = y >= 0 && y < L.ly;
  const bool writer = lane >= OFF && lane < OFF + WW && yin;
  const bool deep_y = strip * WW >= 2 && strip * WW + WW - 1 <= L.ly - 3;  // the producing lanes
  // rows per wave: the template value, or (LX == 0) a run-time value chosen so that one round of resident
  // waves covers the lattice
  const int rows_per_wave = LX > 0 ? LX : seg_rows;
  // segment k starts seg_stride rows after segment k-1: = rows_per_wave for a contiguous row range; larger when one
  // launch covers the two edge-row ranges of a strip
  const int xs = L.xo0 + seg * seg_stride;
  const int xe = xs + rows_per_wave < L.xo1 ? xs + rows_per_wave : L.xo1;

  // Software pipeline. In iteration x (producing row x) the wave issues, in this order,
  //   (1) small gathers: new ids of row x+4, previous-map id of row x+4, the record of the grain that
  //       owned (x+2, y) before (reinit of row x+2), the record of the grain that owns (x+3, y) now
  //   (2) the nine populations of row x+3 into the row buffer it has just consumed (two buffers,
  //       ping-pong, loop unrolled by two: no register copies, so the loads stay in flight for two
  //       iterations)
  //   (3) the nine stores of row x.
  // gfx9 retires vector-memory operations in issue order (one vmcnt counter), so data must be consumed
  // in the order it was requested; no other global load exists inside the loop.
  __shared__ real2 sRec[4 * REC_RING * 4 * 64];
  const RecRing ring{sRec + (threadIdx.x >> 6) * (REC_RING * 4 * 64)};
  // wave-private scratch for the compacted bounce-back evaluation: 64 link slots
  // (Round 3 measured a spare 65th slot that lanes without a link write to, instead of sitting out the writes under an
  // exec mask: eight mask round trips per row less, but 12 B of scratch and twice the scalar spill reloads: +1.7 %.)
  constexpr int LINK_SLOTS = 64;
  __shared__ real sPay[4 * LINK_SLOTS * 4];
  __shared__ int sDesc[4 * LINK_SLOTS];
  real* const pay = sPay + (threadIdx.x >> 6) * (LINK_SLOTS * 4);
  int* const desc = sDesc + (threadIdx.x >> 6) * LINK_SLOTS;
  const real wc_diag = L.wc_diag, wc_axis = L.wc_axis;  // kernel arguments: scalar registers
  auto row_ok = [&](int xl) { return yin && xl >= 0 && xl < L.nxl; };
  const int ycl = y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y);
  auto node_of = [&](int xl) {  // clamped: always a valid address
    const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
    return (long)xc * L.sy + ycl;
  };
  // unconditional (clamped address): every use is guarded by interior(xl), and grain_rec clamps the
  // id. (A `row_ok ? v : -1` select here makes the compiler sink the load into a branch followed by
  // s_waitcnt vmcnt(0), which drains the whole prefetch pipeline once per iteration.)
  auto load_old = [&](int xl) { return ob_old[node_of(xl)]; };
  // off-lattice positions load a clamped neighbour's values; they are never used (pull_one tests the
  // bounds of the source node before touching its populations)
  auto load_raw = [&](int xl, real (&raw)[9]) {
    const long fb = fbase_xy(L, xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl), ycl);
#pragma unroll
    for (int q = 0; q < 9; ++q) raw[q] = fin[fb + q * F_QSTRIDE(L)];
  };
  auto interior = [&](int xl) {
    const int gx = L.gx0 + xl;
    return row_ok(xl) && gx >= 1 && gx <= L.lx - 2 && y >= 1 && y <= L.ly - 2;
  };
  // f* of one node: reinit (previous map) + collide (current map)
  auto make_fstar = [&](int xl, real (&f)[9], int oo, const GP& g, int on) {
    const bool in = interior(xl);
    if (in && oo != -1) grain_equilibrium(L, g, L.gx0 + xl, y, f);
    if (in && on == -1) mrt_collide(L, f);
  };
  auto grain_rec = [&](int id) { return load_gp(G, (id < 0 || id >= L.n) ? 0 : id); };

  real Fm[9], F0[9], Fp[9], bufA[9], bufB[9];

  Ids3 iA = load_ids(ob_new, L, xs - 2, y);  // row x-2 (only needed for act of row x-1)
  Ids3 iB = load_ids(ob_new, L, xs - 1, y);  // row x-1
  Ids3 iC = load_ids(ob_new, L, xs, y);      // row x
  Ids3 iD = load_ids(ob_new, L, xs + 1, y);  // row x+1
  Ids3 iE = load_ids(ob_new, L, xs + 2, y);  // row x+2
  {
    int oo = load_old(xs - 1);
    load_raw(xs - 1, Fm);
    make_fstar(xs - 1, Fm, oo, grain_rec(oo), iB.c);
    oo = load_old(xs);
    load_raw(xs, F0);
    make_fstar(xs, F0, oo, grain_rec(oo), iC.c);
  }
  // records of the current owners of rows x-1 .. x+2 into the ring
  ring.put(xs - 1, lane, grain_rec(iB.c));
  ring.put(xs, lane, grain_rec(iC.c));
  ring.put(xs + 1, lane, grain_rec(iD.c));
  ring.put(xs + 2, lane, grain_rec(iE.c));
  int oo1 = load_old(xs + 1);   // previous-map ids of rows x+1, x+2, x+3
  int oo2 = load_old(xs + 2);
  int oo3 = load_old(xs + 3);
  Ids3 inext = load_ids(ob_new, L, xs + 3, y);
  GP gre = grain_rec(oo1);      // reinit record for row x+1
  GP rec_next = grain_rec(inext.c);  // owner record of row x+3, goes into the ring next iteration
  load_raw(xs + 1, bufA);
  load_raw(xs + 2, bufB);
  bool actm = iB.c != -1 && node_active(L, G, iA, iB, iC, L.gx0 + xs - 1, y, [&] { return ring.get(xs - 1, lane); });
  bool act0 = iC.c != -1 && node_active(L, G, iB, iC, iD, L.gx0 + xs, y, [&] { return ring.get(xs, lane); });

  MT_DECL
  // one iteration; `buf` holds row x+1 on entry and is refilled with row x+3
  auto iterate = [&](int x, real (&buf)[9]) {
    MT(7)   // loop head
#pragma unroll
    for (int q = 0; q < 9; ++q) Fp[q] = buf[q];
#ifdef MARCH_TIMING   /* phase 0 = the wait for the row's populations alone */
#pragma unroll
    for (int q = 0; q < 9; ++q) asm volatile("" ::"v"(Fp[q]));
    MT(0)
#endif
    make_fstar(x + 1, Fp, oo1, gre, iD.c);
    MT(1)
    const Ids3 iF = inext;  // row x+3
    // ---- (1) small gathers
    oo1 = oo2;
    oo2 = oo3;
    gre = grain_rec(oo1);                      // reinit record of row x+2
    inext = load_ids(ob_new, L, x + 4, y);
    oo3 = load_old(x + 4);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (2) the big loads: populations of row x+3
    load_raw(x + 3, buf);
    __builtin_amdgcn_sched_barrier(0);
    MT(2)
    // (the column re-declared opaque: otherwise (double)(y - 1), (double)y, (double)(y + 1) of node_active's rare path are
    // hoisted out of the loop into six registers the kernel does not have, and one pair ends up in scratch -- whose reload
    // drains the row prefetch)
    int y_act = y;
    asm volatile("" : "+v"(y_act));
    const bool actp = iD.c != -1 && node_active(L, G, iC, iD, iE, L.gx0 + x + 1, y_act, [&] { return ring.get(x + 1, lane); });
    MT(3)

    RegCtx C;
    C.ring = ring;
    C.row = x;
    C.lane = lane;
#pragma unroll
    for (int q = 0; q < 9; ++q) C.Fo[q] = F0[q];
    C.In[0] = 0.0;
    C.In[2] = Fm[6];            // (-1, 0): same lane, row x-1, slot opp(2) = 6
    C.In[6] = Fp[2];            // ( 1, 0)
    C.In[1] = shfl_dn1(Fm[5]);  // (-1, 1): lane+1, row x-1, slot 5
    C.In[8] = shfl_dn1(F0[4]);  // ( 0, 1)
    C.In[7] = shfl_dn1(Fp[3]);  // ( 1, 1)
    C.In[3] = shfl_up1(Fm[7]);  // (-1,-1): lane-1
    C.In[4] = shfl_up1(F0[8]);  // ( 0,-1)
    C.In[5] = shfl_up1(Fp[1]);  // ( 1,-1)
    C.o0 = iC.c;
    C.onb[0] = 0;
    C.onb[1] = iB.p; C.onb[2] = iB.c; C.onb[3] = iB.m; C.onb[4] = iC.m;
    C.onb[5] = iD.m; C.onb[6] = iD.c; C.onb[7] = iD.p; C.onb[8] = iC.p;
    const int pack = (actm ? 1 : 0) | (act0 ? 2 : 0) | (actp ? 4 : 0);  // rows x-1, x, x+1 of this lane
    const int pk_up = shfl_dn1(pack);                                  // lane+1 (y+1)
    const int pk_dn = shfl_up1(pack);                                  // lane-1 (y-1)
    C.act = (((pk_up >> 0) & 1u) << 1) | (((pack >> 0) & 1u) << 2) | (((pk_dn >> 0) & 1u) << 3) |
            (((pk_dn >> 1) & 1u) << 4) | (((pk_dn >> 2) & 1u) << 5) | (((pack >> 2) & 1u) << 6) |
            (((pk_up >> 2) & 1u) << 7) | (((pk_up >> 1) & 1u) << 8);
    {
      const int gx = L.gx0 + x;
      const long fb_row = fbase_xy(L, x, y);   // (only used by the producing lanes: y is on the lattice there)
      // wave-uniform: is every producing lane of this row at least two nodes away from all edges?
      const bool deep = deep_y && gx >= 2 && gx <= L.lx - 3;
      // (a) everything but the interpolated bounce-back links: computed and stored
      unsigned ibb = 0, nnm = 0, hzm = 0;
      if (writer && x < xe) {
#ifndef MARCH_BRANCHY
        if (deep) classify_store_row_deep(C, L, fout, fb_row, ibb, nnm, hzm);
#else
        if (deep) classify_store_row<false>(C, L, gx, y, fout, fb_row, ibb, nnm, hzm);
#endif
        else classify_store_row<true>(C, L, gx, y, fout, fb_row, ibb, nnm, hzm);
      }
      MT(4)
      // (b) the bounce-back links of the whole row (typically ~20, spread over all eight directions
      // and a few lanes) are compacted into dense lanes through LDS and evaluated in ONE pass with the
      // direction as data, instead of ~3.5 direction-specific divergent passes of ~130 instructions.
      // slot of link (lane, q) = number of links in directions < q + number in direction q on lower lanes
      int T = 0;
#pragma unroll
      for (int q = 1; q < 9; ++q) T += __popcll(__ballot((ibb >> q) & 1u));
      for (int base = 0; base < T; base += 64) {  // wave-uniform; a second round only if > 64 links
        int before = 0;
#pragma unroll
        for (int q = 1; q < 9; ++q) {
          const unsigned long long b = __ballot((ibb >> q) & 1u);
          const int t = before + (int)mbcnt(b) - base;
          before += __popcll(b);
          if (((ibb >> q) & 1u) && t >= 0 && t < 64) {
            // bits 14..31: the grain that owns S = P - e_q (the slot table is only used with < 2^18 grains)
            desc[t] = lane | (q << 8) | (((nnm >> q) & 1u) << 12) | (((hzm >> q) & 1u) << 13) | (C.onb[OPPq(q)] << 14);
            pay[t * 4 + 0] = C.Fo[OPPq(q)];
            pay[t * 4 + 1] = C.Fo[q];
            pay[t * 4 + 2] = C.In[q];
            pay[t * 4 + 3] = C.In[OPPq(q)];
          }
        }
        __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
        if (base + lane < T) {
          const int d = desc[lane];
          const int src = d & 63;
          RtLink k;
          k.q = (d >> 8) & 15;
          k.gx = gx;
          k.gy = y - lane + src;
          k.own_qo = pay[lane * 4 + 0];
          k.own_q = pay[lane * 4 + 1];
          k.in_q = pay[lane * 4 + 2];
          k.in_qo = pay[lane * 4 + 3];
          k.nn_int = (d >> 12) & 1;
          k.hazard = (d >> 13) & 1;
          // the result goes straight to the population it belongs to: node of lane `src`, plane q
          const real out =
              ibb_eval_rt(L, k, wc_diag, wc_axis, [&](int dx, int dy) { return ring.get(x + dx, src + dy); });
          fout[fbase_xy(L, x, k.gy) + k.q * F_QSTRIDE(L)] = out;
          // ... and the link's momentum-exchange sum f_new[S][opp q] + f_new[P][q] (main.c:1313-1316; the first
          // is f*[P][opp q], streamed unchanged into the solid node) to the owning grain's slot table
          if (S.tab != nullptr) {
            const int ex = (k.q >= 1 && k.q <= 3) ? -1 : ((k.q >= 5 && k.q <= 7) ? 1 : 0);
            const int ey = (k.q == 1 || k.q >= 7) ? 1 : ((k.q >= 3 && k.q <= 5) ? -1 : 0);
            real cx, cy;
            ring.get_centre(x - ex, src - ey, cx, cy);
            const int rel = slot_line(k.gx - ex, k.gy - ey, ex, ey, cx, cy) + S.half;
            if ((unsigned)rel < (unsigned)S.spd)
              S.tab[((long)((unsigned)d >> 14) * 8 + (k.q - 1)) * S.spd + rel] = k.own_qo + out;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      MT(5)
    }
    // row x-1 is no longer needed: its ring slot takes the owner records of row x+3; then request
    // those of row x+4 (consumed at this point of the next iteration)
    ring.put(x + 3, lane, rec_next);
    rec_next = grain_rec(inext.c);
    // rotate
#pragma unroll
    for (int q = 0; q < 9; ++q) { Fm[q] = F0[q]; F0[q] = Fp[q]; }
    iB = iC; iC = iD; iD = iE; iE = iF;
    actm = act0; act0 = actp;
    MT(6)
  };

  // Both halves run unconditionally (a row >= xe stores nothing): with `if (x + 1 < xe)` around the second one the
  // compiler cannot count its loads as younger than the first half's when it places s_waitcnt at the loop head, and
  // waits for more of the pipeline than the data it needs (vmcnt(4) instead of vmcnt(9); 1.5-2 % of the kernel).
  for (int x = xs; x < xe; x += 2) {
    iterate(x, bufA);
#ifdef MARCH_COND2   /* experiment builds: the former form */
    if (x + 1 < xe)
#endif
    iterate(x + 1, bufB);
  }
  MT_FLUSH
}

#ifdef LBMDEM_AB
// ---------------------------------------------------------------------------------------------
// the fused fluid kernel, marching form for THREE wavefronts per SIMD (round 3)
// ---------------------------------------------------------------------------------------------
//
// Same algorithm and arithmetic as k_cs_march, re-organised so that a wavefront needs <= 168 VGPRs and
// 6.4 KB of LDS instead of 250 VGPRs and 18 KB (k_cs_march is limited to two waves per SIMD by BOTH):
//  * no per-node ring of grain records. The bounce-back links of a row depend only on the obstacle ids, so
//    they are found and compacted (lane-major: one DPP prefix scan) at the START of the row's iteration and
//    the record of each link's grain is fetched straight into LDS by the dense lane that will evaluate the
//    link (global_load_lds_dwordx4: no VGPR landing, no ds_write) -- ~20 records per row instead of 64;
//    they arrive while row x+1 is collided. The rare consumers of other records (the hazard partner of a
//    link, `act` next to a higher-index grain) load them inside their own branch.
//  * the six cross-lane moves of a pull are full-wave DPP shifts (v_mov_b32_dpp wave_shr:1 / wave_shl:1)
//    instead of ds_bpermute: no LDS round trip, no address registers.
//  * the LDS-DMA is issued by inline assembly (hipcc drains the whole VM queue at the next use of an ordinary
//    load while a DMA it knows about is in flight) and retired by ONE counted s_waitcnt: gfx9 retires VM operations
//    in order, and exactly 16 unconditional loads (7 small gathers + the 9 populations of row x+3) are issued
//    between the DMA and the wait, so vmcnt(16) is precisely "the DMA has landed".


// 16 bytes per active lane from `gsrc` (per lane) to LDS byte address lds_dst (wave-uniform) + 16 * lane
__device__ __forceinline__ void lds_dma16(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// obstacle ids of one row as this lane sees them: its own column and, in the two end lanes of the wave, the column
// just outside the window (lane 0: y-1, lane 63: y+1); the neighbours' ids come from the adjacent lanes (DPP)
struct IdsRow {
  int c, outer;
  __device__ __forceinline__ int m() const { return __builtin_amdgcn_update_dpp(outer, c, 0x138, 0xf, 0xf, false); }
  __device__ __forceinline__ int p() const { return __builtin_amdgcn_update_dpp(outer, c, 0x130, 0xf, 0xf, false); }
  __device__ __forceinline__ Ids3 all() const { return Ids3{m(), c, p()}; }
};
__device__ __forceinline__ IdsRow load_ids_row(const int* __restrict__ ob, const LatticeView& L, int xl, int y, int lane) {
  const bool rok = xl >= 0 && xl < L.nxl;
  const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
  const int* row = ob + (long)xc * L.sy;
  const int yo = lane == 0 ? y - 1 : (lane == 63 ? y + 1 : y);
  const int cc = y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y);
  const int co = yo < 0 ? 0 : (yo >= L.ly ? L.ly - 1 : yo);
  const int vc = row[cc], vo = row[co];
  IdsRow r;  // off the lattice reads as "wall": never fluid
  r.c = (rok && cc == y) ? vc : L.n;
  r.outer = (rok && co == yo) ? vo : L.n;
  return r;
}

// the last piece of a record: the source address also comes OUT of the statement, which ties a later load to it
__device__ __forceinline__ void lds_dma16_tok(const char*& gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "+v"(gsrc) : "s"(lds_dst) : "memory");
}

#ifdef M3_MANUAL
// Loads the compiler does not see as loads (asm): it places no s_waitcnt of its own for them -- a counted wait after a
// run of conditional stores can only assume that none of them was issued, and so drains the queue -- and the kernel
// waits by hand with the exact number of vector-memory operations it has issued since (gfx9 retires them in order).
typedef int m3_v4i __attribute__((ext_vector_type(4)));
typedef int m3_v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int m3_load_b32(const char* sbase, unsigned voff) {
  int v;
  asm volatile("global_load_dword %0, %1, %2" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
__device__ __forceinline__ m3_v4i m3_load_b128(const void* p) {
  m3_v4i v;
  asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ m3_v4i m3_load_b128_16(const void* p) {
  m3_v4i v;
  asm volatile("global_load_dwordx4 %0, %1, off offset:16" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ m3_v2i m3_load_b64_32(const void* p) {
  m3_v2i v;
  asm volatile("global_load_dwordx2 %0, %1, off offset:32" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ double m3_dbl(int lo, int hi) { return __hiloint2double(hi, lo); }
#endif

#ifdef M3_DMAPOP
// 16 bytes per lane from (sbase + voff) to LDS byte address lds_dst (wave-uniform) + 16 * lane; the per-lane offset also
// comes OUT of the statement so that a later ordinary load can be tied to it (issued after the DMA)
__device__ __forceinline__ void lds_dma16_s(const char* sbase, unsigned& voff, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "+v"(voff) : "s"(sbase), "s"(lds_dst) : "memory");
}
#endif

template <int LX, int MINW, int WW, int NBUF>
__global__ __launch_bounds__(256, MINW) void k_cs_march3(const real* __restrict__ fin, real* __restrict__ fout,
                                                   const int* __restrict__ ob_old,
                                                   const int* __restrict__ ob_new, LatticeView L,
                                                   GrainFluidView G, ForceSlots S, int nstrips, int nwork,
                                                   int xcd_remap, int seg_rows, int seg_stride) {
#ifdef M3_UNIFORM
  int lane = threadIdx.x & 63;   // (not const: re-declared opaque in every iteration, see iterate())
#else
  const int lane = threadIdx.x & 63;
#endif
  int blk = blockIdx.x;
  if (xcd_remap) {
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
#ifdef M3_UNIFORM   /* wave-uniform, said explicitly: the row counter and the row addresses then live in scalar registers */
  const int w = __builtin_amdgcn_readfirstlane(blk * 4 + (int)(threadIdx.x >> 6));
#else
  const int w = blk * 4 + (threadIdx.x >> 6);
#endif
  if (w >= nwork) return;  // whole wave
  const int strip = w % nstrips, seg = w / nstrips;
  constexpr int OFF = (64 - WW) / 2;
  static_assert(OFF >= 1, "the end lanes only feed their neighbours");
#ifdef M3_UNIFORM
  int y = strip * WW - OFF + lane;
#else
  const int y = strip * WW - OFF + lane;
#endif
  const bool yin = y >= 0 && y < L.ly;
  const bool writer = lane >= OFF && lane < OFF + WW && yin;
  const bool deep_y = strip * WW >= 2 && strip * WW + WW - 1 <= L.ly - 3;
  const int rows_per_wave = LX > 0 ? LX : seg_rows;
  const int xs = L.xo0 + seg * seg_stride;
  const int xe = xs + rows_per_wave < L.xo1 ? xs + rows_per_wave : L.xo1;

  // wave-private LDS: link descriptors, link payloads, link grain records (DMA target: [part][dense lane])
  __shared__ real2 sLrec[4 * 4 * 64];
  __shared__ real sPay[4 * 64 * 4];
  __shared__ int sDesc[4 * 64];
  const int wv = threadIdx.x >> 6;
  real2* const lrec = sLrec + wv * (4 * 64);
  real* const pay = sPay + wv * (64 * 4);
  int* const desc = sDesc + wv * 64;
  const unsigned lrec_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)lrec);
#ifdef M3_DMAPOP
  // The populations of the next row travel global -> LDS without a register landing (5 x global_load_lds_dwordx4 per
  // row): chunk g = 64 k + lane (k = 0..4) is the 16-byte pair (y_c, y_c + 1), y_c = y0 - 1 + 2 (g % 33) (even: y0 is odd
  // for WW = 62), of direction g / 33; it lands at staging byte 16 g, so direction q of lane l is double 66 q + l + 1.
  static_assert(WW == 62 && NBUF == 1 && sizeof(real) == 8, "the DMA staging is laid out for the 62-column window of doubles");
  __shared__ real sStage[4 * 640];
  real* const stage = sStage + wv * 640;
  const unsigned stage_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)stage);
  unsigned dma_off[5];   // byte offset of this lane's chunk from the first tile of the row
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int g = 64 * k + lane;
    if (g > 296) g = 296;
    const int q = g / 33, c = g % 33;
    int yc = strip * WW - OFF - 1 + 2 * c;
    yc = yc < 0 ? 0 : (yc > L.sy - 2 ? L.sy - 2 : yc);
    dma_off[k] = (unsigned)(((yc / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (yc % LBMDEM_TILE_Y) + q * LBMDEM_TILE_Y) * 8);
  }
  typedef const int __attribute__((address_space(1))) * gint_ptr_t;
  // request row xl (clamped like node_of); returns the probe whose arrival means the row has landed
  auto dma_row = [&](int xl) {
    const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
    // (wave-uniform, but derived from threadIdx.x >> 6: say so, the DMA wants its base in scalar registers)
    const unsigned long long rbv =
        (unsigned long long)(reinterpret_cast<const char*>(fin) + (long)xc * (L.sy / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y * 8));
    const char* rb = reinterpret_cast<const char*>(
        ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(rbv >> 32)) << 32) |
        (unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)rbv));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the reads of the previous row out of the staging are over
    lds_dma16_s(rb, dma_off[0], stage_lds);
    lds_dma16_s(rb, dma_off[1], stage_lds + 1024);
    lds_dma16_s(rb, dma_off[2], stage_lds + 2048);
    lds_dma16_s(rb, dma_off[3], stage_lds + 3072);
    lds_dma16_s(rb, dma_off[4], stage_lds + 4096);
#ifdef M3_MANUAL
    return 0;
#else
    return *(gint_ptr_t)(unsigned long long)(rb + dma_off[4]);
#endif
  };
  auto stage_read = [&](int probe, real (&f)[9]) {
#ifndef M3_MANUAL
    asm volatile("" ::"v"(probe) : "memory");   // the compiler's counted wait for the probe: the DMA before it has landed
#endif
#pragma unroll
    for (int q = 0; q < 9; ++q) f[q] = stage[66 * q + lane + 1];
  };
#endif
  const real wc_diag = L.wc_diag, wc_axis = L.wc_axis;
  auto row_ok = [&](int xl) { return yin && xl >= 0 && xl < L.nxl; };
  const int ycl = y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y);
  auto node_of = [&](int xl) {
    const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
    return (long)xc * L.sy + ycl;
  };
#ifdef M3_UNIFORM
  // scalar row base + 32-bit lane byte offset, the offset re-declared opaque at every use: otherwise the compiler folds it
  // into a loop-invariant 64-bit per-lane pointer (two registers each, hoisted out of the loop and then spilled)
  const int yo_ = lane == 0 ? y - 1 : (lane == 63 ? y + 1 : y);
  const int co_ = yo_ < 0 ? 0 : (yo_ >= L.ly ? L.ly - 1 : yo_);
  const unsigned ycl4 = 4u * (unsigned)ycl, co4 = 4u * (unsigned)co_;
  const bool c_in = ycl == y, o_in = co_ == yo_;
  auto opaque = [](unsigned v) { asm volatile("" : "+v"(v)); return v; };
  auto row_of = [&](const int* ob, int xl) {
    const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
    return reinterpret_cast<const char*>(ob) + (long)xc * L.sy * 4;
  };
  auto load_old = [&](int xl) { return *reinterpret_cast<const int*>(row_of(ob_old, xl) + opaque(ycl4)); };
  auto load_ids_row_u = [&](int xl) {
    const bool rok = xl >= 0 && xl < L.nxl;
    const char* row = row_of(ob_new, xl);
    const int vc = *reinterpret_cast<const int*>(row + opaque(ycl4));
    const int vo = *reinterpret_cast<const int*>(row + opaque(co4));
    IdsRow r;
    r.c = (rok && c_in) ? vc : L.n;
    r.outer = (rok && o_in) ? vo : L.n;
    return r;
  };
#define M3_LOAD_IDS(xl) load_ids_row_u(xl)
#else
  auto load_old = [&](int xl) { return ob_old[node_of(xl)]; };
#define M3_LOAD_IDS(xl) load_ids_row(ob_new, L, xl, y, lane)
#endif
  auto load_raw = [&](int xl, real (&raw)[9]) {
    const long fb = fbase(node_of(xl));
#pragma unroll
    for (int q = 0; q < 9; ++q) raw[q] = fin[fb + q * F_QSTRIDE(L)];
  };
  auto interior = [&](int xl) {
    const int gx = L.gx0 + xl;
    return row_ok(xl) && gx >= 1 && gx <= L.lx - 2 && y >= 1 && y <= L.ly - 2;
  };
  // f* of one node: reinit (previous map; the old owner's rigid-body velocity at the node) + collide (current map)
  auto make_fstar = [&](int xl, real (&f)[9], int oo, real ux, real uy, int on) {
    const bool in = interior(xl);
    if (in && oo != -1) grain_equilibrium_u(L, ux, uy, f);
    if (in && on == -1) mrt_collide(L, f);
  };
  // what the re-initialisation needs of a record: x1, x2, v1, v2, v3
  struct RRec { real2 a, b; real v3; };
  auto reinit_rec = [&](int id) {
    const real* p = G.pk + (long)((id < 0 || id >= L.n) ? 0 : id) * 8;
    return RRec{*reinterpret_cast<const real2*>(p), *reinterpret_cast<const real2*>(p + 2), p[4]};
  };
  auto rr_ux = [&](const RRec& r) { return r.b.x - (y * L.dx + L.Mby - r.a.y) * r.v3; };        // wall_ux
  auto rr_uy = [&](const RRec& r, int xl) { return r.b.y + ((L.gx0 + xl) * L.dx + L.Mgx - r.a.x) * r.v3; };  // wall_uy
  auto grain_rec = [&](int id) { return load_gp(G, (id < 0 || id >= L.n) ? 0 : id); };

  real Fm[9], F0[9], Fp[9], bufA[9], bufB[9];   // NBUF = 1: only bufA

  IdsRow iB = M3_LOAD_IDS(xs - 1);
  IdsRow iC = M3_LOAD_IDS(xs);
  IdsRow iD = M3_LOAD_IDS(xs + 1);
  IdsRow iE = M3_LOAD_IDS(xs + 2);
  bool actm, act0;
  {
    const IdsRow iA = M3_LOAD_IDS(xs - 2);
    int oo = load_old(xs - 1);
    load_raw(xs - 1, Fm);
    RRec r = reinit_rec(oo);
    make_fstar(xs - 1, Fm, oo, rr_ux(r), rr_uy(r, xs - 1), iB.c);
    oo = load_old(xs);
    load_raw(xs, F0);
    r = reinit_rec(oo);
    make_fstar(xs, F0, oo, rr_ux(r), rr_uy(r, xs), iC.c);
    const Ids3 a3 = iA.all(), b3 = iB.all(), c3 = iC.all(), d3 = iD.all();   // DPP: outside the divergent &&
    actm = iB.c != -1 && node_active(L, G, a3, b3, c3, L.gx0 + xs - 1, y, [&] { return grain_rec(iB.c); });
    act0 = iC.c != -1 && node_active(L, G, b3, c3, d3, L.gx0 + xs, y, [&] { return grain_rec(iC.c); });
  }
  int oo1 = load_old(xs + 1);   // previous-map ids of rows x+1, x+2
  int oo2 = load_old(xs + 2);
  RRec gre = reinit_rec(oo1);   // reinit record of row x+1
#ifdef M3_MANUAL
  // what is in flight across an iteration boundary: ids + previous-map id of row x+2, the reinit record of row x+1
  int pend_c, pend_o, pend_old;
  m3_v4i pend_ra, pend_rb;
  m3_v2i pend_rc;
  auto request_ids = [&](int xl) {
    const char* row = row_of(ob_new, xl);
    pend_c = m3_load_b32(row, opaque(ycl4));
    pend_o = m3_load_b32(row, opaque(co4));
    pend_old = m3_load_b32(row_of(ob_old, xl), opaque(ycl4));
  };
  auto request_reinit = [&](int id) {
    const real* p = G.pk + (long)((id < 0 || id >= L.n) ? 0 : id) * 8;
    pend_ra = m3_load_b128(p);
    pend_rb = m3_load_b128_16(p);
    pend_rc = m3_load_b64_32(p);
  };
  request_ids(xs + 2);
  request_reinit(oo1);
  int pop_probe = dma_row(xs + 1);
  // everything requested so far has landed before the first iteration (the counted waits inside the loop count the
  // operations of ONE iteration)
  asm volatile("s_waitcnt vmcnt(0)" : "+v"(pend_c), "+v"(pend_o), "+v"(pend_old), "+v"(pend_ra), "+v"(pend_rb), "+v"(pend_rc) :: "memory");
#elif defined(M3_DMAPOP)
  int pop_probe = dma_row(xs + 1);
#else
  load_raw(xs + 1, bufA);
  if (NBUF == 2) load_raw(xs + 2, bufB);
#endif

  // one iteration: `buf` holds row x+1 on entry and is refilled with row x+1+NBUF
  auto iterate = [&](int x, real (&buf)[9]) {
#ifdef M3_UNIFORM
    // Nothing derived from the lane or the column may be hoisted out of the loop: the compiler otherwise keeps dozens of
    // trivial loop invariants (lane | q << 8, (double)(y +- 1), per-lane pointers) in registers it does not have, and
    // spills them to scratch -- whose reloads sit in the same in-order queue as the row prefetch.
    asm volatile("" : "+v"(lane), "+v"(y));
#endif
    const int gx = L.gx0 + x;
    const bool deep = deep_y && gx >= 2 && gx <= L.lx - 3;   // wave-uniform
#ifdef M3_MANUAL
    // The operations issued since the requests that are consumed here (ids / previous-map id of row x+2 and the
    // populations of row x+1: at the top of the previous iteration; the reinit record: in its middle) are its nine
    // population stores and the stores of its bounce-back pass: at most nine operations outstanding <=> all of them landed.
    asm volatile("s_waitcnt vmcnt(9)" : "+v"(pend_c), "+v"(pend_o), "+v"(pend_old), "+v"(pend_ra), "+v"(pend_rb), "+v"(pend_rc) :: "memory");
    {
      const bool rok = x + 2 >= 0 && x + 2 < L.nxl;
      iE.c = (rok && c_in) ? pend_c : L.n;
      iE.outer = (rok && o_in) ? pend_o : L.n;
      oo2 = pend_old;
      gre = RRec{make_real2(m3_dbl(pend_ra.x, pend_ra.y), m3_dbl(pend_ra.z, pend_ra.w)),
                 make_real2(m3_dbl(pend_rb.x, pend_rb.y), m3_dbl(pend_rb.z, pend_rb.w)), m3_dbl(pend_rc.x, pend_rc.y)};
    }
#endif
    // the old owner's velocity at (x+1, y): frees the record's registers before anything else is requested
    const real re_ux = rr_ux(gre), re_uy = rr_uy(gre, x + 1);
    int onb[9];
    {
      const Ids3 b = iB.all(), c = iC.all(), d = iD.all();
      onb[0] = 0;
      onb[1] = b.p; onb[2] = b.c; onb[3] = b.m; onb[4] = c.m;
      onb[5] = d.m; onb[6] = d.c; onb[7] = d.p; onb[8] = c.p;
    }
    // ---- (0) the bounce-back links of row x, from the ids alone: link (P, q) <=> P fluid, S = P - e_q a grain node
    // (lattice-edge and off-lattice positions carry the id L.n). nnm / hzm as in classify_store_row.
    unsigned ibb = 0, hzm = 0;   // hzm: NN = P + e_q is not fluid and q <= 4 (only looked at when NN is interior)
    if (writer && x < xe && iC.c == -1) {
#pragma unroll
      for (int q = 1; q < 9; ++q) {
        const int oS = onb[OPPq(q)];
        if (oS != -1 && oS != L.n) {
          ibb |= 1u << q;
          if (q <= 4 && onb[q] != -1) hzm |= 1u << q;
        }
      }
    }
    // lane-major slots: the links of lane l occupy [t0, t0 + popc(ibb)) in ascending q
    const int cnt = __popc(ibb);
    const int incl = wave_inclusive_scan(cnt);
    const int t0 = incl - cnt;
    const int T = __builtin_amdgcn_readlane(incl, 63);
    auto write_desc = [&](int base, const int (&nb)[9]) {
      int t = t0 - base;
#pragma unroll
      for (int q = 1; q < 9; ++q) {
        if ((ibb >> q) & 1u) {
          if (t >= 0 && t < 64)
            desc[t] = lane | (q << 8) | (((hzm >> q) & 1u) << 13) | (nb[OPPq(q)] << 14);
          ++t;
        }
      }
    };
    const char* rec_src = reinterpret_cast<const char*>(G.pk);
    if (T > 0) {
      write_desc(0, onb);
      __builtin_amdgcn_wave_barrier();
      // the dense lane that will evaluate link `lane` fetches the record of the link's grain (lanes without a link
      // fetch record 0 into slots nobody reads: no divergent branch around the DMA)
      rec_src = reinterpret_cast<const char*>(G.pk) + (long)(lane < T ? (unsigned)desc[lane] >> 14 : 0u) * 64;
      lds_dma16(rec_src, lrec_lds);
      lds_dma16(rec_src + 16, lrec_lds + 1024);
      lds_dma16(rec_src + 32, lrec_lds + 2048);
      rec_src += 48;
      lds_dma16_tok(rec_src, lrec_lds + 3072);
    }
    // The DMA is invisible to hipcc's s_waitcnt bookkeeping (on purpose: with a DMA it knows about in flight it drains
    // the whole queue at the next use of any load). What orders it is this probe: an ordinary load whose ADDRESS
    // comes out of the DMA statement, so it is issued after the DMA; gfx9 retires vector-memory operations in order, so
    // once the compiler's own counted wait for `probe` is over, the records have landed.
    // (an address that comes out of an asm statement has lost its address space: say "global", or the load becomes a
    // FLAT one and the compiler falls back to vmcnt(0) everywhere)
#ifdef M3_MANUAL
    const int probe = 0;
    request_ids(x + 3);
#else
    typedef const int __attribute__((address_space(1))) * gint_ptr;
    const int probe = *(gint_ptr)(unsigned long long)rec_src;
    __builtin_amdgcn_sched_barrier(0);
    // ---- (1) 12 unconditional loads: ids of row x+3 (2), previous-map id of row x+3 (1), populations of row x+2 (9)
    const IdsRow inext = M3_LOAD_IDS(x + 3);
    const int oo3 = load_old(x + 3);
#endif
#ifdef M3_DMAPOP
    stage_read(pop_probe, Fp);
    pop_probe = dma_row(x + 2);
#else
#pragma unroll
    for (int q = 0; q < 9; ++q) Fp[q] = buf[q];
    load_raw(x + 1 + NBUF, buf);
#endif
    __builtin_amdgcn_sched_barrier(0);
    make_fstar(x + 1, Fp, oo1, re_ux, re_uy, iD.c);

    // the six cross-lane moves of a pull. DPP reads nothing from a lane that is switched off, so these run here, in
    // wave-uniform control flow, never inside a divergent branch.
    real In[9];
    In[0] = 0.0;
    In[2] = Fm[6];            // (-1, 0): same lane, row x-1, slot opp(2) = 6
    In[6] = Fp[2];            // ( 1, 0)
    In[1] = dpp_dn1(Fm[5]);   // (-1, 1): lane+1, row x-1, slot 5
    In[8] = dpp_dn1(F0[4]);   // ( 0, 1)
    In[7] = dpp_dn1(Fp[3]);   // ( 1, 1)
    In[3] = dpp_up1(Fm[7]);   // (-1,-1): lane-1
    In[4] = dpp_up1(F0[8]);   // ( 0,-1)
    In[5] = dpp_up1(Fp[1]);   // ( 1,-1)
    const long node = (long)x * L.sy + y;
    // ---- (2) the payloads of the bounce-back links to LDS (the first 64 links of the row; a row with more: see (5))
    auto write_pay = [&](int base, const real (&in)[9]) {
      int t = t0 - base;
#pragma unroll
      for (int q = 1; q < 9; ++q) {
        if ((ibb >> q) & 1u) {
          if (t >= 0 && t < 64) {
            // f*[P][opp q], f*[P][q], f*[P + e_q][opp q], f*[P - e_q][q]
            const int qo = OPPq(q);
            pay[t * 4 + 0] = F0[qo];
            pay[t * 4 + 1] = F0[q];
            pay[t * 4 + 2] = in[q];
            pay[t * 4 + 3] = in[qo];
          }
          ++t;
        }
      }
    };
    if (T > 0) write_pay(0, In);
    // ---- (3) the reinit record of row x+2: requested late, converted at the top of the next iteration
#ifdef M3_MANUAL
    request_reinit(oo2);
#else
    gre = reinit_rec(oo2);
#endif
    // ---- (4) everything but the bounce-back links: computed and stored
    const Ids3 a3 = iC.all(), b3 = iD.all(), c3 = iE.all();   // DPP: outside the divergent &&
    const bool actp = iD.c != -1 && node_active(L, G, a3, b3, c3, L.gx0 + x + 1, y, [&] { return grain_rec(iD.c); });
    {
      RegCtx3 C;
#pragma unroll
      for (int q = 0; q < 9; ++q) { C.Fo[q] = F0[q]; C.onb[q] = onb[q]; C.In[q] = In[q]; }
      C.o0 = iC.c;
      const int pack = (actm ? 1 : 0) | (act0 ? 2 : 0) | (actp ? 4 : 0);
      const int pk_up = dpp_dn1(pack);   // lane+1 (y+1)
      const int pk_dn = dpp_up1(pack);   // lane-1 (y-1)
      C.act = (((pk_up >> 0) & 1u) << 1) | (((pack >> 0) & 1u) << 2) | (((pk_dn >> 0) & 1u) << 3) |
              (((pk_dn >> 1) & 1u) << 4) | (((pk_dn >> 2) & 1u) << 5) | (((pack >> 2) & 1u) << 6) |
              (((pk_up >> 2) & 1u) << 7) | (((pk_up >> 1) & 1u) << 8);
      if (writer && x < xe) {
        if (deep) classify_store_all<false>(C, L, gx, y, fout, node);
        else classify_store_all<true>(C, L, gx, y, fout, node);
      }
    }
    // ---- (5) the bounce-back links, LAST: everything the evaluation needs is in LDS, and the registers of row x-1,
    // the incoming populations and the neighbour ids are dead by now (this pass is the register peak of the loop).
#ifndef M3_OVERFLOW
#define M3_OVERFLOW 1
#endif
    for (int base = 0; base < (M3_OVERFLOW ? T : (T < 64 ? T : 64)); base += 64) {   // wave-uniform; more than one round only if the row has > 64 links
      if (base == 0) {
#ifdef M3_MANUAL
        // since the record DMA: 3 id loads, 5 population DMAs, 3 record loads, 9 population stores
        asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
#else
        asm volatile("" ::"v"(probe) : "memory");   // the compiler waits for `probe` here => the records are in LDS
#endif
      } else {
        // rare: descriptors and payloads of the next 64 links, rebuilt from what is still in registers (rows x-1, x,
        // x+1 and the ids; the cross-lane moves are repeated, in wave-uniform control flow)
        const Ids3 b = iB.all(), c = iC.all(), d = iD.all();
        const int nb[9] = {0, b.p, b.c, b.m, c.m, d.m, d.c, d.p, c.p};
        const real in2[9] = {0.0, dpp_dn1(Fm[5]), Fm[6], dpp_up1(Fm[7]), dpp_up1(F0[8]), dpp_up1(Fp[1]), Fp[2],
                               dpp_dn1(Fp[3]), dpp_dn1(F0[4])};
        write_desc(base, nb);
        write_pay(base, in2);
      }
      __builtin_amdgcn_wave_barrier();
      if (base + lane < T) {
        const int d = desc[lane];
        const int src = d & 63;
        RtLink k;
        k.q = (d >> 8) & 15;
        k.gx = gx;
        k.gy = y - lane + src;
        k.own_qo = pay[lane * 4 + 0];
        k.own_q = pay[lane * 4 + 1];
        k.in_q = pay[lane * 4 + 2];
        k.in_qo = pay[lane * 4 + 3];
        const int ex = (k.q >= 1 && k.q <= 3) ? -1 : ((k.q >= 5 && k.q <= 7) ? 1 : 0);
        const int ey = (k.q == 1 || k.q >= 7) ? 1 : ((k.q >= 3 && k.q <= 5) ? -1 : 0);
        // is NN = P + e_q an interior node? (nn_interior<q, EDGE>)
        k.nn_int = deep || (k.gx + ex >= 1 && k.gx + ex <= L.lx - 2 && k.gy + ey >= 1 && k.gy + ey <= L.ly - 2);
        k.hazard = k.nn_int && ((d >> 13) & 1);
        GP g;
        if (base == 0) {
          const real2 ra = lrec[lane], rb = lrec[64 + lane], rc = lrec[128 + lane], re = lrec[192 + lane];
          g = GP{ra.x, ra.y, rb.x, rb.y, rc.x, rc.y, re.x, re.y};
        } else {
          g = load_gp(G, (int)((unsigned)d >> 14));
        }
        const real out = ibb_eval_rt(L, k, wc_diag, wc_axis, [&](int dx, int dy) {
          if (dx == -ex && dy == -ey) return g;
          // the hazard partner: the grain that owns NN = P + e_q (rare; its loads stay inside this branch)
          return load_gp(G, ob_new[(long)(x + dx) * L.sy + (k.gy + dy)]);
        });
        fout[fidx(k.q, node - lane + src)] = out;
        if (S.tab != nullptr) {
          const int rel = slot_line(k.gx - ex, k.gy - ey, ex, ey, g.xc, g.yc) + S.half;
          if ((unsigned)rel < (unsigned)S.spd)
            S.tab[((long)((unsigned)d >> 14) * 8 + (k.q - 1)) * S.spd + rel] = k.own_qo + out;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) { Fm[q] = F0[q]; F0[q] = Fp[q]; }
#ifdef M3_MANUAL
    iB = iC; iC = iD; iD = iE;   // (iE and oo2 are decoded at the top of the next iteration)
    oo1 = oo2;
#else
    iB = iC; iC = iD; iD = iE; iE = inext;
    oo1 = oo2; oo2 = oo3;
#endif
    actm = act0; act0 = actp;
  };
  if (NBUF == 1) {
    for (int x = xs; x < xe; ++x) iterate(x, bufA);
  } else {
    // ping-pong buffers, unrolled by two (no register copies); a row beyond the range stores nothing
    for (int x = xs; x < xe; x += 2) {
      iterate(x, bufA);
      iterate(x + 1, bufB);
    }
  }
}

#endif  // LBMDEM_AB

// ---------------------------------------------------------------------------------------------
// hydrodynamic force and torque (main.c:1285-1333)
// ---------------------------------------------------------------------------------------------

#undef M3_LOAD_IDS
__device__ __forceinline__ bool grain_box(const LatticeView& L, const GrainFluidView& G, int i, int& xi,
                                          int& xf, int& yi, int& yf) {
  const real xc = G.xc[i], yc = G.yc[i], rbl0 = G.rbl0[i];
  xi = (int)(xc - rbl0); if (xi < 1) xi = 1;                 // int max(real->int, 1): main.c:1300
  xf = (int)(xc + rbl0); if (xf > L.lx - 2) xf = L.lx - 2;   // main.c:1301
  yi = (int)(yc - rbl0); if (yi < 1) yi = 1;
  yf = (int)(yc + rbl0); if (yf > L.ly - 2) yf = L.ly - 2;
  return xi <= xf && yi <= yf;
}

// A grain is computed by the rank that owns the lattice column of its centre (first/last rank also
// take centres left/right of the lattice). On one GPU every grain is owned.
__device__ __forceinline__ bool grain_owned(const LatticeView& L, real xc) {
  const int lo = L.gx0 + L.xo0, hi = L.gx0 + L.xo1;  // owned global rows [lo, hi)
  const bool first = (lo == 0), last = (hi == L.lx);
  return (first || xc >= (real)lo) && (last || xc < (real)hi);
}

// Parity kernel: ONE WAVEFRONT PER GRAIN, bit-exact with the reference's serial x -> y -> q
// accumulation (main.c:1305-1321). The kernel is latency-bound (a grain touches ~0.5 KB of obst and
// ~2 KB of f scattered over ~20 rows), so it is organised to need only TWO dependent global round
// trips per grain and little LDS (many resident waves):
//  A0  the grain's footprint (bounding box + 1) of "obst == i" flags is staged in LDS: one round of
//      independent loads.
//  A1  lanes scan the bounding box in the reference's order (x outer, y inner), find boundary nodes
//      from the LDS flags and compact them (ballot prefix) into an LDS list with their link masks.
//  A2  one lane per boundary node: all populations of all its links are loaded in one round; every
//      link yields a term (fnx, fny, -fnx*(y-yc), fny*(x-xc)) -- the products do not depend on the
//      running sums -- stored in LDS in scan order (prefix sum of link counts).
//  B   serial, as it must be: three lanes replay the additions in that exact order, one lane per
//      accumulator chain: h1 += fnx; h2 += fny; h3 = (h3 - fnx*(y-yc)) + fny*(x-xc), all written as
//      h = (h + a) + b with b = +0.0 for the first two (x + (+0.0) is exact; x - p == x + (-p)).
constexpr int FORCE_TILE = 40;       // footprint edge staged in LDS (bounding box + 2); larger grains take the slow path
constexpr int FORCE_BN_CAP = 512;    // boundary nodes kept per grain before a flush
constexpr int FORCE_TERMS_CAP = 128; // terms replayed per batch

struct ForceLds {
  real sT[FORCE_TERMS_CAP * 4 + 2];             // [term][fnx, fny, -p1, p2], then a zero slot
  unsigned char sIn[FORCE_TILE * FORCE_TILE];     // 1 = node belongs to grain i
  unsigned short sBnK[FORCE_BN_CAP];              // boundary node: index in the bounding box
  unsigned char sBnM[FORCE_BN_CAP];               //                link mask (bit q-1)
};

// Phase B for up to 64 boundary nodes, one per lane (in scan order): `mask` = the node's links (bit q-1),
// fx/fy[q-1] = fnx, fny of link q, (wx, wy) = (x - xc, y - yc). Terms go to LDS in scan order, three lanes
// replay the additions. `h` is the running accumulator of lanes 0..2. One-wave workgroup: barriers are cheap.
__device__ __forceinline__ void force_replay(ForceLds& sh, int lane, unsigned mask, const real (&fx)[8],
                                             const real (&fy)[8], real wx, real wy, real& h) {
  const int zero_slot = FORCE_TERMS_CAP * 4;
  const int a_off = lane < 3 ? lane : 0;  // fnx | fny | -p1
  const int cnt = __popc(mask);
  int pos = cnt;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int v = __shfl_up(pos, off, 64);
    if (lane >= off) pos += v;
  }
  const int nterms = __shfl(pos, 63, 64);
  pos -= cnt;
  for (int lo = 0; lo < nterms; lo += FORCE_TERMS_CAP) {  // wave-uniform
    __syncthreads();  // the previous batch has been consumed
    int p = pos;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      if (mask & (1u << q)) {
        if (p >= lo && p < lo + FORCE_TERMS_CAP) {
          real* t = &sh.sT[(p - lo) * 4];
          t[0] = fx[q];
          t[1] = fy[q];
          t[2] = fx[q] * (-wy);
          t[3] = fy[q] * wx;
        }
        ++p;
      }
    }
    const int nb = nterms - lo < FORCE_TERMS_CAP ? nterms - lo : FORCE_TERMS_CAP;
    // pad the batch to a multiple of 8 with zero terms: (h + 0.0) + 0.0 == h exactly
    const int nb8 = (nb + 7) & ~7;
    if (lane < 4 * (nb8 - nb)) sh.sT[nb * 4 + lane] = 0.0;
    __syncthreads();
    if (lane < 3) {
      const int b_off = lane == 2 ? 3 : -1;
      for (int t = 0; t < nb8; t += 8) {
        real a[8], b[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {  // loads do not depend on h: issue them ahead of the chain
          a[u] = sh.sT[(t + u) * 4 + a_off];
          b[u] = sh.sT[b_off >= 0 ? (t + u) * 4 + b_off : zero_slot];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) h = (h + a[u]) + b[u];
      }
    }
  }
}

// The ordered sums of one grain gathered from the obstacle map and the lattice (phases A0, A1, A2, B above).
// Returns h (lanes 0..2).
__device__ __forceinline__ real force_gather(ForceLds& sh, const real* __restrict__ f,
                                               const int* __restrict__ obst, const LatticeView& L, int i, real xc,
                                               real yc, int xi, int xf, int yi, int yf, int lane) {
  real h = 0.0;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  const int nx = xf - xi + 1, ny = yf - yi + 1;
  const int total = nx * ny;
  const int tx = nx + 2, ty = ny + 2;
  const bool staged = tx <= FORCE_TILE && ty <= FORCE_TILE && total <= 65535;
  __syncthreads();
  if (staged) {  // A0
    for (int k = lane; k < tx * ty; k += 64) {
      const int x = xi - 1 + k / ty, y = yi - 1 + k % ty;  // in-bounds: the box is clamped to [1, l-2]
      sh.sIn[k] = obst[(long)(x - L.gx0) * L.sy + y] == i ? 1 : 0;
    }
    __syncthreads();
  }
  // boundary-node list is consumed whenever it fills up or the scan ends
  int nbn = 0;
  for (int base = 0; base < total || nbn > 0; base += 64) {
    // A1: classify 64 bounding-box nodes
    if (base < total) {
      const int k = base + lane;
      unsigned m = 0;
      if (k < total) {
        const int bx = k / ny, by = k % ny;
        if (staged) {
          const unsigned char* c = &sh.sIn[(bx + 1) * ty + (by + 1)];
          if (c[0]) {
#pragma unroll
            for (int q = 1; q < 9; ++q)
              if (!c[EXq(q) * ty + EYq(q)]) m |= 1u << (q - 1);
          }
        } else {
          const int x = xi + bx, y = yi + by;
          if (obst[(long)(x - L.gx0) * L.sy + y] == i) {
#pragma unroll
            for (int q = 1; q < 9; ++q)
              if (obst[(long)(x + EXq(q) - L.gx0) * L.sy + (y + EYq(q))] != i) m |= 1u << (q - 1);
          }
        }
      }
      const unsigned long long bal = __ballot(m != 0);
      if (m != 0) {
        const int slot = nbn + __popcll(bal & lt_mask);
        sh.sBnK[slot] = (unsigned short)k;
        sh.sBnM[slot] = (unsigned char)m;
      }
      nbn += __popcll(bal);
    }
    const bool last = base + 64 >= total;
    if (!(last || nbn > FORCE_BN_CAP - 64)) continue;  // wave-uniform: keep collecting
    __syncthreads();
    // A2 + B over the collected boundary nodes, 64 at a time
    for (int b0 = 0; b0 < nbn; b0 += 64) {
      const int bi = b0 + lane;
      unsigned mask = 0;
      real fx[8] = {0, 0, 0, 0, 0, 0, 0, 0}, fy[8] = {0, 0, 0, 0, 0, 0, 0, 0};
      real wx = 0.0, wy = 0.0;
      if (bi < nbn) {
        const int k = sh.sBnK[bi];
        mask = sh.sBnM[bi];
        const int x = xi + k / ny, y = yi + k % ny;
        const long nodeP = (long)(x - L.gx0) * L.sy + y;
        wx = x - xc;
        wy = y - yc;
#pragma unroll
        for (int q = 1; q < 9; ++q) {
          if (mask & (1u << (q - 1))) {
            const int qo = OPPq(q);
            const long nodeN = (long)(x + EXq(q) - L.gx0) * L.sy + (y + EYq(q));
            const real s = f[fidx(qo, nodeP)] + f[fidx(q, nodeN)];
            fx[q - 1] = s * EXq(qo);
            fy[q - 1] = s * EYq(qo);
          }
        }
      }
      force_replay(sh, lane, mask, fx, fy, wx, wy, h);
    }
    __syncthreads();
    nbn = 0;
    if (last) break;
  }
  return h;
}

__global__ __launch_bounds__(64) void k_forces_parity(const real* __restrict__ f,
                                                      const int* __restrict__ obst, LatticeView L,
                                                      GrainFluidView G, double scale12, double scale3,
                                                      real* __restrict__ fhf,
                                                      unsigned char* __restrict__ owner) {
  __shared__ ForceLds sh;
  const int lane = threadIdx.x;
  const int i = blockIdx.x;
  const real xc = G.xc[i], yc = G.yc[i];
  const bool own = grain_owned(L, xc);
  real h = 0.0;  // lanes 0,1,2 hold h1,h2,h3
  if (lane == 0) sh.sT[FORCE_TERMS_CAP * 4] = 0.0;
  int xi, xf, yi, yf;
  if (own && grain_box(L, G, i, xi, xf, yi, yf)) h = force_gather(sh, f, obst, L, i, xc, yc, xi, xf, yi, yf, lane);
  if (lane == 0 && owner) owner[i] = own ? 1 : 0;
  if (lane < 3) fhf[lane * L.n + i] = own ? h * (lane == 2 ? scale3 : scale12) : 0.0;
}

// The same sums from the ForceSlots table the fused kernel filled while it evaluated the bounce-back links, without
// touching the lattice. One wavefront serves GW grains.
//
// Phase A, per grain, all lanes. For a grain whose disc overlaps no other disc (rasteriser flag) and is not cut by
// the lattice-edge clamp, the grain's nodes are exactly those passing the paint test, so every lattice line
// parallel to a direction e that meets the disc carries exactly two boundary links: from its last in-disc node
// forwards (direction q(e)) and from its first in-disc node backwards (the opposite direction). One lane takes
// one line: the chord ends come from the line/circle intersection (one square root); chords whose ends could be a
// rounding error away from a lattice node (or that are nearly tangent) send the grain to the gather queue, for all
// others the chord IS the painted run of nodes (error bound in the code). A slot must hold a sum iff that link exists -- an empty slot where the geometry has a
// link means the link ends in a non-fluid node (another grain, a lattice-edge wall) or belongs to another
// rank's rows: the grain is then GATHERED from obst and f like in k_forces_parity. The links are ranked in the
// reference's scan order (x outer, y, q; main.c:1305-1309) through an LDS bitmap over (node, q), and the
// addends of the three accumulators are stored in that order. Addends that are exact zeros by construction
// (fnx of a vertical link, ...) are left out: h + (+-0.0) == h for the accumulators, which are never -0.0.
// Phase B: lanes 3g, 3g+1, 3g+2 replay the additions of grain g's three accumulators (the serial part is
// shared by the GW grains of the wave). The wave also resets the tables to empty for the next step.
// PASSES = passes over the four line families: 2 when two families fit a wave (spd <= 32), else 4.

// What k_forces_table does with the grains it is given
enum : int {
  FT_CONSUME = 0,  // grains this rank owns: complete the table, replay the sums, write fhf
  FT_PACK = 1,     // listed grains owned by a neighbour rank: complete this rank's part of the table and write
                   // {id, slots} to the message buffer (strip decomposition; the owner merges it into its own)
  FT_FAST = 2      // as FT_CONSUME, but the addends are summed by a cross-lane reduction instead of being replayed in
                   // the reference's order (force mode 1: same terms, last-bit differences)
};

constexpr unsigned long long M_DIAG = 0x5555555555555555ull;  // bits of q = 1, 3, 5, 7 in a bitmap word
constexpr unsigned long long M_XDIR = 0x2222222222222222ull;  // q = 2, 6 (ey = 0)
constexpr unsigned long long M_YDIR = 0x8888888888888888ull;  // q = 4, 8 (ex = 0)

constexpr int FT_WAVES = 4;  // waves per workgroup of k_forces_table

// FT_PACK takes both neighbours in one launch: blockIdx.y = side (0 low, 1 high); a null buffer skips the side
struct PackSides { const int* list[2]; const int* count[2]; real* buf[2]; };

template <int GW, int PASSES>
__global__ __launch_bounds__(64 * FT_WAVES) void k_forces_table(const real* __restrict__ f, const int* __restrict__ obst,
                                                                LatticeView L, GrainFluidView G, ForceSlots S, int cap1,
                                                                int cap3, int nw64, double scale12, double scale3,
                                                                real* __restrict__ fhf,
                                                                unsigned char* __restrict__ owner, int mode,
                                                                const int* __restrict__ list,
                                                                const int* __restrict__ list_count,
                                                                PackSides sides,
                                                                const unsigned char* __restrict__ mask, int list_cap) {
  extern __shared__ real sDyn[];
  real* __restrict__ packbuf = nullptr;
  if (mode == FT_PACK) {
    packbuf = sides.buf[blockIdx.y];
    if (!packbuf) return;
    list = sides.list[blockIdx.y];
    list_count = sides.count[blockIdx.y];
  }
  // per grain of the workgroup: addends of fhf1 [cap1] | fhf2 [cap1] | fhf3 [cap3]
  // then per wave: the bitmap [nw64] and the three per-word prefix counts [3][nw64]; then [FT_WAVES * GW][4] counts
  const int per_grain = 2 * cap1 + cap3;
  const int wave = threadIdx.x >> 6;
  const size_t lists_doubles = (size_t)FT_WAVES * GW * per_grain;
  unsigned long long* const bm = reinterpret_cast<unsigned long long*>(sDyn + lists_doubles) + (size_t)wave * nw64 * 3;
  int* const pw = reinterpret_cast<int*>(bm + nw64);  // 3 * nw64 ints = 1.5 * nw64 words: the wave's 3 * nw64 words hold both
  int* const counts = reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(sDyn + lists_doubles) +
                                             (size_t)FT_WAVES * nw64 * 3);
  const int lane = threadIdx.x & 63;
  const int gslot0 = wave * GW;                       // this wave's grains within the workgroup
  const int g0 = (blockIdx.x * FT_WAVES + wave) * GW; // position in the list (or the grain index itself without a list)
  const int ntodo = list ? (*list_count < list_cap ? *list_count : list_cap) : L.n;   // (an overflowing list is flagged by its producer)
  auto grain_at = [&](int pos) { return list ? list[pos] : pos; };
  const int own_lo = L.gx0 + L.xo0, own_hi = L.gx0 + L.xo1;  // rows whose links this rank produces: [own_lo, own_hi)
  const bool consume = mode != FT_PACK;
  const int spd = S.spd, HB = S.hb;
  const int B = 2 * HB + 1;
  constexpr int FPP = 4 / PASSES;   // line families per pass
  constexpr int LPF = 64 / FPP;     // lanes per family
  const int rel = lane & (LPF - 1);
  const bool lane_has_line = rel < spd;
  const int c = rel - S.half;       // slot_line() of this lane's line for the family's forward direction

  // line families: direction e, its q, the opposite q
  auto fam_ex = [](int fm) { return fm == 1 ? 0 : 1; };                  // (1,0) (0,1) (1,1) (1,-1)
  auto fam_ey = [](int fm) { return fm == 0 ? 0 : (fm == 3 ? -1 : 1); };
  auto fam_q = [](int fm) { return fm == 0 ? 6 : (fm == 1 ? 8 : (fm == 2 ? 7 : 5)); };

  unsigned long long fw[GW][PASSES], bw[GW][PASSES];  // slots of the forward / backward link of the lane's lines
  int n1[GW], n2[GW], n3[GW];                           // addends per accumulator; n1 < 0: gather
  int gid[GW];                                          // grain index, -1: nothing to do
  bool own[GW];
  // ---- tables of all GW grains: fetch, reset (one round of global latency for the whole wave)
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    gid[g] = g0 + g < ntodo ? grain_at(g0 + g) : -1;
    const int i = gid[g];
    bool local = false;
    if (i >= 0 && mask) {
      local = mask[i] != 0;      // strip decomposition: the grains the rasteriser saw (the others' geometry is stale)
      if (!local) gid[g] = -1;
    } else if (i >= 0) {
      const real xc = G.xc[i], rbl0 = G.rbl0[i];
      local = xc + rbl0 + 2.0 >= (real)L.gx0 && xc - rbl0 - 2.0 <= (real)(L.gx0 + L.nxl);
    }
    unsigned long long* tg = reinterpret_cast<unsigned long long*>(S.tab) + (long)(i >= 0 ? i : 0) * 8 * spd;
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int fm = j * FPP + lane / LPF;
      const int q = fam_q(fm), qo = q - 4;
      fw[g][j] = bw[g][j] = LBMDEM_SLOT_EMPTY;
      if (local && lane_has_line) {
        const int kf = (q - 1) * spd + rel;
        fw[g][j] = tg[kf];
        tg[kf] = LBMDEM_SLOT_EMPTY;
        const int rb = S.half - c;  // the same line seen from the opposite direction
        if (rb >= 0 && rb < spd) {
          const int kb = (qo - 1) * spd + rb;
          bw[g][j] = tg[kb];
          tg[kb] = LBMDEM_SLOT_EMPTY;
        }
      }
    }
  }
  // ---- phase A
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    const int i = gid[g];
    n1[g] = n2[g] = n3[g] = 0;
    own[g] = false;
    if (i < 0) continue;
    const real xc = G.xc[i], yc = G.yc[i], r2 = G.r2[i];
    own[g] = grain_owned(L, xc);
    const bool was_touched = S.touched[i] != 0;
    if (was_touched && lane == 0 && consume) S.touched[i] = 0;  // the rasteriser sets it again while it applies
    int xi, xf, yi, yf;
    const bool todo = (consume ? own[g] : !own[g]) && grain_box(L, G, i, xi, xf, yi, yf);
    if (!todo) continue;
    // overlapping discs: a lattice line may then carry several links of one direction -- not a table case
    if (was_touched) {
      n1[g] = -1;
      if (mode == FT_PACK && lane == 0) packbuf[1 + (long)(g0 + g) * (1 + 8 * spd)] = -1.0;
      continue;
    }
    const int X0 = (int)xc, Y0 = (int)yc;
    for (int w = lane; w < nw64; w += 64) bm[w] = 0ull;
    __builtin_amdgcn_wave_barrier();
    bool bad = false;
    real h1 = 0.0, h2 = 0.0, h3 = 0.0;  // FT_FAST: this lane's part of the three sums
    int keyf[PASSES], keyb[PASSES];  // -1: no link; else ((bx * B + by) * 8 + q - 1) | bx << 20 | by << 26
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int fm = j * FPP + lane / LPF;
      const int ex = fam_ex(fm), ey = fam_ey(fm), q = fam_q(fm);
      // a node of the line slot_line() == c, and the line's nodes: (px0 + m ex, py0 + m ey)
      const int px0 = X0 + ey * c, py0 = Y0 - (ey == 0 ? ex * c : 0);
      const real ax = px0 - xc, ay = py0 - yc;
      const real inv_ee = (ex != 0 && ey != 0) ? 0.5 : 1.0;
      const real ee = (ex != 0 && ey != 0) ? 2.0 : 1.0;
      const real be = ax * ex + ay * ey;
      real disc = be * be - ee * (ax * ax + ay * ay - r2);
      const bool far = disc < -4.0;
      // The rasteriser's test (x - xc)^2 + (y - yc)^2 <= r2 and this chord can only disagree about a node whose d2 is
      // within rounding (~1e-13) of r2, i.e. a node within 1e-13 / (2 sqrt(disc)) of a chord end: with disc >= 1e-6
      // that is < 1e-9 of an end (`shaky` below), and lines with |disc| < 1e-6 (all but tangent) are left to the gather
      // path altogether. So no node has to be tested against the disc here.
      const bool tangent = disc > -1e-6 && disc < 1e-6;
      const bool cuts = disc >= 1e-6;
      if (disc < 0.0) disc = 0.0;
      const real sq = sqrt(disc);
      const real mf = (sq - be) * inv_ee, mb = (-sq - be) * inv_ee;  // chord ends, in steps of e
      const real ff = floor(mf), cb = ceil(mb);
      // a chord end within rounding distance of a lattice node: let the gather path decide
      // (mf - ff and cb - mb lie in [0, 1): near 0 or near 1 <=> far from 1/2)
      const bool shaky = fabs((mf - ff) - 0.5) > 0.5 - 1e-9 || fabs((cb - mb) - 0.5) > 0.5 - 1e-9;
      int kf = (int)ff, kb = (int)cb;
      // the part of the line inside the paint box (the lattice-interior clamp cuts discs that reach a wall)
      {
        int lo = -(1 << 20), hi = 1 << 20;
        if (ex != 0) { lo = max(lo, xi - px0); hi = min(hi, xf - px0); }           // ex = +1 in every family
        else if (px0 < xi || px0 > xf) hi = lo - 1;
        if (ey > 0) { lo = max(lo, yi - py0); hi = min(hi, yf - py0); }
        else if (ey < 0) { lo = max(lo, py0 - yf); hi = min(hi, py0 - yi); }
        else if (py0 < yi || py0 > yf) hi = lo - 1;
        kf = min(kf, hi);
        kb = max(kb, lo);
      }
      const bool has = cuts && kb <= kf;      // the line carries a chord of in-disc nodes (inside the paint box)
      const bool meets = has && !far && lane_has_line;
      bool ffill = fw[g][j] != LBMDEM_SLOT_EMPTY, bfill = bw[g][j] != LBMDEM_SLOT_EMPTY;
      if (lane_has_line && !far && (tangent || (cuts && shaky))) bad = true;
      if (!meets && (ffill || bfill)) bad = true;   // a sum where the geometry has no link
      // A link of the geometry without a sum ends in a non-fluid node (a lattice-edge wall, another grain): the
      // fused kernel does not log those. If its far end lies in this rank's rows, gather the two populations
      // (main.c:1313-1316); otherwise it is the neighbour rank's to complete.
      if (meets && (!ffill || !bfill)) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
          if (side == 0 ? ffill : bfill) continue;
          const int km = side == 0 ? kf : kb, sg = side == 0 ? 1 : -1;
          const int px = px0 + km * ex, py = py0 + km * ey, nx = px + sg * ex, ny = py + sg * ey;
          if (nx < own_lo || nx >= own_hi) continue;
          const int ql = side == 0 ? q : q - 4, qlo = side == 0 ? q - 4 : q;
          const long nodeP = (long)(px - L.gx0) * L.sy + py, nodeN = (long)(nx - L.gx0) * L.sy + ny;
          if (obst[nodeP] != i || obst[nodeN] == -1) { bad = true; continue; }  // the table should have had it
          const real sum = f[fidx(qlo, nodeP)] + f[fidx(ql, nodeN)];
          if (side == 0) { fw[g][j] = (unsigned long long)__double_as_longlong(sum); ffill = true; }
          else { bw[g][j] = (unsigned long long)__double_as_longlong(sum); bfill = true; }
        }
        if (consume && (!ffill || !bfill)) bad = true;  // the owner must end up with every sum
      }
      keyf[j] = keyb[j] = -1;
      if (meets && mode == FT_FAST) {
        // forward link: direction q out of node kf, momentum along the opposite direction (main.c:1315-1318)
        const real sf = __longlong_as_double((long long)fw[g][j]), sb = __longlong_as_double((long long)bw[g][j]);
        const real wxf = px0 + kf * ex - xc, wyf = py0 + kf * ey - yc;
        const real wxb = px0 + kb * ex - xc, wyb = py0 + kb * ey - yc;
        const real fxf = sf * -ex, fyf = sf * -ey, fxb = sb * ex, fyb = sb * ey;
        h1 = h1 + fxf + fxb;
        h2 = h2 + fyf + fyb;
        h3 = h3 - fxf * wyf + fyf * wxf - fxb * wyb + fyb * wxb;
      }
      if (meets && mode == FT_CONSUME) {
        // in-disc nodes lie within +-hb of the truncated centre (hb >= largest reduced radius + 1)
        const int bxf = px0 + kf * ex - (X0 - HB), byf = py0 + kf * ey - (Y0 - HB);
        const int bxb = px0 + kb * ex - (X0 - HB), byb = py0 + kb * ey - (Y0 - HB);
        if ((unsigned)bxf >= (unsigned)B || (unsigned)byf >= (unsigned)B || (unsigned)bxb >= (unsigned)B ||
            (unsigned)byb >= (unsigned)B) bad = true;
        else {
          const int kyf = (bxf * B + byf) * 8 + (q - 1);
          const int kyb = (bxb * B + byb) * 8 + (q - 5);   // the opposite direction
          atomicOr(&bm[kyf >> 6], 1ull << (kyf & 63));
          atomicOr(&bm[kyb >> 6], 1ull << (kyb & 63));
          keyf[j] = kyf | (bxf << 20) | (byf << 26);        // B <= 63, B * B * 8 < 2^20
          keyb[j] = kyb | (bxb << 20) | (byb << 26);
        }
      }
    }
    if (__any(bad)) {
      n1[g] = -1;
      if (mode == FT_PACK && lane == 0)   // nothing usable for this grain: the entry must not keep an older period's data
        packbuf[1 + (long)(g0 + g) * (1 + 8 * spd)] = -1.0;
      continue;
    }
    if (mode == FT_FAST) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        h1 += __shfl_xor(h1, off, 64);
        h2 += __shfl_xor(h2, off, 64);
        h3 += __shfl_xor(h3, off, 64);
      }
      if (lane == 0) {
        fhf[i] = h1 * scale12;
        fhf[L.n + i] = h2 * scale12;
        fhf[2 * L.n + i] = h3 * scale3;
      }
      n1[g] = -2;  // written; nothing to replay, nothing to queue
      continue;
    }
    if (mode == FT_PACK) {  // {id, slots} to the message; the owner merges
      const int nslot = 8 * spd;
      unsigned long long* e = reinterpret_cast<unsigned long long*>(packbuf) + 1 + (long)(g0 + g) * (1 + nslot);
      if (lane == 0) reinterpret_cast<real*>(e)[0] = (real)i;
      for (int k = lane; k < nslot; k += 64) e[1 + k] = LBMDEM_SLOT_EMPTY;
      __builtin_amdgcn_wave_barrier();  // one wave's stores to the same address keep their order
      __threadfence_block();
#pragma unroll
      for (int j = 0; j < PASSES; ++j) {
        const int fm = j * FPP + lane / LPF;
        const int q = fam_q(fm), rb = S.half - c;
        if (lane_has_line) {
          e[1 + (q - 1) * spd + rel] = fw[g][j];
          if (rb >= 0 && rb < spd) e[1 + (q - 5) * spd + rb] = bw[g][j];
        }
      }
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    // per-word exclusive prefixes of the three addend counts
    {
      const int wpl = (nw64 + 63) >> 6;  // words per lane, contiguous
      int cd = 0, cx = 0, cy = 0;
      for (int u = 0; u < wpl; ++u) {
        const int w = lane * wpl + u;
        if (w < nw64) {
          const unsigned long long v = bm[w];
          cd += __popcll(v & M_DIAG); cx += __popcll(v & M_XDIR); cy += __popcll(v & M_YDIR);
        }
      }
      // one scan for the three counts: 10 bits each (a grain has < 1024 addends per accumulator: cap3 check below)
      const int mine3 = (cd + cx) | ((cd + cy) << 10) | ((2 * cd + cx + cy) << 20);
      const int incl = wave_inclusive_scan(mine3);
      const int excl = incl - mine3;
      int r1 = excl & 1023, r2_ = (excl >> 10) & 1023, r3 = (excl >> 20) & 1023;
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      for (int u = 0; u < wpl; ++u) {
        const int w = lane * wpl + u;
        if (w < nw64) {
          const unsigned long long v = bm[w];
          pw[w] = r1; pw[nw64 + w] = r2_; pw[2 * nw64 + w] = r3;
          const int d = __popcll(v & M_DIAG), x = __popcll(v & M_XDIR), y = __popcll(v & M_YDIR);
          r1 += d + x; r2_ += d + y; r3 += 2 * d + x + y;
        }
      }
      n1[g] = tot & 1023; n2[g] = (tot >> 10) & 1023; n3[g] = (tot >> 20) & 1023;
    }
    if (n1[g] > cap1 || n2[g] > cap1 || n3[g] > cap3) { n1[g] = -1; continue; }
    __builtin_amdgcn_wave_barrier();
    real* const l1 = sDyn + (size_t)(gslot0 + g) * per_grain;
    real* const l2 = l1 + cap1;
    real* const l3 = l2 + cap1;
    // (exo, eyo): the direction opposite to the link's, along which its momentum is exchanged
    auto emit = [&](int keyp, unsigned long long slot, int exo, int eyo) {
      const int key = keyp & 0xFFFFF;
      const int x = X0 - HB + ((keyp >> 20) & 63), y = Y0 - HB + ((keyp >> 26) & 63);
      const int w = key >> 6;
      const unsigned long long below = bm[w] & ((1ull << (key & 63)) - 1ull);
      const int bd = __popcll(below & M_DIAG), bx = __popcll(below & M_XDIR), by = __popcll(below & M_YDIR);
      const int r1 = pw[w] + bd + bx, r2_ = pw[nw64 + w] + bd + by, r3 = pw[2 * nw64 + w] + 2 * bd + bx + by;
      const real sum = __longlong_as_double((long long)slot);
      const real fnx = sum * exo, fny = sum * eyo;  // main.c:1315-1316
      const real wx = x - xc, wy = y - yc;
      int r3b = r3;
      if (exo != 0) { l1[r1] = fnx; l3[r3b++] = fnx * (-wy); }   // - fnx * (y - yc)
      if (eyo != 0) { l2[r2_] = fny; l3[r3b] = fny * wx; }        // + fny * (x - xc)
    };
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      if (keyf[j] >= 0) {
        const int fm = j * FPP + lane / LPF;
        const int ex = fam_ex(fm), ey = fam_ey(fm);
        emit(keyf[j], fw[g][j], -ex, -ey);   // forward link, direction e: opposite -e
        emit(keyb[j], bw[g][j], ex, ey);     // backward link, direction -e
      }
    }
    // pad every list to a multiple of 8 addends with zeros
    if (lane < 8) {
      if (n1[g] + lane < ((n1[g] + 7) & ~7)) l1[n1[g] + lane] = 0.0;
      if (n2[g] + lane < ((n2[g] + 7) & ~7)) l2[n2[g] + lane] = 0.0;
      if (n3[g] + lane < ((n3[g] + 7) & ~7)) l3[n3[g] + lane] = 0.0;
    }
  }
  // hand the wave's counts to the replaying wave; queue what the table could not serve
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    const int i = gid[g];
    if (lane == 0) {
      int* cnt = counts + (gslot0 + g) * 5;
      cnt[0] = n1[g]; cnt[1] = n2[g]; cnt[2] = n3[g]; cnt[3] = own[g] ? 1 : 0; cnt[4] = i;
      if (i >= 0 && consume) {
        if (owner) owner[i] = own[g] ? 1 : 0;
        if (n1[g] == -1) S.queue[atomicAdd(S.gathered, 1)] = i;
      }
      // a neighbour's grain whose table this rank cannot complete (overlapping discs across a strip cut)
      if (i >= 0 && mode == FT_PACK && n1[g] < 0) atomicOr(S.error, 1);
    }
  }
  if (mode == FT_PACK) {
    if (blockIdx.x == 0 && threadIdx.x == 0) packbuf[0] = (real)ntodo;
    return;
  }
  __syncthreads();
  if (wave != 0) return;
  // ---- phase B, first wave only: lane 3g + a replays accumulator a of grain g of the workgroup (the serial part
  // is shared by FT_WAVES * GW grains)
  {
    constexpr int NG = FT_WAVES * GW;
    static_assert(3 * NG <= 64, "one lane per accumulator");
    const int g = lane / 3, a = lane - 3 * g;
    int gi = -1;
    int mine = 0, longest = 0;
    bool replayed = false, mine_own = false;
    if (g < NG) {
      const int* cnt = counts + g * 5;
      replayed = cnt[0] >= 0;
      mine_own = cnt[3] != 0;
      gi = cnt[4];
      if (replayed) mine = (cnt[a] + 7) & ~7;
    }
    longest = mine;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const int v = __shfl_xor(longest, off, 64);
      longest = v > longest ? v : longest;
    }
    const real* tl = sDyn + (size_t)(g < NG ? g : 0) * per_grain + (a == 0 ? 0 : (a == 1 ? cap1 : 2 * cap1));
    real h = 0.0;
    for (int t = 0; t < longest; t += 8) {  // wave-uniform trip count
      if (t < mine) {
        real2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const real2*>(tl + t)[u];  // issued ahead of the chain
#pragma unroll
        for (int u = 0; u < 4; ++u) { h = h + v[u].x; h = h + v[u].y; }
      }
    }
    if (g < NG && gi >= 0 && replayed) fhf[a * L.n + gi] = mine_own ? h * (a == 2 ? scale3 : scale12) : 0.0;
  }
}

// The queued grains, gathered from obst and f: one wavefront per grain, a fixed grid strides over the queue.
__global__ __launch_bounds__(64) void k_forces_gather_queue(const real* __restrict__ f, const int* __restrict__ obst,
                                                            LatticeView L, GrainFluidView G, ForceSlots S,
                                                            double scale12, double scale3, real* __restrict__ fhf) {
  __shared__ ForceLds sh;
  const int lane = threadIdx.x;
  const int count = *S.gathered;
  if (blockIdx.x == 0 && lane == 0) *S.gathered_next = 0;   // the next step's counter (nobody reads or adds to it now)
  if (lane == 0) sh.sT[FORCE_TERMS_CAP * 4] = 0.0;
  for (int k = blockIdx.x; k < count; k += gridDim.x) {  // wave-uniform
    const int i = S.queue[k];
    int xi, xf, yi, yf;
    grain_box(L, G, i, xi, xf, yi, yf);
    if (xi - 1 < L.gx0 || xf + 1 > L.gx0 + L.nxl - 1) {  // the footprint leaves this rank's rows: cannot gather it
      if (lane == 0) atomicOr(S.error, 2);
      continue;
    }
    const real h = force_gather(sh, f, obst, L, i, G.xc[i], G.yc[i], xi, xf, yi, yf, lane);
    if (lane < 3) fhf[lane * L.n + i] = h * (lane == 2 ? scale3 : scale12);
  }
}

// Fast kernel: one wavefront per grain, lanes take bounding-box nodes, cross-lane shuffle reduction.
// Same terms as the parity kernel, different summation tree (differs in the last bits).
__global__ void k_forces_fast(const real* __restrict__ f, const int* __restrict__ obst, LatticeView L,
                              GrainFluidView G, double scale12, double scale3,
                              real* __restrict__ fhf, unsigned char* __restrict__ owner) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= L.n) return;
  const real xc = G.xc[i], yc = G.yc[i];
  const bool own = grain_owned(L, xc);
  real h1 = 0, h2 = 0, h3 = 0;
  int xi, xf, yi, yf;
  if (own && grain_box(L, G, i, xi, xf, yi, yf)) {
    const int ny = yf - yi + 1;
    const int total = (xf - xi + 1) * ny;
    for (int k = lane; k < total; k += 64) {
      const int x = xi + k / ny, y = yi + k % ny;
      const long rowP = (long)(x - L.gx0) * L.sy;
      if (obst[rowP + y] != i) continue;
#pragma unroll
      for (int q = 1; q < 9; ++q) {
        const int ex = EXq(q), ey = EYq(q), qo = OPPq(q);
        const long nodeN = (long)(x + ex - L.gx0) * L.sy + (y + ey);
        if (obst[nodeN] == i) continue;
        const real s = f[fidx(qo, rowP + y)] + f[fidx(q, nodeN)];
        const real fnx = s * EXq(qo);
        const real fny = s * EYq(qo);
        h1 = h1 + fnx;
        h2 = h2 + fny;
        h3 = h3 - fnx * (y - yc) + fny * (x - xc);
      }
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    h1 += __shfl_down(h1, off, 64);
    h2 += __shfl_down(h2, off, 64);
    h3 += __shfl_down(h3, off, 64);
  }
  if (lane == 0) {
    if (owner) owner[i] = own ? 1 : 0;
    fhf[i] = own ? h1 * scale12 : 0.0;
    fhf[L.n + i] = own ? h2 * scale12 : 0.0;
    fhf[2 * L.n + i] = own ? h3 * scale3 : 0.0;
  }
}

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
#if LBMDEM_F_TILES
    const int q = (int)((k % (9 * LBMDEM_TILE_Y)) / LBMDEM_TILE_Y);  // f[tile][q][TILE_Y]
#else
    const int q = (int)(k / L.plane);
#endif
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

inline int grid_for(long total, int block = 256, int cap = 256 * 8) {
  long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

void launch_obst_fill(int* obst, const LatticeView& L, hipStream_t st) { launch_obst_fill_rows(obst, L, 0, L.nxl, st); }

void launch_obst_fill_rows(int* obst, const LatticeView& L, int row0, int row1, hipStream_t st) {
  if (row1 <= row0) return;
  hipLaunchKernelGGL(k_obst_fill, dim3(grid_for((long)(row1 - row0) * L.sy / 4)), dim3(256), 0, st, obst, L, row0, row1);
}

void launch_obst_paint(int* obst, const LatticeView& L, int n, const real* x1, const real* x2, const real* r,
                       const real* rLB, const real* v1, const real* v2, const real* v3, real* xc,
                       real* yc, real* r2, real* rbl0, real* pk, unsigned char* touched,
                       const unsigned char* mask, unsigned* mincov, unsigned epoch, const int* list,
                       const int* list_count, int list_cap, const int* voff, const int* vnbr, hipStream_t st) {
  const long threads = (long)(list ? list_cap : n) * 64;
  hipLaunchKernelGGL(k_obst_paint, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, obst, L, n, x1, x2, r,
                     rLB, v1, v2, v3, xc, yc, r2, rbl0, pk, touched, mask, mincov, epoch, list, list_count, list_cap, voff,
                     vnbr);
}

void launch_slots_clear(const ForceSlots& S, int n, hipStream_t st) {
  const long count = (long)n * 8 * S.spd;
  hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(count)), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(S.tab),
                     count, (unsigned long long)LBMDEM_SLOT_EMPTY);
}

template <int TX, int TY>
static void launch_cs(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                      const LatticeView& L, const GrainFluidView& G, int remap, hipStream_t st) {
  const int rows = L.xo1 - L.xo0;
  const int tiles_y = (L.ly + TY - 1) / TY, tiles_x = (rows + TX - 1) / TX;
  const int ntiles = tiles_y * tiles_x;
  const int grid = remap ? ((ntiles + 7) / 8) * 8 : ntiles;
  hipLaunchKernelGGL((k_collide_stream<TX, TY>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new,
                     L, G, tiles_y, ntiles, remap);
}

#ifdef LBMDEM_AB
// A/B builds only (make AB=1): LBMDEM_CS_VARIANT = kernel shape + 8 * xcd_remap
static int cs_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("LBMDEM_CS_VARIANT");
    v = e ? atoi(e) : 25;
  }
  return v;
}
#endif

// rows per wave of the marching kernel: 32 (the measured optimum on a full lattice) unless that leaves fewer than two
// rounds of resident waves; then the largest even count (the kernel works through its rows in pairs) that still gives
// two full rounds, as short as 8. Measured on 508 rows x 66 windows (the interior of a 512-row strip), fused kernel
// in us for 4 / 6 / 8 / 10 / 12 / 14 / 16 / 18 / 20 rows: 146 / 144 / 138 / 150 / 157 / 169 / 145 / 144 / 156 -- a
// little more than a whole number of rounds is good, a little less than the next one is bad.
static int march_segment_rows(int rows, int nstrips) {
  const int resident = 256 * 2 * 4;
  int nseg = (2 * resident + nstrips - 1) / nstrips;
  if (nseg < 1) nseg = 1;
  int seg_rows = rows / nseg;
#ifndef MARCH_COND2
  seg_rows &= ~1;
#endif
  if (seg_rows < 8) seg_rows = 8;
  if (seg_rows > 32) seg_rows = 32;
  return seg_rows;
}

// which marching kernel: 2 = k_cs_march (two waves per SIMD), 3 = k_cs_march3 (three)
#ifndef LBMDEM_MARCH_DEFAULT
#define LBMDEM_MARCH_DEFAULT 2
#endif
__attribute__((unused)) static int march_kernel() {
#ifdef LBMDEM_AB
  static const int v = getenv("LBMDEM_MARCH") ? atoi(getenv("LBMDEM_MARCH")) : LBMDEM_MARCH_DEFAULT;
  return v;
#else
  return LBMDEM_MARCH_DEFAULT;
#endif
}

template <int LX, int MINW, int WW = 62>
static void launch_march(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                         const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int remap,
                         hipStream_t st) {
  const int rows = L.xo1 - L.xo0;
  const int nstrips = (L.ly + WW - 1) / WW;
  int seg_rows = LX;
  if (LX == 0) {
    // run-time segment length for SHORT row ranges (a strip of a multi-GPU decomposition, the rows next to a cut):
    // 256 CUs x 2 workgroups x 4 waves are resident (VGPR- and LDS-limited) and a wave takes about as long for 32
    // rows as the whole lattice takes per round, so fewer rows than two full rounds are cut into shorter segments
    seg_rows = march_segment_rows(rows, nstrips);
#ifdef LBMDEM_AB
    static const int env_rows = getenv("LBMDEM_CS_ROWS") ? atoi(getenv("LBMDEM_CS_ROWS")) : 0;
    if (env_rows > 0) seg_rows = env_rows;
#endif
  }
  const int nseg = (rows + seg_rows - 1) / seg_rows;
  const int nwork = nstrips * nseg;
  int grid = (nwork + 3) / 4;
  if (remap) grid = ((grid + 7) / 8) * 8;
#ifdef LBMDEM_AB   // k_cs_march3 only exists in the experiment build
#ifdef M3_DMAPOP
  if constexpr (WW == 62) {
    if (march_kernel() == 3 || march_kernel() == 21) {
      if (march_kernel() == 3)
        hipLaunchKernelGGL((k_cs_march3<LX, 3, WW, 1>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G,
                           S, nstrips, nwork, remap, seg_rows, seg_rows);
      else
        hipLaunchKernelGGL((k_cs_march3<LX, 2, WW, 1>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G,
                           S, nstrips, nwork, remap, seg_rows, seg_rows);
      return;
    }
  }
#else
  switch (march_kernel()) {
    case 3:
      hipLaunchKernelGGL((k_cs_march3<LX, 3, WW, 1>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G,
                         S, nstrips, nwork, remap, seg_rows, seg_rows);
      return;
    case 21:
      hipLaunchKernelGGL((k_cs_march3<LX, 2, WW, 1>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G,
                         S, nstrips, nwork, remap, seg_rows, seg_rows);
      return;
    case 22:
      hipLaunchKernelGGL((k_cs_march3<LX, 2, WW, 2>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G,
                         S, nstrips, nwork, remap, seg_rows, seg_rows);
      return;
    default: break;
  }
#endif
#endif
  unsigned dyn_lds = 0;
#ifdef LBMDEM_AB   // occupancy experiment: extra (unused) dynamic LDS so that only ONE workgroup fits a CU
  static const int env_lds = getenv("LBMDEM_MARCH_DYNLDS") ? atoi(getenv("LBMDEM_MARCH_DYNLDS")) : 0;
  dyn_lds = (unsigned)env_lds;
#endif
  hipLaunchKernelGGL((k_cs_march<LX, MINW, WW>), dim3(grid), dim3(256), dyn_lds, st, fin, fout, obst_old, obst_new, L, G,
                     S, nstrips, nwork, remap, seg_rows, seg_rows);
}

// Two row ranges of equal width w <= 32 (the rows next to the two cuts of a strip) in ONE launch: two segments of w rows,
// the second `stride` rows after the first.
static void launch_march_two_ranges(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                                    const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int lo0, int w,
                                    int hi0, hipStream_t st) {
  constexpr int WW = 62;
  LatticeView Ls = L;
  Ls.xo0 = lo0; Ls.xo1 = hi0 + w;
  const int nstrips = (L.ly + WW - 1) / WW;
  const int nwork = nstrips * 2;
  const int grid = (nwork + 3) / 4;
#ifdef LBMDEM_AB
  if (march_kernel() == 3) {
    hipLaunchKernelGGL((k_cs_march3<0, 3, WW, 1>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, Ls, G, S,
                       nstrips, nwork, 0, w, hi0 - lo0);
    return;
  }
#endif
  hipLaunchKernelGGL((k_cs_march<0, 2, WW>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, Ls, G, S, nstrips,
                     nwork, 0, w, hi0 - lo0);
}

// The marching kernel assumes reductionR < 1 (always true in the reference); other configurations run the
// LDS-tile kernel, which does not fill the slot table.
bool collide_stream_fills_slots(const LatticeView& L) { return L.reduced_lt1 != 0; }

void launch_collide_stream(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                           const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, hipStream_t st) {
#ifdef LBMDEM_AB
  if (getenv("LBMDEM_CS_VARIANT")) {   // experiments with other kernel shapes; without the variable: the product's choice
    int v = cs_variant();
    const int remap = (v >> 3) & 1;
    if (!L.reduced_lt1 && (v & ~8) >= 16) v = 1;
    switch (v & ~8) {
      case 0: launch_cs<8, 64>(fin, fout, obst_old, obst_new, L, G, remap, st); return;
      case 1: launch_cs<4, 64>(fin, fout, obst_old, obst_new, L, G, remap, st); return;
      case 2: launch_cs<4, 128>(fin, fout, obst_old, obst_new, L, G, remap, st); return;
      case 16: launch_march<16, 2>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;
      case 19: launch_march<64, 2>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;
      case 20: launch_march<0, 2>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;  // one balanced round
      case 21: launch_march<32, 2, 56>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;  // 64-byte aligned stores
      default: launch_march<32, 2>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;
    }
  }
#endif
  if (L.reduced_lt1) {
    if (march_segment_rows(L.xo1 - L.xo0, (L.ly + 61) / 62) >= 32)
      launch_march<32, 2>(fin, fout, obst_old, obst_new, L, G, S, /*xcd remap*/ 1, st);
    else
      launch_march<0, 2>(fin, fout, obst_old, obst_new, L, G, S, /*xcd remap*/ 1, st);
  }
  else launch_cs<4, 64>(fin, fout, obst_old, obst_new, L, G, 0, st);
}

void launch_collide_stream_edges(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                                 const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int lo0, int lo1,
                                 int hi0, int hi1, hipStream_t st) {
  bool one_launch = lo1 > lo0 && hi1 > hi0 && lo1 - lo0 == hi1 - hi0 && lo1 - lo0 <= 32 && lo1 <= hi0 && L.reduced_lt1;
#ifdef LBMDEM_AB
  if (getenv("LBMDEM_CS_VARIANT")) one_launch = false;   // an experiment with another fused kernel
#endif
  if (one_launch) {
    launch_march_two_ranges(fin, fout, obst_old, obst_new, L, G, S, lo0, lo1 - lo0, hi0, st);
    return;
  }
  LatticeView Ls = L;
  if (lo1 > lo0) { Ls.xo0 = lo0; Ls.xo1 = lo1; launch_collide_stream(fin, fout, obst_old, obst_new, Ls, G, S, st); }
  if (hi1 > hi0) { Ls.xo0 = hi0; Ls.xo1 = hi1; launch_collide_stream(fin, fout, obst_old, obst_new, Ls, G, S, st); }
}

void launch_forces_parity(const real* f, const int* obst, const LatticeView& L,
                          const GrainFluidView& G, double scale12, double scale3, real* fhf,
                          unsigned char* owner, hipStream_t st) {
  hipLaunchKernelGGL(k_forces_parity, dim3(L.n), dim3(64), 0, st, f, obst, L, G, scale12, scale3, fhf, owner);
}

template <int GW, int PASSES>
static void launch_forces_table_t(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                                  const ForceSlots& S, double scale12, double scale3, real* fhf, unsigned char* owner,
                                  int fast, hipStream_t st) {
  const int HB = S.hb, B = 2 * HB + 1;
  const int nw64 = (B * B * 8 + 63) / 64;
  // addends per accumulator: one link per direction and lattice line meeting the disc (<= spd lines for the
  // diagonal families, <= 2 hb + 1 for the axis families); fhf3 takes two addends per diagonal link
  const int cap1 = (4 * S.spd + 2 * B + 7) & ~7, cap3 = (8 * S.spd + 4 * B + 7) & ~7;
  const size_t lists_doubles = (size_t)FT_WAVES * GW * (2 * cap1 + cap3);
  const size_t lds = lists_doubles * 8 + (size_t)FT_WAVES * nw64 * 24 + (size_t)FT_WAVES * GW * 20;
  const int per_block = FT_WAVES * GW;
  const int ntodo = S.local_list ? S.local_cap : L.n;   // strips: the compacted list of local grains bounds the launch
  hipLaunchKernelGGL((k_forces_table<GW, PASSES>), dim3((ntodo + per_block - 1) / per_block), dim3(64 * FT_WAVES), lds, st,
                     f, obst, L, G, S, cap1, cap3, nw64, scale12, scale3, fhf, owner, fast ? (int)FT_FAST : (int)FT_CONSUME, S.local_list,
                     S.local_count, PackSides{}, S.mask, S.local_cap);
  const int grid = L.n < 256 ? L.n : 256;
  hipLaunchKernelGGL(k_forces_gather_queue, dim3(grid), dim3(64), 0, st, f, obst, L, G, S, scale12, scale3, fhf);
}

void launch_forces_table_pack(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                              const ForceSlots& S, const int* const list[2], const int* const list_count[2], int cap,
                              real* const buf[2], hipStream_t st) {
  const int HB = S.hb, B = 2 * HB + 1;
  const int nw64 = (B * B * 8 + 63) / 64;
  const int cap1 = (4 * S.spd + 2 * B + 7) & ~7, cap3 = (8 * S.spd + 4 * B + 7) & ~7;
  const size_t lds = (size_t)FT_WAVES * (2 * cap1 + cap3) * 8 + (size_t)FT_WAVES * nw64 * 24 + (size_t)FT_WAVES * 20;
  const int blocks = (cap + FT_WAVES - 1) / FT_WAVES;   // the list lengths are only known on the device
  const PackSides P{{list[0], list[1]}, {list_count[0], list_count[1]}, {buf[0], buf[1]}};
  if (S.spd <= 32)
    hipLaunchKernelGGL((k_forces_table<1, 2>), dim3(blocks, 2), dim3(64 * FT_WAVES), lds, st, f, obst, L, G, S, cap1, cap3,
                       nw64, 0.0, 0.0, (real*)nullptr, (unsigned char*)nullptr, (int)FT_PACK, (const int*)nullptr,
                       (const int*)nullptr, P, S.mask, cap);
  else
    hipLaunchKernelGGL((k_forces_table<1, 4>), dim3(blocks, 2), dim3(64 * FT_WAVES), lds, st, f, obst, L, G, S, cap1, cap3,
                       nw64, 0.0, 0.0, (real*)nullptr, (unsigned char*)nullptr, (int)FT_PACK, (const int*)nullptr,
                       (const int*)nullptr, P, S.mask, cap);
}

void launch_forces_slots(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                         const ForceSlots& S, double scale12, double scale3, real* fhf, unsigned char* owner,
                         int fast, hipStream_t st) {
#ifdef LBMDEM_AB
  static const int gw = getenv("LBMDEM_FORCE_GW") ? atoi(getenv("LBMDEM_FORCE_GW")) : 1;
#else
  const int gw = 1;
#endif
  if (S.spd <= 32) {
    switch (gw) {
#ifdef LBMDEM_AB
      case 2: launch_forces_table_t<2, 2>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, st); break;
      case 3: launch_forces_table_t<3, 2>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, st); break;
      case 5: launch_forces_table_t<5, 2>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, st); break;
#endif
      default: launch_forces_table_t<1, 2>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, st); break;
    }
  } else {
    launch_forces_table_t<1, 4>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, st);
  }
}

void launch_forces_fast(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                        double scale12, double scale3, real* fhf, unsigned char* owner,
                        hipStream_t st) {
  const long threads = (long)L.n * 64;
  hipLaunchKernelGGL(k_forces_fast, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, f, obst, L,
                     G, scale12, scale3, fhf, owner);
}

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
