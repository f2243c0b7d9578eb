// lbm_device.h -- device code shared by the fluid translation units (lbm_fused.hip, lbm_fused_ab.hip, lbm_forces.hip,
// lbm_obst.hip, lbm_lattice.hip): the D2Q9 tables, the population layout, the MRT collision, the grain equilibrium and the
// interpolated bounce-back (main.c:966-986, 1071-1243), each formula with the reference's expression association
// (the library is compiled -ffp-contract=off). Everything sits in an unnamed namespace: every TU gets its own copy.
#pragma once

#include "lbmdem_internal.h"

#include <stdlib.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "these kernels are written for gfx950 (CDNA4, wave64): DPP wave shifts, in-order vector-memory retirement, store ordering within a wave"
#endif

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
// (LBMDEM_TILE_Y = 16 nodes for double, 32 for float: one 128-byte line per tile and direction either way)
__device__ __forceinline__ long fbase(long node) {
  return (node / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (node % LBMDEM_TILE_Y);
}
// the same from the local row and the column (sy is a multiple of the tile)
__device__ __forceinline__ long fbase_xy(const LatticeView& L, int xl, int y) {
  return (long)xl * L.sy * 9 + (y / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (y % LBMDEM_TILE_Y);
}
#define F_QSTRIDE(L) ((long)LBMDEM_TILE_Y)
#define fidx(q, node) (fbase(node) + (q) * F_QSTRIDE(L))   // needs the LatticeView `L` in scope

// the fluid-side record of one grain
struct GP { real x1, x2, v1, v2, v3, xc, yc, r2; };

__device__ __forceinline__ GP load_gp(const GrainFluidView& G, int i) {
  const real2* p = reinterpret_cast<const real2*>(G.pk + (long)i * 8);
  const real2 a = p[0], b = p[1], c = p[2], d = p[3];
  return GP{a.x, a.y, b.x, b.y, c.x, c.y, d.x, d.y};
}

// what reinit_obst_density needs of it: position and velocities (main.c:974-975)
struct GPv { real x1, x2, v1, v2, v3; };
__device__ __forceinline__ GPv load_gpv(const GrainFluidView& G, int i) {
  const real2* p = reinterpret_cast<const real2*>(G.pk + (long)i * 8);
  const real2 a = p[0], b = p[1];
  return GPv{a.x, a.y, b.x, b.y, G.pk[(long)i * 8 + 4]};
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

  // three true divisions by the same rho (rounds 1-2 formed them from one reciprocal with exact corrections: the same bits,
  // fewer instructions, but guards and selects on the dependent chain -- 0-4.7 % slower)
  const real d1 = 3 * (j_x2 + j_y2) / rho, d2 = (j_x2 - j_y2) / rho, d3 = j_x * j_y / rho;
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
  const real u_squ = (ux * ux + uy * uy) / L.cc;   // L.cc = c * c in `real` arithmetic (host), main.c:976
  const real e1 = (-ux + uy) / L.c;     // q = 1: (-1, 1)
  const real e2 = (-ux) / L.c;          // q = 2: (-1, 0)
  const real e3 = (-ux + (-uy)) / L.c;  // q = 3: (-1,-1)
  const real e4 = (-uy) / L.c;          // q = 4: ( 0,-1)
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
__device__ __forceinline__ void grain_equilibrium(const LatticeView& L, const GPv& g, int x, int y, real (&f)[9]) {
  grain_equilibrium_u(L, g.v1 - (y * L.dx + L.Mby - g.x2) * g.v3, g.v2 + (x * L.dx + L.Mgx - g.x1) * g.v3, f);   // wall_ux, wall_uy
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

// g = the record of the grain that owns S = P - e_q; rec_nn() = that of the grain that owns NN = P + e_q (hazard links only)
template <class RecFn>
__device__ __forceinline__ real ibb_eval_rt(const LatticeView& L, const RtLink& k, real wc_diag,
                                              real wc_axis, const GP& g, RecFn rec_nn) {
  const int q = k.q;
  const int ex = (q >= 1 && q <= 3) ? -1 : ((q >= 5 && q <= 7) ? 1 : 0);
  const int ey = (q == 1 || q >= 7) ? 1 : ((q >= 3 && q <= 5) ? -1 : 0);
  const real wc = (q & 1) ? wc_diag : wc_axis;
  const int sx = k.gx - ex, sy = k.gy - ey;
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
      const GP gn = rec_nn();
      IbbLink b;
      b.d = link_delta_rt(nx, ny, -ex, -ey, gn.xc, gn.yc, gn.r2);
      b.uw = (-ex) * wall_ux(L, gn, ny) + (-ey) * wall_uy(L, gn, nx);
      if (b.d >= 0.5) f2 = ibb_far_rt(L, b, k.own_q, k.own_qo, wc);
      else if (b.d > 0. && b.d < 0.5) f2 = 2 * b.d * k.own_q + (1 - 2 * b.d) * k.in_qo + 6 * wc * b.uw;
    }
  }
  return 2 * a.d * k.own_qo + (1 - 2 * a.d) * f2 + 6 * wc * a.uw;
}

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
inline int grid_for(long total, int block = 256, int cap = 256 * 8) {
  long g = (total + block - 1) / block;
  if (g > cap) g = cap;
  if (g < 1) g = 1;
  return (int)g;
}

}  // namespace
