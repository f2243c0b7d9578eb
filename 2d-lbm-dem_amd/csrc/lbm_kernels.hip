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
//   obst_construction clearing a 9-double-per-node delta array).
//
// Bit parity with the reference's serial loops is a design constraint: expression association is
// kept, the file is compiled with -ffp-contract=off, and the one order-dependent read of the
// reference's in-place IBB loop (a node two links out that the x-outer/y-inner scan has already
// rewritten) is reproduced by recomputing that node's new value (see pull_one()).
//
// The equivalence swap+stream == pull, f_new[P][q] = f*[P - e_q][q] (or f*[P][opp q] when P - e_q is
// off the array), is derived in SURVEY.md "Notes" and verified by tests against the oracle.

#include "lbmdem_internal.h"

namespace {

// D2Q9 direction table (main.c:70-71) and weights (main.c:53-54)
__host__ __device__ constexpr int EXq(int q) {
  return (q == 1 || q == 2 || q == 3) ? -1 : ((q == 5 || q == 6 || q == 7) ? 1 : 0);
}
__host__ __device__ constexpr int EYq(int q) {
  return (q == 1 || q == 7 || q == 8) ? 1 : ((q == 3 || q == 4 || q == 5) ? -1 : 0);
}
__host__ __device__ constexpr int OPPq(int q) { return q == 0 ? 0 : (q <= 4 ? q + 4 : q - 4); }
__host__ __device__ constexpr double Wq(int q) {
  return q == 0 ? 4. / 9 : ((q & 1) ? 1. / 36 : 1. / 9);
}

// rigid-body velocity of grain i at global node (x, y): main.c:974-975,1172-1173
__device__ __forceinline__ double wall_ux(const LatticeView& L, const GrainFluidView& G, int i, int y) {
  return G.v1[i] - (y * L.dx + L.Mby - G.x2[i]) * G.v3[i];
}
__device__ __forceinline__ double wall_uy(const LatticeView& L, const GrainFluidView& G, int i, int x) {
  return G.v2[i] + (x * L.dx + L.Mgx - G.x1[i]) * G.v3[i];
}

// main.c:1082-1116, in registers
__device__ __forceinline__ void mrt_collide(const LatticeView& L, double (&f)[9]) {
  const double a = 1. / 36;
  const double f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3], f4 = f[4], f5 = f[5], f6 = f[6],
               f7 = f[7], f8 = f[8];
  const double rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
  const double e = -4 * f0 + 2 * f1 - f2 + 2 * f3 - f4 + 2 * f5 - f6 + 2 * f7 - f8;
  const double eps = 4 * f0 + f1 - 2 * f2 + f3 - 2 * f4 + f5 - 2 * f6 + f7 - 2 * f8;
  const double j_x = f5 + f6 + f7 - f1 - f2 - f3;
  const double q_x = -f1 + 2 * f2 - f3 + f5 - 2 * f6 + f7;
  const double j_y = f1 + f8 + f7 - f3 - f4 - f5;
  const double q_y = f1 - f3 + 2 * f4 - f5 + f7 - 2 * f8;
  const double p_xx = f2 - f4 + f6 - f8;
  const double p_xy = -f1 + f3 - f5 + f7;

  const double j_x2 = j_x * j_x;
  const double j_y2 = j_y * j_y;

  const double eO = e - L.s2 * (e + 2 * rho - 3 * (j_x2 + j_y2) / rho);
  const double epsO = eps - L.s3 * (eps - rho + 3 * (j_x2 + j_y2) / rho);
  const double q_xO = q_x - L.s5 * (q_x + j_x);
  const double q_yO = q_y - L.s7 * (q_y + j_y);
  const double p_xxO = p_xx - L.s8 * (p_xx - (j_x2 - j_y2) / rho);
  const double p_xyO = p_xy - L.s9 * (p_xy - j_x * j_y / rho);

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

// equilibrium at rho = 1 and the grain's rigid-body velocity: main.c:974-981
__device__ __forceinline__ void grain_equilibrium(const LatticeView& L, const GrainFluidView& G, int i,
                                                  int x, int y, double (&f)[9]) {
  const double ux = wall_ux(L, G, i, y), uy = wall_uy(L, G, i, x);
  const double u_squ = (ux * ux + uy * uy) / (L.c * L.c);
#pragma unroll
  for (int q = 0; q < 9; ++q) {
    const double eu = (EXq(q) * ux + EYq(q) * uy) / L.c;
    f[q] = Wq(q) * (1. + 3 * eu + 4.5 * eu * eu - 1.5 * u_squ);
  }
}

// wall distance along link q from solid node (x, y) of a disc (xc, yc, r2): main.c:1054-1058
template <int q>
__device__ __forceinline__ double link_delta(int x, int y, double xc, double yc, double r2) {
  constexpr int ex = EXq(q), ey = EYq(q);
  const double aa = (double)(ex < 0 ? -ex : ex) + (double)(ey < 0 ? -ey : ey);
  const double bb = (x + ex - xc) * ex + (y + ey - yc) * ey;
  const double cc = (x + ex - xc) * (x + ex - xc) + (y + ey - yc) * (y + ey - yc) - r2;
  return (bb - sqrt(fabs(bb * bb - aa * cc))) / aa;
}

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// obst = -1 in the interior, nbgrains on the four lattice edges (main.c:669-683, 997-999)
__global__ void k_obst_fill(int* __restrict__ obst, LatticeView L) {
  const long total = (long)L.nxl * L.sy;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int xl = (int)(k / L.sy), y = (int)(k % L.sy);
    const int gx = L.gx0 + xl;
    const bool edge = (gx == 0 || gx == L.lx - 1 || y == 0 || y >= L.ly - 1);
    obst[k] = edge ? L.n : -1;
  }
}

// per-grain lattice geometry (main.c:1009-1013)
__global__ void k_grain_geom(int n, const double* __restrict__ x1, const double* __restrict__ x2,
                             const double* __restrict__ r, const double* __restrict__ rLB, double Mgx,
                             double Mby, double dx, double* __restrict__ xc, double* __restrict__ yc,
                             double* __restrict__ r2, double* __restrict__ rbl0) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xc[i] = (x1[i] - Mgx) / dx;
  yc[i] = (x2[i] - Mby) / dx;
  r2[i] = rLB[i] * rLB[i];
  rbl0[i] = r[i] / dx;
}

// Rasterise the reduced discs (main.c:1016-1032). One wavefront per grain, lanes sweep the
// bounding box with y fastest (coalesced). Overlaps resolve to the highest grain index, which is
// what the reference's ascending serial paint produces (main.c:1028) -> atomicMax.
__global__ void k_obst_paint(int* __restrict__ obst, LatticeView L, GrainFluidView G) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= L.n) return;
  const double xc = G.xc[i], yc = G.yc[i], r2 = G.r2[i], rbl0 = G.rbl0[i];
  const double R2 = rbl0 * rbl0;
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
  const int total = (xf - xi + 1) * ny;
  for (int k = lane; k < total; k += 64) {
    const int x = xi + k / ny, y = yi + k % ny;
    const double d2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
    if (d2 <= R2 && d2 <= r2) atomicMax(&obst[(long)(x - L.gx0) * L.sy + y], i);
  }
}

// ---------------------------------------------------------------------------------------------
// the fused fluid kernel
// ---------------------------------------------------------------------------------------------

template <int TX, int TY>
struct Tile {
  static constexpr int RX = TX + 2, RY = TY + 2;  // staged populations: halo 1
  static constexpr int OX = TX + 4, OY = TY + 4;  // staged obstacle ids: halo 2 (act of halo-1 nodes)
  double* sF;  // [9][RX][RY]
  int* sO;     // [OX][OY]
  __device__ __forceinline__ double& F(int q, int tx, int ty) const {
    return sF[(q * RX + (tx + 1)) * RY + (ty + 1)];
  }
  __device__ __forceinline__ int O(int tx, int ty) const { return sO[(tx + 2) * OY + (ty + 2)]; }
  // A solid node is "active" when one of its 8 neighbours was fluid at the moment its owning grain
  // was painted (main.c:1039-1052). Grains are painted in ascending index, so besides the
  // neighbours that are fluid in the final map this also counts neighbours now covered by a
  // HIGHER-index grain that do not lie inside the owner's own disc (they were still fluid when the
  // owner was painted). Only reachable when reduced discs of different grains touch or overlap.
  // (A neighbour additionally covered by a third, lower-index disc is not detected: DESIGN.md.)
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
    const double xc = G.xc[oS], yc = G.yc[oS], r2 = G.r2[oS], rb = G.rbl0[oS];
    const double R2 = rb * rb;
#pragma unroll
    for (int q = 1; q < 9; ++q) {
      const int o = O(tx + EXq(q), ty + EYq(q));
      if (o > oS && o != L.n) {
        const int x = gx + EXq(q), y = gy + EYq(q);
        const double d2 = (x - xc) * (x - xc) + (y - yc) * (y - yc);
        if (!(d2 <= R2 && d2 <= r2)) return true;
      }
    }
    return false;
  }
};

// Interpolated bounce-back (Bouzidi, moving wall) at solid node S = (sx, sy) of grain i for link q
// towards the fluid node N = S + e_q: main.c:1166-1185 / 1198-1217.
struct IbbLink { double d, uw; };

template <int q>
__device__ __forceinline__ IbbLink ibb_link(const LatticeView& L, const GrainFluidView& G, int i, int sx,
                                            int sy) {
  constexpr int ex = EXq(q), ey = EYq(q);
  IbbLink k;
  k.d = link_delta<q>(sx, sy, G.xc[i], G.yc[i], G.r2[i]);
  k.uw = ex * wall_ux(L, G, i, sy) + ey * wall_uy(L, G, i, sx);
  return k;
}
__device__ __forceinline__ bool ibb_far(const IbbLink& k) { return k.d >= 0.5; }
__device__ __forceinline__ bool ibb_near(const IbbLink& k) { return k.d > 0. && k.d < 0.5; }
// delta >= 1/2: fN_opp = f*[N][opp q], fN_q = f*[N][q]
template <int q>
__device__ __forceinline__ double ibb_far_value(const LatticeView& L, const IbbLink& k, double fN_opp,
                                                double fN_q) {
  return fN_opp / (2 * k.d) + (2 * k.d - 1) * fN_q / (2 * k.d) + 3 * (Wq(q) / L.c) * k.uw / k.d;
}
// 0 < delta < 1/2: f2 = the population read two links out, f[N + e_q][opp q]
template <int q>
__device__ __forceinline__ double ibb_near_value(const LatticeView& L, const IbbLink& k, double fN_opp,
                                                 double f2) {
  return 2 * k.d * fN_opp + (1 - 2 * k.d) * f2 + 6 * (Wq(q) / L.c) * k.uw;
}

// f_new[P][q] for one direction. (px, py) tile coordinates of P, (gx, gy) global.
template <int q, int TX, int TY>
__device__ __forceinline__ double pull_one(const Tile<TX, TY>& T, const LatticeView& L,
                                           const GrainFluidView& G, int px, int py, int gx, int gy) {
  constexpr int ex = EXq(q), ey = EYq(q), qo = OPPq(q);
  const int sxg = gx - ex, syg = gy - ey;  // source node S = P - e_q
  if (sxg < 0 || sxg >= L.lx || syg < 0 || syg >= L.ly) return T.F(qo, px, py);  // array edge: main.c:1237
  const bool s_interior = sxg >= 1 && sxg <= L.lx - 2 && syg >= 1 && syg <= L.ly - 2;
  if (!s_interior) {
    // S is a lattice-edge wall node. Its slot q was overwritten by the edge copies
    // (main.c:1123-1145) with f*[P][opp q] when P is interior, and -- because the y-edge loop runs
    // before the x-edge loop -- also when S sits on a y edge and P on an x edge; otherwise it
    // still holds its old value.
    const bool s_yedge = (syg == 0 || syg == L.ly - 1) && sxg >= 1 && sxg <= L.lx - 2;
    const bool copied = gy >= 1 && gy <= L.ly - 2 && ((gx >= 1 && gx <= L.lx - 2) || s_yedge);
    return copied ? T.F(qo, px, py) : T.F(q, px - ex, py - ey);
  }
  const int oS = T.O(px - ex, py - ey);
  if (oS == -1) return T.F(q, px - ex, py - ey);  // plain streaming from a fluid node
  const int oP = T.O(px, py);
  if (oP != -1)  // solid -> non-fluid link: active solid nodes reset the slot to w (main.c:1161-1162)
    return T.active(L, G, px - ex, py - ey, sxg, syg) ? Wq(q) : T.F(q, px - ex, py - ey);

  // P fluid, S an (active) solid node of grain oS: interpolated bounce-back
  const IbbLink k = ibb_link<q>(L, G, oS, sxg, syg);
  if (ibb_far(k)) return ibb_far_value<q>(L, k, T.F(qo, px, py), T.F(q, px, py));
  if (!ibb_near(k)) return T.F(q, px - ex, py - ey);  // neither branch fires: slot keeps its value

  // 0 < delta < 1/2: the reference reads f[NN][opp q], NN = P + e_q, *in place* (main.c:1181,1213).
  double f2;
  const int nxg = gx + ex, nyg = gy + ey;
  const bool n_interior = nxg >= 1 && nxg <= L.lx - 2 && nyg >= 1 && nyg <= L.ly - 2;
  if (!n_interior) {
    f2 = T.F(q, px, py);  // edge wall node: its slot opp q was set by the edge copy to f*[P][q]
  } else {
    const int oN = T.O(px + ex, py + ey);
    f2 = T.F(qo, px + ex, py + ey);  // fluid: post-collision; solid: value before the IBB loop
    // NN is solid and precedes S in the reference's x-outer/y-inner scan (e_q lexicographically
    // negative, q = 1..4): S reads the value NN's own IBB update has already produced. That update
    // saw S's slot q in its pre-loop state (S comes later), so the chain ends here.
    if (oN != -1 && q <= 4) {
      const IbbLink kn = ibb_link<qo>(L, G, oN, nxg, nyg);
      if (ibb_far(kn)) f2 = ibb_far_value<qo>(L, kn, T.F(q, px, py), T.F(qo, px, py));
      else if (ibb_near(kn)) f2 = ibb_near_value<qo>(L, kn, T.F(q, px, py), T.F(q, px - ex, py - ey));
    }
  }
  return ibb_near_value<q>(L, k, T.F(qo, px, py), f2);
}

template <int TX, int TY>
__global__ __launch_bounds__(256) void k_collide_stream(const double* __restrict__ fin,
                                                        double* __restrict__ fout,
                                                        const int* __restrict__ ob_old,
                                                        const int* __restrict__ ob_new, LatticeView L,
                                                        GrainFluidView G) {
  using TT = Tile<TX, TY>;
  __shared__ double sF[9 * TT::RX * TT::RY];
  __shared__ int sO[TT::OX * TT::OY];
  TT T{sF, sO};
  const int tid = threadIdx.x;
  const int ty0 = blockIdx.x * TY;           // global y of the tile origin
  const int txl0 = L.xo0 + blockIdx.y * TX;  // local row of the tile origin

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
    double f[9];
    // reinit_obst_density (main.c:966-986) acts on the PREVIOUS obstacle map with the current grain
    // state: nodes that were solid restart from the grain's equilibrium
    const int oo = interior ? ob_old[node] : -1;
    if (oo != -1) {
      grain_equilibrium(L, G, oo, gx, y, f);
    } else {
#pragma unroll
      for (int q = 0; q < 9; ++q) f[q] = fin[q * L.plane + node];
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
    fout[node] = T.F(0, px, py);
    fout[1 * L.plane + node] = pull_one<1>(T, L, G, px, py, gx, gy);
    fout[2 * L.plane + node] = pull_one<2>(T, L, G, px, py, gx, gy);
    fout[3 * L.plane + node] = pull_one<3>(T, L, G, px, py, gx, gy);
    fout[4 * L.plane + node] = pull_one<4>(T, L, G, px, py, gx, gy);
    fout[5 * L.plane + node] = pull_one<5>(T, L, G, px, py, gx, gy);
    fout[6 * L.plane + node] = pull_one<6>(T, L, G, px, py, gx, gy);
    fout[7 * L.plane + node] = pull_one<7>(T, L, G, px, py, gx, gy);
    fout[8 * L.plane + node] = pull_one<8>(T, L, G, px, py, gx, gy);
  }
}

// ---------------------------------------------------------------------------------------------
// hydrodynamic force and torque (main.c:1285-1333)
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ bool grain_box(const LatticeView& L, const GrainFluidView& G, int i, int& xi,
                                          int& xf, int& yi, int& yf) {
  const double xc = G.xc[i], yc = G.yc[i], rbl0 = G.rbl0[i];
  xi = (int)(xc - rbl0); if (xi < 1) xi = 1;                 // int max(double->int, 1): main.c:1300
  xf = (int)(xc + rbl0); if (xf > L.lx - 2) xf = L.lx - 2;   // main.c:1301
  yi = (int)(yc - rbl0); if (yi < 1) yi = 1;
  yf = (int)(yc + rbl0); if (yf > L.ly - 2) yf = L.ly - 2;
  return xi <= xf && yi <= yf;
}

// A grain is computed by the rank that owns the lattice column of its centre (first/last rank also
// take centres left/right of the lattice). On one GPU every grain is owned.
__device__ __forceinline__ bool grain_owned(const LatticeView& L, double xc) {
  const int lo = L.gx0 + L.xo0, hi = L.gx0 + L.xo1;  // owned global rows [lo, hi)
  const bool first = (lo == 0), last = (hi == L.lx);
  return (first || xc >= (double)lo) && (last || xc < (double)hi);
}

// Parity kernel: one thread per grain, the reference's serial x -> y -> q accumulation order.
__global__ void k_forces_parity(const double* __restrict__ f, const int* __restrict__ obst, LatticeView L,
                                GrainFluidView G, double scale12, double scale3,
                                double* __restrict__ fhf, unsigned char* __restrict__ owner) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= L.n) return;
  const double xc = G.xc[i], yc = G.yc[i];
  const bool own = grain_owned(L, xc);
  if (owner) owner[i] = own ? 1 : 0;
  double h1 = 0, h2 = 0, h3 = 0;
  int xi, xf, yi, yf;
  if (own && grain_box(L, G, i, xi, xf, yi, yf)) {
    for (int x = xi; x <= xf; ++x) {
      const long rowP = (long)(x - L.gx0) * L.sy;
      for (int y = yi; y <= yf; ++y) {
        if (obst[rowP + y] != i) continue;
#pragma unroll
        for (int q = 1; q < 9; ++q) {
          const int ex = EXq(q), ey = EYq(q), qo = OPPq(q);
          const long nodeN = (long)(x + ex - L.gx0) * L.sy + (y + ey);
          if (obst[nodeN] == i) continue;
          const double s = f[qo * L.plane + rowP + y] + f[q * L.plane + nodeN];
          const double fnx = s * EXq(qo);
          const double fny = s * EYq(qo);
          h1 = h1 + fnx;
          h2 = h2 + fny;
          h3 = h3 - fnx * (y - yc) + fny * (x - xc);
        }
      }
    }
  }
  if (own) {
    fhf[i] = h1 * scale12;
    fhf[L.n + i] = h2 * scale12;
    fhf[2 * L.n + i] = h3 * scale3;
  } else {
    fhf[i] = 0; fhf[L.n + i] = 0; fhf[2 * L.n + i] = 0;
  }
}

// Fast kernel: one wavefront per grain, lanes take bounding-box nodes, cross-lane shuffle reduction.
// Same terms as the parity kernel, different summation tree (differs in the last bits).
__global__ void k_forces_fast(const double* __restrict__ f, const int* __restrict__ obst, LatticeView L,
                              GrainFluidView G, double scale12, double scale3,
                              double* __restrict__ fhf, unsigned char* __restrict__ owner) {
  const int lane = threadIdx.x & 63;
  const int i = (int)(((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6);
  if (i >= L.n) return;
  const double xc = G.xc[i], yc = G.yc[i];
  const bool own = grain_owned(L, xc);
  double h1 = 0, h2 = 0, h3 = 0;
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
        const double s = f[qo * L.plane + rowP + y] + f[q * L.plane + nodeN];
        const double fnx = s * EXq(qo);
        const double fny = s * EYq(qo);
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
__global__ void k_aos_to_soa(const double* __restrict__ aos, double* __restrict__ f, LatticeView L) {
  const long total = (long)L.nxl * L.ly * 9;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k % 9);
    const long node = k / 9;
    const int y = (int)(node % L.ly), xl = (int)(node / L.ly);
    f[q * L.plane + (long)xl * L.sy + y] = aos[k];
  }
}
__global__ void k_soa_to_aos(const double* __restrict__ f, double* __restrict__ aos, LatticeView L, int xl0,
                             int nrows) {
  const long total = (long)nrows * L.ly * 9;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k % 9);
    const long node = k / 9;
    const int y = (int)(node % L.ly), xr = (int)(node / L.ly);
    aos[k] = f[q * L.plane + (long)(xl0 + xr) * L.sy + y];
  }
}

// init_density (main.c:716-724): f = w[q] everywhere
__global__ void k_fill_equilibrium(double* __restrict__ f, LatticeView L) {
  const long total = 9 * L.plane;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k / L.plane);
    f[k] = q == 0 ? 4. / 9 : ((q & 1) ? 1. / 36 : 1. / 9);
  }
}

// rho, rho*u sums in the order write_vtk forms them (main.c:315-319)
__global__ void k_macro(const double* __restrict__ f, LatticeView L, int xl0, int nrows,
                        double* __restrict__ rho, double* __restrict__ ux, double* __restrict__ uy) {
  const long total = (long)nrows * L.ly;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int y = (int)(k % L.ly), xr = (int)(k / L.ly);
    const long node = (long)(xl0 + xr) * L.sy + y;
    double s = 0.0, sx = 0.0, sy = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) {
      const double v = f[q * L.plane + node];
      s += v;
      sx += v * EXq(q);
      sy += v * EYq(q);
    }
    rho[k] = s; ux[k] = sx; uy[k] = sy;
  }
}

// Total mass, per-block partial sums over the owned rows (check_density, main.c:1249-1261).
// Summation order differs from the reference's serial sweep; compared with a tolerance.
__global__ void k_density_partial(const double* __restrict__ f, LatticeView L, double* __restrict__ partial) {
  __shared__ double red[256];
  const long rows = L.xo1 - L.xo0;
  const long total = rows * L.ly;
  double s = 0.0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int y = (int)(k % L.ly), xr = (int)(k / L.ly);
    const long node = (long)(L.xo0 + xr) * L.sy + y;
#pragma unroll
    for (int q = 0; q < 9; ++q) s += f[q * L.plane + node];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// halo rows <-> contiguous buffer [9][nrows][ly]
__global__ void k_halo_pack(const double* __restrict__ f, LatticeView L, int xl0, int nrows,
                            double* __restrict__ buf) {
  const long per = (long)nrows * L.ly, total = 9 * per;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k / per);
    const long r = k % per;
    const int y = (int)(r % L.ly), xr = (int)(r / L.ly);
    buf[k] = f[q * L.plane + (long)(xl0 + xr) * L.sy + y];
  }
}
__global__ void k_halo_unpack(double* __restrict__ f, LatticeView L, int xl0, int nrows,
                              const double* __restrict__ buf) {
  const long per = (long)nrows * L.ly, total = 9 * per;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < total; k += (long)gridDim.x * blockDim.x) {
    const int q = (int)(k / per);
    const long r = k % per;
    const int y = (int)(r % L.ly), xr = (int)(r / L.ly);
    f[q * L.plane + (long)(xl0 + xr) * L.sy + y] = buf[k];
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

void launch_obst_fill(int* obst, const LatticeView& L, hipStream_t st) {
  hipLaunchKernelGGL(k_obst_fill, dim3(grid_for((long)L.nxl * L.sy)), dim3(256), 0, st, obst, L);
}

void launch_grain_geom(int n, const double* x1, const double* x2, const double* r, const double* rLB,
                       double Mgx, double Mby, double dx, double* xc, double* yc, double* r2,
                       double* rbl0, hipStream_t st) {
  hipLaunchKernelGGL(k_grain_geom, dim3((n + 255) / 256), dim3(256), 0, st, n, x1, x2, r, rLB, Mgx, Mby,
                     dx, xc, yc, r2, rbl0);
}

void launch_obst_paint(int* obst, const LatticeView& L, const GrainFluidView& G, hipStream_t st) {
  const long threads = (long)L.n * 64;
  hipLaunchKernelGGL(k_obst_paint, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, obst, L, G);
}

void launch_collide_stream(const double* fin, double* fout, const int* obst_old, const int* obst_new,
                           const LatticeView& L, const GrainFluidView& G, hipStream_t st) {
  constexpr int TX = 8, TY = 64;
  const int rows = L.xo1 - L.xo0;
  dim3 grid((L.ly + TY - 1) / TY, (rows + TX - 1) / TX);
  hipLaunchKernelGGL((k_collide_stream<TX, TY>), grid, dim3(256), 0, st, fin, fout, obst_old, obst_new,
                     L, G);
}

void launch_forces_parity(const double* f, const int* obst, const LatticeView& L,
                          const GrainFluidView& G, double scale12, double scale3, double* fhf,
                          unsigned char* owner, hipStream_t st) {
  hipLaunchKernelGGL(k_forces_parity, dim3((L.n + 63) / 64), dim3(64), 0, st, f, obst, L, G, scale12,
                     scale3, fhf, owner);
}

void launch_forces_fast(const double* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                        double scale12, double scale3, double* fhf, unsigned char* owner,
                        hipStream_t st) {
  const long threads = (long)L.n * 64;
  hipLaunchKernelGGL(k_forces_fast, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, f, obst, L,
                     G, scale12, scale3, fhf, owner);
}

void launch_aos_to_soa(const double* aos_rows, double* f, const LatticeView& L, hipStream_t st) {
  hipLaunchKernelGGL(k_aos_to_soa, dim3(grid_for((long)L.nxl * L.ly * 9)), dim3(256), 0, st, aos_rows, f, L);
}
void launch_soa_to_aos(const double* f, double* aos_rows, const LatticeView& L, int xl0, int nrows,
                       hipStream_t st) {
  hipLaunchKernelGGL(k_soa_to_aos, dim3(grid_for((long)nrows * L.ly * 9)), dim3(256), 0, st, f, aos_rows, L,
                     xl0, nrows);
}
void launch_fill_equilibrium(double* f, const LatticeView& L, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_equilibrium, dim3(grid_for(9 * L.plane)), dim3(256), 0, st, f, L);
}
void launch_macro(const double* f, const LatticeView& L, int xl0, int nrows, double* rho, double* ux,
                  double* uy, hipStream_t st) {
  hipLaunchKernelGGL(k_macro, dim3(grid_for((long)nrows * L.ly)), dim3(256), 0, st, f, L, xl0, nrows, rho,
                     ux, uy);
}
void launch_density_partial(const double* f, const LatticeView& L, double* partial, int nblocks,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_density_partial, dim3(nblocks), dim3(256), 0, st, f, L, partial);
}
void launch_halo_pack(const double* f, const LatticeView& L, int xl0, int nrows, double* buf,
                      hipStream_t st) {
  hipLaunchKernelGGL(k_halo_pack, dim3(grid_for(9L * nrows * L.ly)), dim3(256), 0, st, f, L, xl0, nrows, buf);
}
void launch_halo_unpack(double* f, const LatticeView& L, int xl0, int nrows, const double* buf,
                        hipStream_t st) {
  hipLaunchKernelGGL(k_halo_unpack, dim3(grid_for(9L * nrows * L.ly)), dim3(256), 0, st, f, L, xl0, nrows,
                     buf);
}
