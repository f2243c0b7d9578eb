// dem_kernels.hip -- soft-disc DEM kernels of the MI355X LBM-DEM stepper (gfx950, wave64).
//
//  * k_dem_substep: ONE launch per DEM sub-step, one thread per grain. The reference's sub-step
//    (main.c:1733-1764) is three serial loops -- drift + half kick, acceleration_grains
//    (main.c:1336-1516), second half kick -- with a global dependency between the first and the
//    second (contacts need every grain's drifted state). Instead of two launches, each thread
//    re-derives the drifted state of its few partners from the previous state with the same
//    arithmetic (bit-identical), so the whole sub-step is a single read-old/write-new pass over a
//    ping-pong state. No atomics: every contact is evaluated by both partners in the reference's
//    canonical (i < j) frame (main.c:1442-1448) and partners are visited in ascending index, which
//    is exactly the order in which the reference's serial loop accumulates into a grain.
//
//  * Verlet list (main.c:1519-1594): the reference tests all N^2/2 pairs. Here grains are binned on
//    a uniform grid (cell = 2 r_max + distVerlet), sorted by cell with a one-digit radix sort of our own (the cell index
//    is the digit: histogram with arrival ranks, hipCUB exclusive scan, scatter -- 4 launches; hipCUB's pair sort took 7
//    at 50 000 keys and a rebuild is launch latency), and each
//    grain scans its 3x3 cells with the reference's three predicates (main.c:1529-1532) evaluated
//    in the (i < j) frame -> the same pair set. The list is kept symmetric (CSR, partners ascending).
//
// Diagnostics that depend on the serial contact order (slip, rw, fr, ice through pft/pff/pf/ic,
// main.c:130-131; they never feed back into x, v, a -- SURVEY.md hard part 7) are replayed in the reference's
// order by a separate small pipeline in the sub-steps that feed write_DEM: see "Order-dependent contact
// diagnostics" below.

#include "lbmdem_internal.h"
#include <vector>

#include <hipcub/hipcub.hpp>

namespace {

struct Force3 { real f1, f2, f3, fn, ft, xij, yij, vt; };  // fn, ft, branch vector, vt: for the diagnostics

__device__ __forceinline__ real maxt(real x, real y) { return (x < y) ? 0. : y; }  // main.c:211-216

struct GrainState { real x1, x2, v1, v2, v3, r; };

// drifted + half-kicked state of grain j from the previous sub-step's state: main.c:1748-1753
__device__ __forceinline__ GrainState advance(const Kin& K, const real* __restrict__ r, int j,
                                              const DemParams& P) {
  GrainState s;
  const real a1 = K.a1[j], a2 = K.a2[j], a3 = K.a3[j];
  const real v1 = K.v1[j], v2 = K.v2[j], v3 = K.v3[j];
  s.x1 = K.x1[j] + P.dt * v1 + P.dt2 * a1 / 2.;
  s.x2 = K.x2[j] + P.dt * v2 + P.dt2 * a2 / 2.;
  s.v1 = v1 + P.dt * a1 / 2.;
  s.v2 = v2 + P.dt * a2 / 2.;
  s.v3 = v3 + P.dt * a3 / 2.;
  s.r = r[j];
  return s;
}

// strip decomposition: a grain this rank does not integrate keeps its state, and BOTH ping-pong buffers carry it (a
// run of sub-steps in one launch swaps the buffers once, single sub-steps once each: nothing may depend on the parity)
__device__ __forceinline__ void carry_over(const Kin& in, const Kin& out, int i) {
  out.x1[i] = in.x1[i]; out.x2[i] = in.x2[i]; out.x3[i] = in.x3[i];
  out.v1[i] = in.v1[i]; out.v2[i] = in.v2[i]; out.v3[i] = in.v3[i];
  out.a1[i] = in.a1[i]; out.a2[i] = in.a2[i]; out.a3[i] = in.a3[i];
}

// contact force on grain A (lower index) from grain B (higher index).
// FILM = false: force_grains, main.c:739-774. FILM = true: the inline law of main.c:1365-1395.
template <bool FILM>
__device__ __forceinline__ Force3 contact(const GrainState& A, const GrainState& B, const DemParams& P,
                                          bool& touched) {
  Force3 F = {0., 0., 0., 0., 0., 0., 0., 0.};
  const real xij = A.x1 - B.x1;
  const real yij = A.x2 - B.x2;
  const real dist = (real)sqrt((double)(xij * xij + yij * yij));   // <math.h>'s double sqrt, rounded to real (main.c:742)
  const real dn = dist - A.r - B.r;
  touched = !(dn >= 0);
  if (dn >= 0) return F;
  const real vx = A.v1 - B.v1;
  const real vy = A.v2 - B.v2;
  const real xn = xij / dist;
  const real yn = yij / dist;
  const real vn = vx * xn + vy * yn;
  const real vt = -vx * yn + vy * xn - A.v3 * A.r - B.v3 * B.r;
  if (!FILM) {
    // force_grains declares `double fn, ft` (main.c:736) whatever `real` is: f1, f2 and the arguments of Maxt are formed
    // in double and rounded to real once (fn and ft themselves always hold real values)
    double fn = -P.kg * dn - P.nug * vn;
    if (fn < 0) fn = 0.0;
    double ft = -P.kt * vt * P.dt;
    const real ftest = P.mu * fn;
    if (fabs(ft) > ftest) ft = (ft < 0.0) ? ftest : -ftest;
    F.f3 = -maxt((real)(ft * A.r), (real)(fn * P.murf * A.r * B.r));
    F.f1 = fn * xn - ft * yn;
    F.f2 = fn * yn + ft * xn;
    F.fn = fn;
    F.ft = ft;
  } else {
    // the inline film law uses acceleration_grains' own `real fn, ft` (main.c:1340)
    real fn = -P.kg * dn - P.nug * vn;
    if (fn < 0) fn = 0.0;
    real ft = P.kt * vt * P.dt;
    const real ftest = P.mu * ft;  // sic, main.c:1385
    if (fabs((double)ft) > ftest) ft = (ft > 0.0) ? ftest : -ftest;
    F.f3 = -ft * A.r * P.murf;
    F.f1 = fn * xn - ft * yn;
    F.f2 = fn * yn + ft * xn;
    F.fn = fn;
    F.ft = ft;
  }
  F.xij = xij;
  F.yij = yij;
  F.vt = vt;
  return F;
}

// The four wall laws applied to one grain, in the reference's order bottom, top, left, right
// (main.c:1455-1508); wf = the grain's wall candidate flags (VerletWall, main.c:1563-1593).
// what the wall laws leave for the "previous contact" carries (CarryTrack): which walls the grain touched and the ft
// (bottom: also f3) the reference assigns to pft / pff / pf there (main.c:842-843, 919, 942, 1466, 1494)
struct WallHits { unsigned mask = 0; real ftB = 0., f3B = 0., ftL = 0., ftR = 0.; };

template <bool DIAG>
__device__ __forceinline__ void walls(const GrainState& me, unsigned wf, const DemParams& P, real& a1,
                                      real& a2, real& a3, real& pr, real& ds, real& df1, int& dz,
                                      real& dM11, real& dM12, real& dM21, real& dM22, WallHits& wh) {
  if (wf & 1u) {
    const real dn = me.x2 - me.r - P.Mby;
    if (dn < 0) {  // force_WallB, main.c:809-828
      const real vn = me.v2, vt = me.v1;
      real fn = -P.km * dn - P.num * vn;
      if (fn < 0) fn = 0.;
      real ft = P.ktm * vt;
      const real ftest = P.mumb * fn;
      if (fabs((double)ft) > ftest) ft = (ft < 0.0) ? ftest : -ftest;
      a1 = a1 + ft; a2 = a2 + fn; a3 = a3 + (-(ft * me.r * P.murf));
      wh.mask |= 1u; wh.ftB = ft; wh.f3B = -(ft * me.r * P.murf);
      pr += fn;
      if (DIAG) { ds += ft; df1 += ft; dz += 1; dM12 += ft * P.dt; dM22 += fn * P.dt; }  // main.c:830-838
    }
  }
  if (wf & 2u) {
    const real dn = -me.x2 - me.r + P.Mhy;
    if (dn < 0) {  // force_WallT, main.c:846-871
      const real vn = me.v2;
      real fn = P.km * dn - P.num * vn;
      if (fn > 0.) fn = 0.;
      const real vt = me.v1 + me.v3 * me.r - P.wallT_vel;   // wallT_vel = amp * freq * cos(freq * t) is a double (main.c:855)
      real ft = fabs((double)(P.ktm * vt));
      real ftmax;
      if (vt >= 0) ftmax = P.mumb * fn - P.nugt * vt; else ftmax = P.mumb * fn + P.nugt * vt;
      if (ft > ftmax) ft = ftmax;
      if (vt > 0) ft = -ft;
      a1 = a1 + ft; a2 = a2 + fn; a3 = a3 + ft * me.r * P.murf;
      if (DIAG) { dM12 += ft * fabs(P.dt); dM22 += fn * fabs(P.dt); }  // main.c:875-882
      pr += fn;
      if (DIAG) { ds += ft; dz += 1; }
    }
  }
  if (wf & 4u) {
    const real dn = me.x1 - me.r - P.Mgx;
    if (dn < 0) {  // force_WallL, main.c:888-904
      const real vn = me.v1;
      real fn = -P.km * dn + P.num * vn;
      if (fn < 0.) fn = 0.;
      const real vt = me.v2;
      real ft = P.mum * fn;
      if (vt > 0) ft = -ft;
      a1 = a1 + fn; a2 = a2 + ft; a3 = a3 + ft * me.r * P.murf;
      wh.mask |= 4u; wh.ftL = ft;
      if (DIAG) { dM11 += fn * fabs(P.dt); dM21 += ft * fabs(P.dt); }  // main.c:907-915
      pr += fn;
      if (DIAG) { ds += ft; df1 += fn; dz += 1; }
    }
  }
  if (wf & 8u) {
    const real dn = -me.x1 - me.r + P.Mdx;
    if (dn < 0) {  // force_WallR, main.c:923-936 (ft from the unclamped fn)
      const real vn = me.v1;
      real fn = P.km * dn - P.num * vn;
      const real vt = me.v2;
      real ft = P.mum * fn;
      if (vt > 0) ft = -ft;
      wh.mask |= 8u; wh.ftR = ft;
      if (fn > 0.) fn = 0.;
      a1 = a1 + fn; a2 = a2 + (-ft); a3 = a3 + ft * me.r * P.murf;
      pr += fn;
      if (DIAG) { df1 += fn; dM11 += fn * fabs(P.dt); dM21 += (-ft) * fabs(P.dt); dz += 1; }  // main.c:938-949
    }
  }
}

// Per-grain contact diagnostics of one sub-step (the fields write_DEM prints, main.c:413-420), in the
// reference's accumulation order. Only produced when DIAG (the sub-step before an output).
struct DiagOut { real *s, *f1, *f2, *ifm, *M11, *M12, *M21, *M22; int *z, *zz; DiagExtra X; };

template <bool FILM, bool DIAG>
__global__ void k_dem_substep(Kin in, Kin out, const real* __restrict__ r, const real* __restrict__ m,
                              const real* __restrict__ It, const real* __restrict__ fhf,
                              const int* __restrict__ offsets, const int* __restrict__ nbr,
                              const unsigned char* __restrict__ wallflags, real* __restrict__ pout,
                              DiagOut D, DemParams P, const unsigned char* __restrict__ active) {
  LBMDEM_GATE(P.gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= P.n) return;
  if (active && !active[i]) { carry_over(in, out, i); return; }  // strip decomposition: a grain another rank integrates
  const GrainState me = advance(in, r, i, P);
  const real x3 = in.x3[i] + P.dt * in.v3[i] + P.dt2 * in.a3[i] / 2.;

  // acceleration_grains: start from the hydrodynamic force (main.c:1429-1431)
  real a1 = fhf[i], a2 = fhf[P.n + i], a3 = fhf[2 * P.n + i];
  // grain pressure g.p (main.c:1734, 776-777, 830, 880, 912, 938): reset every sub-step, += fn per
  // contact in the same order as the accelerations; a separated pair (dn >= 0) adds nothing
  real pr = 0.0;
  bool touched;
  real ds = 0., df1 = 0., df2 = 0., difm = 0., dM11 = 0., dM12 = 0., dM21 = 0., dM22 = 0.;
  int dz = 0, dzz = 0;
  const int k0 = offsets[i], k1 = offsets[i + 1];
  for (int k = k0; k < k1; ++k) {
    const int j = nbr[k];
    const GrainState other = advance(in, r, j, P);
    if (i < j) {  // main.c:1443-1445
      const Force3 F = contact<FILM>(me, other, P, touched);
      a1 = a1 + F.f1; a2 = a2 + F.f2; a3 = a3 + F.f3;
      if (touched) pr += F.fn;
      if (DIAG) {  // what the carry pass needs of this contact (launch_diag_extra)
        D.X.e_touched[k] = touched ? 1 : 0;
        D.X.e_ft[k] = F.ft; D.X.e_f3[k] = F.f3;
        D.X.e_avt[k] = fabs(F.vt * P.dt); D.X.e_av3[k] = fabs(me.v3 * P.dt);
      }
      if (DIAG && touched) {  // this grain is the `i` of force_grains(i, j): main.c:776-799 / 1397-1416
        ds += F.ft;
        dz += 1;
        dM11 += F.f1 * F.xij; dM12 += F.f1 * F.yij; dM21 += F.f2 * F.xij; dM22 += F.f2 * F.yij;
        if (!FILM) {
          df1 += F.f1; df2 += F.f2;
          dzz += 1;
          if (F.fn == 0) difm = 0; else difm += fabs(F.ft / (P.mu * F.fn));
        }
      }
    } else {      // main.c:1446-1448
      const Force3 F = contact<FILM>(other, me, P, touched);
      a1 = a1 - F.f1; a2 = a2 - F.f2; a3 = a3 + F.f3;
      if (touched) pr += F.fn;
      if (DIAG && touched) ds += F.ft;  // the `j` side only receives p and s
      if (DIAG) D.X.e_touched[k] = 0;
    }
  }
  if (DIAG) D.X.a1gc[i] = a1;
  // walls: bottom, top, left, right (main.c:1455-1508)
  WallHits wh;
  walls<DIAG>(me, wallflags[i], P, a1, a2, a3, pr, ds, df1, dz, dM11, dM12, dM21, dM22, wh);
  // main.c:1511-1515 (mw = 0: SURVEY.md hard part 5)
  const real mi = m[i], mw = 0.0;
  a1 = a1 / mi + ((mi - mw) / mi) * P.xG;
  a2 = (a2 / mi) + ((mi - mw) / mi) * P.yG;
  a3 = a3 / It[i];
  // second half kick, main.c:1760-1762
  out.x1[i] = me.x1; out.x2[i] = me.x2; out.x3[i] = x3;
  out.v1[i] = me.v1 + P.dt * a1 / 2.;
  out.v2[i] = me.v2 + P.dt * a2 / 2.;
  out.v3[i] = me.v3 + P.dt * a3 / 2.;
  out.a1[i] = a1; out.a2[i] = a2; out.a3[i] = a3;
  pout[i] = pr;
  if (DIAG) {
    D.s[i] = ds; D.f1[i] = df1; D.f2[i] = df2; D.ifm[i] = difm;
    D.M11[i] = dM11; D.M12[i] = dM12; D.M21[i] = dM21; D.M22[i] = dM22;
    D.z[i] = dz; D.zz[i] = dzz;
  }
}

// ---------------------------------------------------------------------------------------------
// The sub-step, one lane per CONTACT CANDIDATE
// ---------------------------------------------------------------------------------------------
//
// k_dem_substep spends its time in arithmetic, not in memory: one thread walks all partners of its
// grain, ~150 dependent fp64 instructions each (a square root and two divisions), twice because
// lanes with i < j and i > j take different branches -- ~14 000 cycles of one wave per SIMD while
// three quarters of the chip's issue slots idle. Here the Verlet list itself is the parallel axis:
//   phase 1  one lane per list entry (grain i, partner j): the contact force in the canonical
//            (lower, higher) frame (operands selected, not branched), signed for i, staged in LDS;
//   phase 2  one lane per grain adds its entries in list order (ascending partner = the reference's
//            accumulation order, main.c:1442-1448), applies the wall laws and the second half kick.
// A workgroup owns DEM_GRAINS consecutive grains and therefore a contiguous slice of the list.
// x - y == x + (-y) in IEEE arithmetic, so `a1 + (i < j ? f1 : -f1)` is bit for bit the reference's
// `a1 + f1` / `a1 - f1`.
constexpr int DEM_GRAINS = DEM_TILE;  // grains per workgroup
#define DEM_GRID(nbe) ((((nbe) + 7) / 8) * 8)   /* tile slots: a multiple of the 8 XCDs, see the kernel */
constexpr int DEM_ENTRIES = 512;  // list entries staged per round (8 per grain; denser lists take more rounds)
#ifndef DEM_THREADS
#define DEM_THREADS 256           /* lanes per workgroup of k_dem_entries (phase 1: one list entry per lane and pass) */
#endif

template <bool FILM>
__global__ __launch_bounds__(DEM_THREADS) void k_dem_entries(Kin in, Kin out, const real* __restrict__ r,
                                                     const real* __restrict__ m,
                                                     const real* __restrict__ It,
                                                     const real* __restrict__ fhf,
                                                     const int* __restrict__ offsets,
                                                     const int* __restrict__ nbr, const int* __restrict__ own,
                                                     const unsigned char* __restrict__ wallflags,
                                                     real* __restrict__ pout, DemParams P,
                                                     const unsigned char* __restrict__ active, CarryTrack T,
                                                     long long stamp, const unsigned char* __restrict__ owner,
                                                     ObstFillJob fill, int tiles) {
  LBMDEM_GATE(P.gate);
  __shared__ real sF1[DEM_ENTRIES], sF2[DEM_ENTRIES], sF3[DEM_ENTRIES], sFn[DEM_ENTRIES];
  __shared__ unsigned char sTouched[DEM_ENTRIES];
  __shared__ int sLast;   // highest list entry of this tile that is a touching contact in the reference's frame
  const int tid = threadIdx.x;
  // Workgroups are handed out round-robin over the 8 XCDs (b % 8), each with a private L2. XCD k takes the k-th CONTIGUOUS
  // eighth of the tiles: a tile's partners sit in the neighbouring tiles (grains are numbered along the packing's rows),
  // and every sub-step finds the state its own XCD wrote in the sub-step before: 9.5 -> 8.5 us per sub-step (A/B, round 4).
  const int tslots = ((tiles + 7) / 8) * 8;
  if ((int)blockIdx.x >= tslots) {   // the workgroups behind the grain tiles reset a slice of the next obstacle map
    obst_fill_range(fill.map, fill.L, (long)(blockIdx.x - tslots) * DEM_THREADS + tid, (long)(gridDim.x - tslots) * DEM_THREADS, fill.row0, fill.row1);
    return;
  }
  const int tile = ((int)blockIdx.x & 7) * (tslots >> 3) + ((int)blockIdx.x >> 3);
  if (tile >= tiles) return;
  if (T.stamp) {   // before anything is in flight: the barrier costs nothing here
    if (tid == 0) sLast = -1;
    __syncthreads();
  }
  const int g0 = tile * DEM_GRAINS;
  const int g1 = g0 + DEM_GRAINS < P.n ? g0 + DEM_GRAINS : P.n;
  // phase-2 lanes: everything that does not depend on the partners is requested now
  const int i = g0 + tid;
  // strip decomposition: only the grains this rank integrates (owned + margin); the others keep whatever they hold
  const bool mine = tid < DEM_GRAINS && i < P.n && (!active || active[i]);
  if (active && !__syncthreads_or(mine ? 1 : 0)) {
    if (tid < DEM_GRAINS && i < P.n) carry_over(in, out, i);
    return;
  }
  if (active && !mine && tid < DEM_GRAINS && i < P.n) carry_over(in, out, i);
  const int e0 = offsets[g0], e1 = offsets[g1];
  GrainState me{};
  real x3 = 0., a1 = 0., a2 = 0., a3 = 0., pr = 0.0, mi = 1., Iti = 1.;
  int k0 = 0, k1 = 0;
  unsigned wf = 0;
  if (mine) {
    me = advance(in, r, i, P);
    x3 = in.x3[i] + P.dt * in.v3[i] + P.dt2 * in.a3[i] / 2.;
    a1 = fhf[i]; a2 = fhf[P.n + i]; a3 = fhf[2 * P.n + i];  // main.c:1429-1431
    k0 = offsets[i]; k1 = offsets[i + 1];
    wf = wallflags[i];
    mi = m[i]; Iti = It[i];
  }
  int last_e = -1;            // this thread's youngest touching contact (own < partner: the reference evaluates it there)
  real last_ft = 0., last_f3 = 0.;
  long long last_who = 0;     // ... and the pair, (own << 32) | partner: the reference's contact order, whatever the list
  // strips with distributed grains: a rank only vouches for (and records) the contacts of the grains it OWNS -- every
  // contact is evaluated in the frame of its lower grain, so exactly one rank records it
  for (int base = e0; base < e1; base += DEM_ENTRIES) {
    const int lim = base + DEM_ENTRIES < e1 ? base + DEM_ENTRIES : e1;
    for (int e = base + tid; e < lim; e += DEM_THREADS) {
      const int gi = own[e], gj = nbr[e];
      if (active && !active[gi]) continue;   // nobody adds this entry up
      const GrainState a = advance(in, r, gi, P), b = advance(in, r, gj, P);
      const bool lower = gi < gj;
      bool touched;
      const Force3 F = contact<FILM>(lower ? a : b, lower ? b : a, P, touched);  // main.c:1443-1448
      if (touched && lower && (!owner || owner[gi])) {
        last_e = e; last_ft = F.ft; last_f3 = F.f3; last_who = ((long long)gi << 32) | (unsigned)gj;
        if (T.stamp) atomicMax(&sLast, e);
      }
      const int s = e - base;
      sF1[s] = lower ? F.f1 : -F.f1;
      sF2[s] = lower ? F.f2 : -F.f2;
      sF3[s] = F.f3;
      sFn[s] = F.fn;
      sTouched[s] = touched ? 1 : 0;
    }
    __syncthreads();
    if (mine) {
      const int lo = k0 > base ? k0 : base, hi = k1 < lim ? k1 : lim;
      for (int k = lo; k < hi; ++k) {
        const int s = k - base;
        a1 = a1 + sF1[s]; a2 = a2 + sF2[s]; a3 = a3 + sF3[s];
        if (sTouched[s]) pr += sFn[s];
      }
    }
    __syncthreads();
  }
  if (T.stamp) {  // the tile's last grain contact, for the carries (every round of the loop above ends with a barrier)
    if (last_e >= 0 && last_e == sLast) {
      const long rec = (long)tile * 4 + CARRY_GRAIN;
      T.stamp[rec] = stamp; T.val[2 * rec] = last_ft; T.val[2 * rec + 1] = last_f3; T.who[rec] = last_who;
    }
  }
  if (tid >= 64) return;   // the first wavefront holds the tile's grains
  real ds = 0., df1 = 0., dM11 = 0., dM12 = 0., dM21 = 0., dM22 = 0.;
  int dz = 0;
  WallHits wh;
  if (mine) walls<false>(me, wf, P, a1, a2, a3, pr, ds, df1, dz, dM11, dM12, dM21, dM22, wh);
  if (T.stamp) {  // per wall, the highest grain of the tile that touched it (the wall lists are in grain order)
#pragma unroll
    for (int kind = CARRY_BOTTOM; kind <= CARRY_RIGHT; ++kind) {
      const unsigned bit = kind == CARRY_BOTTOM ? 1u : (kind == CARRY_LEFT ? 4u : 8u);
      const unsigned long long hit = __ballot((wh.mask & bit) != 0 && (!owner || (mine && owner[i])));
      if (hit != 0 && tid == 63 - __builtin_clzll(hit)) {
        const long rec = (long)tile * 4 + kind;
        T.stamp[rec] = stamp;
        T.who[rec] = (long long)i << 32;
        T.val[2 * rec] = kind == CARRY_BOTTOM ? wh.ftB : (kind == CARRY_LEFT ? wh.ftL : wh.ftR);
        T.val[2 * rec + 1] = wh.f3B;
      }
    }
  }
  if (!mine) return;
  const real mw = 0.0;
  a1 = a1 / mi + ((mi - mw) / mi) * P.xG;
  a2 = (a2 / mi) + ((mi - mw) / mi) * P.yG;
  a3 = a3 / Iti;
  out.x1[i] = me.x1; out.x2[i] = me.x2; out.x3[i] = x3;
  out.v1[i] = me.v1 + P.dt * a1 / 2.;
  out.v2[i] = me.v2 + P.dt * a2 / 2.;
  out.v3[i] = me.v3 + P.dt * a3 / 2.;
  out.a1[i] = a1; out.a2[i] = a2; out.a3[i] = a3;
  pout[i] = pr;
}

// ---------------------------------------------------------------------------------------------
// Several sub-steps in ONE launch
// ---------------------------------------------------------------------------------------------
//
// A sub-step of k_dem_entries costs ~8 us of which ~2 are arithmetic: the rest is the dependent launch and three
// dependent rounds of memory loads, and npDEM (12) of them sit between two fluid steps. k_dem_chain runs a whole run of
// ordinary sub-steps (regular contact law, no diagnostics, no list rebuild in between) in one launch:
//   * a workgroup owns the same tile of 64 consecutive grains for the whole launch; the first wavefront keeps their
//     kinematics in registers from sub-step to sub-step;
//   * what a sub-step needs of OTHER tiles -- the drifted state (x1, x2, v1, v2, v3: main.c:1748-1753) of the tile's
//     "halo" grains, the distinct partners outside the tile that k_tile_halo listed after the last list rebuild -- travels
//     through `pub`: per grain and parity of the sub-step one 128-byte line of five 16-byte slots {lo, tag, hi, tag}, tag =
//     the sub-step's sequence number. Stores are write-through (sc1), loads bypass the L1 (sc1): the per-XCD L2s are not
//     coherent with each other and an L1 never sees another CU's stores. Each 8-byte half carries its own tag, so the
//     data is its own flag: a reader re-reads a slot until both tags match -- no flag round trip, no fence, no grid
//     barrier; a tile only ever waits for the tiles of its own partners.
//   * TWO copies of every line. 782 tiles polling ~135 lines each through the fabric cost 17 us per sub-step (first
//     version, profiles/r05_dem_chain_ab.txt). Tiles are laid out in contiguous eighths per XCD (block b runs on XCD
//     b % 8 -- observed, not promised), so nearly all partners of a tile run on its own XCD, whose L2 IS coherent for
//     them: the LOCAL copy is written with plain stores (they stay in that L2) and read with L1-bypassing loads (L2 hits),
//     the REMOTE copy is written through to memory (sc1) for the readers on other XCDs -- by the tiles that have a partner
//     in another eighth (tile_far; the list is symmetric, so these are the tiles somebody reads from another eighth). A
//     reader picks the copy by the XCD the partner's tile is expected on, for staged halo grains and for partners beyond
//     the staging alike. A workgroup that did not land where expected leaves its readers waiting for a local copy they
//     cannot see: their bounded spins run out, the launch gives up -- and is undone (below). The tags make a wrong guess
//     slow, never wrong.
//   * two parities suffice: a tile overwrites its sub-step s-1 line in sub-step s+1, i.e. after it has read the sub-step s
//     state of every halo grain, which their tiles published after they had finished reading for sub-step s-1 (the list
//     is symmetric: whoever reads this tile is read by it).
//   * the drifted state is computed ONCE, by the owner, with the arithmetic every partner used to repeat (advance()):
//     same bits. Phase 1 / phase 2 are those of k_dem_entries, fed from LDS.
// Every spin is bounded; a tile that gives up poisons its lines so that its partners give up at once, reports the launch
// (its first sequence number) to the host and raises the handle's stop word, which every kernel of the step path reads
// first: whatever is queued behind the launch does nothing, the device keeps the state the launch started from (its input
// buffers are the other half of the ping-pong), and the host goes back there and repeats the sub-steps one launch each
// (lbmdem_capi.hip, chain_settle_impl). All workgroups of the tile slots must be resident together -- checked once per
// handle by a census launch of the same kernel (dem_chain_census), not promised by HIP afterwards: another process on the
// GPU, a CU mask.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int CH_SLOTS = DEM_TILE + DEM_CHAIN_HALO;   // grains staged in LDS: the tile's own, then its halo
constexpr int CH_ENTRIES = 448;                       // list entries staged per round (k_dem_entries: DEM_ENTRIES); the reference's own 50 000-grain packing has up to 387 per tile
constexpr int CH_META = 768;                          // list entries whose emeta word is kept in LDS
constexpr int CH_ITEMS = 3;                          // 16-byte slots a lane has in flight per pass of the halo fetch (153 grains)
constexpr unsigned CH_SPINS = 1u << 22;   // (~ seconds: a workgroup that is late -- the first launch of a process, another stream's kernels -- is not one that never comes)
constexpr unsigned CH_POISON = 0xFFFFFFFEu;
constexpr unsigned CH_SKIP = 0xFFFFFFFFu;             // emeta of an entry nobody adds up (its grain is not integrated here)

__host__ __device__ __forceinline__ unsigned chain_tag(long long seq) {   // 1 .. 2^31 - 3, consecutive sub-steps differ (| 2^31: a launch's static tag, never CH_POISON)
  return (unsigned)((unsigned long long)seq % 0x7FFFFFFDull) + 1u;
}

struct ChainRead { real x1, x2, v1, v2, v3; };

// one 16-byte slot straight from memory (a partner beyond the staged halo; rare): bounded spin
__device__ __forceinline__ bool chain_read_slot(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, unsigned stag, real& out) {
  unsigned spins = 0;
#pragma nounroll
  for (;;) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
    if (v.y == CH_POISON || v.w == CH_POISON || ++spins > CH_SPINS) return false;
    if ((v.y == tag || v.y == stag) && (v.w == tag || v.w == stag)) {
      out = __longlong_as_double((long long)(((unsigned long long)v.z << 32) | v.x));
      return true;
    }
    __builtin_amdgcn_s_sleep(2);
  }
}
__device__ __forceinline__ bool chain_read_direct(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, unsigned stag,
                                                  ChainRead& R) {
  return chain_read_slot(rs, off, tag, stag, R.x1) && chain_read_slot(rs, off + 16u, tag, stag, R.x2) &&
         chain_read_slot(rs, off + 32u, tag, stag, R.v1) && chain_read_slot(rs, off + 48u, tag, stag, R.v2) &&
         chain_read_slot(rs, off + 64u, tag, stag, R.v3);
}

template <int AUX>   // 0: plain store (stays in the XCD's L2), 16 = sc1: written through to memory
__device__ __forceinline__ void chain_publish(__amdgpu_buffer_rsrc_t rs, unsigned off, unsigned tag, real x1, real x2,
                                              real v1, real v2, real v3) {
  const real val[5] = {x1, x2, v1, v2, v3};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    const unsigned long long b = (unsigned long long)__double_as_longlong((double)val[k]);
    u32x4 v; v.x = (unsigned)b; v.y = tag; v.z = (unsigned)(b >> 32); v.w = tag;
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, off + 16u * k, 0, AUX);
  }
}

__global__ __launch_bounds__(DEM_THREADS, 4) void k_dem_chain(Kin in, Kin out, const real* __restrict__ r,
                                                           const real* __restrict__ m, const real* __restrict__ It,
                                                           const real* __restrict__ fhf, const int* __restrict__ offsets,
                                                           const int* __restrict__ nbr, const unsigned* __restrict__ emeta,
                                                           const int* __restrict__ halo_ids,
                                                           const int* __restrict__ halo_cnt,
                                                           const int* __restrict__ tile_grains,
                                                           const int* __restrict__ where,
                                                           const unsigned char* __restrict__ tile_far,
                                                           const unsigned char* __restrict__ wallflags,
                                                           real* __restrict__ pout, DemParams P,
                                                           const unsigned char* __restrict__ active, CarryTrack T,
                                                           long long stamp0, const unsigned char* __restrict__ owner,
                                                           ObstFillJob fill, int tiles, int nsteps, void* pub,
                                                           unsigned pub_bytes, int* __restrict__ err, int* census,
                                                           long long* dbg, int flags, ChainPaint paint, int* gate) {
  LBMDEM_GATE(P.gate);
  __shared__ real sF1[CH_ENTRIES], sF2[CH_ENTRIES], sF3[CH_ENTRIES], sFn[CH_ENTRIES];
  __shared__ unsigned char sTouched[CH_ENTRIES];
  __shared__ real sS[5 * CH_SLOTS];     // drifted x1, x2, v1, v2, v3 of the staged grains, one array per field
  __shared__ real sR[CH_SLOTS];
  __shared__ unsigned sMeta[CH_META];
  __shared__ unsigned char sFlag[DEM_TILE];   // bit 0: integrated here, bit 1: owned (records its contacts)
  __shared__ unsigned sHoff[DEM_CHAIN_HALO];  // the halo grains' lines in `pub` (byte offset within a parity)
  // per grain of the tile, kept out of the registers: the angle (no contact needs it), the hydrodynamic force, 1 / the
  // inertia terms' divisors and the gravity terms
  __shared__ real sX3[DEM_TILE], sFh[3 * DEM_TILE], sMI[2 * DEM_TILE], sG[2 * DEM_TILE];
  __shared__ int sK[2 * DEM_TILE];            // first and last list entry of the tile's grains
  __shared__ int sPre[DEM_TILE + 1];          // ... and where their entries stand among the tile's (the tile's entries: its grains' list ranges one after the other)
  __shared__ int sLast, sFail, sScan[2];
  const int tid = threadIdx.x;
  const bool one_xcd = (flags & 1) != 0;
  // one_xcd: every eighth block is a tile (they all land on one XCD, whose L2 then carries every hand-off), the others idle
  const int tslots = one_xcd ? tiles * 8 : ((tiles + 7) / 8) * 8;
  if ((int)blockIdx.x >= tslots) {   // the workgroups behind the tile slots reset a slice of the next obstacle map
    obst_fill_range(fill.map, fill.L, (long)(blockIdx.x - tslots) * DEM_THREADS + tid, (long)(gridDim.x - tslots) * DEM_THREADS, fill.row0, fill.row1);
    return;
  }
  if (nsteps < 0) {   // census: do all tile slots run at the same time, and do blocks that agree mod 8 share an XCD?
    if (tid == 0) {
      unsigned xcc_id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_id));
      // bit (8 * (block mod 8) + XCD): a regular placement sets exactly one bit per byte, eight different ones
      atomicOr(reinterpret_cast<unsigned long long*>(census + 2), 1ull << (8 * (blockIdx.x & 7) + (xcc_id & 7u)));
      __hip_atomic_fetch_add(census, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      unsigned spins = 0;
      while (__hip_atomic_load(census, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < tslots) {
        __builtin_amdgcn_s_sleep(8);
        if (++spins > (1u << 15)) { *err = 1; break; }
      }
    }
    return;
  }
  if (one_xcd && ((int)blockIdx.x & 7) != 0) return;
  const int tile = one_xcd ? (int)blockIdx.x >> 3 : ((int)blockIdx.x & 7) * (tslots >> 3) + ((int)blockIdx.x >> 3);   // as in k_dem_entries
  if (tile >= tiles) return;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(pub, 0, (int)pub_bytes, 0x00020000);
  const unsigned line_par = (unsigned)P.n * 128u;   // bytes between the two parities
  const unsigned remote = 2u * line_par;            // ... and between the local and the remote copy
  // Workgroups go to the XCDs round-robin (measured: block b on the XCD with HW_REG_XCC_ID (b + 7) % 8, every launch), so
  // two blocks share an XCD iff their indices agree mod 8 -- the tiles of one contiguous eighth. HIP does not promise it:
  // the census launch checks it once per handle, and a reader whose local copy stays stale gives up (loudly).
  const int tiles_per_xcd = tslots >> 3;
#ifdef LBMDEM_CHAIN_TIMING
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const bool placed = true;
#endif
  // the tile's grains: a patch of the packing (VerletDevice::tile_grains), ascending, one per lane of the first wavefront
  const int i = tid < DEM_GRAINS ? tile_grains[(long)tile * DEM_TILE + tid] : -1;
  const bool have = i >= 0;
  const bool mine = have && (!active || active[i]);
  const unsigned tag0 = chain_tag(stamp0), stag = 0x80000000u | tag0;
  // the grains of this tile, as the previous sub-step left them
  real x1 = 0., x2 = 0., v1 = 0., v2 = 0., v3 = 0., a1 = 0., a2 = 0., a3 = 0.;
  if (have) {
    sX3[tid] = in.x3[i];
    x1 = in.x1[i]; x2 = in.x2[i]; v1 = in.v1[i]; v2 = in.v2[i]; v3 = in.v3[i];
    a1 = in.a1[i]; a2 = in.a2[i]; a3 = in.a3[i];
  }
  if (active && !__syncthreads_or(mine ? 1 : 0)) {
    // nobody here is integrated by this rank: the state stands for the whole launch (both parities, the launch's static
    // tag), and follows the ping-pong of the buffers
    if (have) {
      for (unsigned par = 0; par < 2; ++par) {
        chain_publish<0>(rs, par * line_par + (unsigned)i * 128u, stag, x1, x2, v1, v2, v3);
        chain_publish<16>(rs, remote + par * line_par + (unsigned)i * 128u, stag, x1, x2, v1, v2, v3);
      }
      out.x1[i] = x1; out.x2[i] = x2; out.x3[i] = sX3[tid]; out.v1[i] = v1; out.v2[i] = v2; out.v3[i] = v3;
      out.a1[i] = a1; out.a2[i] = a2; out.a3[i] = a3;
    }
    return;
  }
  {   // the grains' list ranges and their places among the tile's entries
    const int kk0 = have ? offsets[i] : 0, cnt = have ? offsets[i + 1] - kk0 : 0;
    if (tid < DEM_GRAINS) {
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(incl, d);
        if (tid >= d) incl += up;
      }
      sK[tid] = kk0; sK[DEM_TILE + tid] = kk0 + cnt; sPre[tid + 1] = incl;
      if (tid == 0) sPre[0] = 0;
    }
  }
  __syncthreads();
  const int E = sPre[DEM_TILE];   // entries of the tile
  // position among the tile's entries -> the list entry (the few places that need the entry itself: a partner beyond the
  // staged halo, the record of the tile's last contact, entries beyond the staged meta words)
  auto entry_of = [&](int p) {
    int lo = 0, hi = DEM_TILE - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sPre[mid] <= p) lo = mid; else hi = mid - 1; }
    return sK[lo] + (p - sPre[lo]);
  };
  const int hcnt = halo_cnt[tile];
  const int nitems = hcnt * 5;
  for (int h = tid; h < hcnt; h += DEM_THREADS) {
    const int g = halo_ids[(long)tile * DEM_CHAIN_HALO + h];
    // bit 0: read the remote copy (the grain's tile is expected on another XCD)
    sHoff[h] = (unsigned)g * 128u | ((!one_xcd && (where[g] >> 6) / tiles_per_xcd != tile / tiles_per_xcd) ? 1u : 0u);
    sR[DEM_TILE + h] = r[g];
  }
  auto grain_at = [&](int li) { return tile_grains[(long)tile * DEM_TILE + li]; };   // (rare paths)
  for (int k = tid; k < CH_META && k < E; k += DEM_THREADS) {
    unsigned w = emeta[entry_of(k)];
    if (active && !active[grain_at((int)(w & 63u))]) w = CH_SKIP;
    sMeta[k] = w;
  }
  real pr = 0.0;
  int k0 = 0, k1 = 0;
  unsigned wf = 0;
  if (tid < DEM_GRAINS) {
    sR[tid] = have ? r[i] : (real)0.;
    sFlag[tid] = (unsigned char)((mine ? 1 : 0) | ((mine && (!owner || owner[i])) ? 2 : 0));
  }
  if (mine) {
    sFh[tid] = fhf[i]; sFh[DEM_TILE + tid] = fhf[P.n + i]; sFh[2 * DEM_TILE + tid] = fhf[2 * P.n + i];   // main.c:1429-1431: constant between two fluid steps
    k0 = sPre[tid]; k1 = sPre[tid + 1];   // this grain's entries, as positions among the tile's
    wf = wallflags[i];
    const real mi = m[i], mw = 0.0;
    sMI[tid] = mi; sMI[DEM_TILE + tid] = It[i];
    // main.c:1511-1512: the gravity term of a grain's acceleration does not change from sub-step to sub-step
    sG[tid] = ((mi - mw) / mi) * P.xG; sG[DEM_TILE + tid] = ((mi - mw) / mi) * P.yG;
  }
  if (tid == 0) sFail = 0;
  __syncthreads();

#ifdef LBMDEM_CHAIN_TIMING   /* experiment build: where a tile's time goes (100 MHz clock), dbg[tile][16] */
  long long t_wait = 0, t_work = 0, n_spins = 0, t_begin = wall_clock64();
  long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm = t_begin;
#define CH_MARK(k) do { const long long now_ = wall_clock64(); ph[k] += now_ - tm; tm = now_; } while (0)
#else
#define CH_MARK(k) do { } while (0)
#endif
  // A tile with a partner on another XCD publishes both copies (only the remote one, for all its readers: measured slower,
  // 137 k against 154 k sub-steps/s -- more tiles then wait for the fabric)
  const bool any_far = !one_xcd && tile_far[tile];
  // which copy of grain g's lines a reader in THIS tile takes when g is not among the staged halo grains (as sHoff's bit 0)
  auto direct_copy = [&](int g) -> unsigned {
    return (!one_xcd && (where[g] >> 6) / tiles_per_xcd != tile / tiles_per_xcd) ? remote : 0u;
  };
  GrainState me{};
  me.r = tid < DEM_GRAINS ? sR[tid] : (real)0.;
  // drift + first half kick (main.c:1748-1753) of the sub-step with sequence number `stamp`, published to the partners'
  // tiles and to this tile's LDS
  auto drift_publish = [&](long long stamp) {
    if (tid < DEM_GRAINS) {
      const unsigned pb = (stamp & 1) ? line_par : 0u;
      if (mine) {
        x1 = x1 + P.dt * v1 + P.dt2 * a1 / 2.;
        x2 = x2 + P.dt * v2 + P.dt2 * a2 / 2.;
        sX3[tid] = sX3[tid] + P.dt * v3 + P.dt2 * a3 / 2.;
        v1 = v1 + P.dt * a1 / 2.;
        v2 = v2 + P.dt * a2 / 2.;
        v3 = v3 + P.dt * a3 / 2.;
      }
      if (have) {
        // (the remote copy first: measured 1.3 % slower)
        chain_publish<0>(rs, pb + (unsigned)i * 128u, chain_tag(stamp), x1, x2, v1, v2, v3);
        if (any_far) chain_publish<16>(rs, remote + pb + (unsigned)i * 128u, chain_tag(stamp), x1, x2, v1, v2, v3);
      }
      sS[tid] = x1; sS[CH_SLOTS + tid] = x2; sS[2 * CH_SLOTS + tid] = v1; sS[3 * CH_SLOTS + tid] = v2; sS[4 * CH_SLOTS + tid] = v3;
      me.x1 = x1; me.x2 = x2; me.v1 = v1; me.v2 = v2; me.v3 = v3;
      if (tid == 0) sLast = -1;
    }
  };
  drift_publish(stamp0);
  for (int s = 0; s < nsteps; ++s) {
#ifdef LBMDEM_CHAIN_TIMING
    const long long t_a = wall_clock64();
    tm = t_a;
#endif
    const long long stamp = stamp0 + s;
    const unsigned tag = chain_tag(stamp);
    const unsigned pbase = (stamp & 1) ? line_par : 0u;
#ifdef LBMDEM_AB   // lbmdem_debug_chain_giveup (tests): one tile behaves as if a partner never showed up
    if (((flags >> 17) & 1) && tile == 3 % tiles && s == nsteps / 2 && tid == 0) sFail = 1;
#endif
    // (the other wavefronts park here while the first one finishes the sub-step before)
    __syncthreads();
    CH_MARK(1);   // barrier A
    // ---- the halo grains' state of this sub-step: re-read until both tags of a slot match
    for (int first = 0; first < nitems; first += CH_ITEMS * DEM_THREADS) {
      u32x4 v[CH_ITEMS];
      unsigned off[CH_ITEMS];
#pragma unroll
      for (int it = 0; it < CH_ITEMS; ++it) {
        const int idx = first + tid + it * DEM_THREADS;
        if (idx < nitems) {
          const unsigned ho = sHoff[idx / 5];
          off[it] = ((ho & 1u) ? remote : 0u) + pbase + (ho & ~1u) + 16u * (unsigned)(idx % 5);
          v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[it], 0, 16);   // sc1: past the L1, served by the L2
        }
      }
      unsigned spins = 0;
      for (;;) {
        bool ok = true, bad = false, waitfar = false;
#pragma unroll
        for (int it = 0; it < CH_ITEMS; ++it)
          if (first + tid + it * DEM_THREADS < nitems) {
            const bool g = (v[it].y == tag || v[it].y == stag) && (v[it].w == tag || v[it].w == stag);
            bad = bad || v[it].y == CH_POISON || v[it].w == CH_POISON;
            if (!g) {
              waitfar = waitfar || off[it] >= remote;
              v[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[it], 0, 16);
            }
            ok = ok && g;
          }
        if (__any(bad) || ++spins > CH_SPINS) { sFail = 1; break; }
#ifdef LBMDEM_CHAIN_TIMING
        ++n_spins;
#endif
        if (__all(ok)) break;
        if (__any(waitfar)) __builtin_amdgcn_s_sleep(4); else __builtin_amdgcn_s_sleep(1);   // polls through the fabric: fewer
#ifdef LBMDEM_CHAIN_TIMING
        for (int z = 0; z < ((flags >> 8) & 0xFF); ++z) __builtin_amdgcn_s_sleep(1);   // experiment: longer pauses between polls
#endif
      }
#pragma unroll
      for (int it = 0; it < CH_ITEMS; ++it) {
        const int idx = first + tid + it * DEM_THREADS;
        if (idx < nitems)
          sS[(idx % 5) * CH_SLOTS + DEM_TILE + idx / 5] =
              __longlong_as_double((long long)(((unsigned long long)v[it].z << 32) | v[it].x));
      }
    }
    CH_MARK(2);   // halo landed (this wavefront's part)
    __syncthreads();
    CH_MARK(3);   // barrier B
#ifdef LBMDEM_CHAIN_TIMING
    const long long t_b = wall_clock64();
    t_wait += t_b - t_a;
#endif
    if (sFail) break;
    // ---- contacts: phase 1 (one lane per list entry) and phase 2 (one lane per grain, list order) of k_dem_entries
    real c1 = 0., c2 = 0., c3 = 0.;
    if (mine) { c1 = sFh[tid]; c2 = sFh[DEM_TILE + tid]; c3 = sFh[2 * DEM_TILE + tid]; }
    pr = 0.0;
    int last_e = -1, slast = -1;
    real last_ft = 0., last_f3 = 0.;
    bool lost = false;
    for (int base = 0; base < E; base += CH_ENTRIES) {   // (positions among the tile's entries)
      const int lim = base + CH_ENTRIES < E ? base + CH_ENTRIES : E;
      for (int e = base + tid; e < lim; e += DEM_THREADS) {
        unsigned w;
        if (e < CH_META) w = sMeta[e];
        else { w = emeta[entry_of(e)]; if (active && !active[grain_at((int)(w & 63u))]) w = CH_SKIP; }
        if (w == CH_SKIP) continue;
        const int li = (int)(w & 63u);
        const bool lower = (w & 64u) != 0;
        const unsigned slot = w >> 8;
        GrainState a, b;
        a.x1 = sS[li]; a.x2 = sS[CH_SLOTS + li]; a.v1 = sS[2 * CH_SLOTS + li]; a.v2 = sS[3 * CH_SLOTS + li];
        a.v3 = sS[4 * CH_SLOTS + li]; a.r = sR[li];
        if (slot != DEM_CHAIN_DIRECT) {
          b.x1 = sS[slot]; b.x2 = sS[CH_SLOTS + slot]; b.v1 = sS[2 * CH_SLOTS + slot]; b.v2 = sS[3 * CH_SLOTS + slot];
          b.v3 = sS[4 * CH_SLOTS + slot]; b.r = sR[slot];
        } else {
          const int gj = nbr[entry_of(e)];
          ChainRead R{};
          // (the copy its tile is expected to write: the remote one only exists for tiles with a partner in another XCD's
          // eighth -- which a tile read from another eighth is, the list being symmetric)
          if (!chain_read_direct(rs, direct_copy(gj) + pbase + (unsigned)gj * 128u, tag, stag, R)) lost = true;
          b.x1 = R.x1; b.x2 = R.x2; b.v1 = R.v1; b.v2 = R.v2; b.v3 = R.v3; b.r = r[gj];
        }
        bool touched;
        const Force3 F = contact<false>(lower ? a : b, lower ? b : a, P, touched);  // main.c:1443-1448
        if (touched && lower && (sFlag[li] & 2)) {
          last_e = e; last_ft = F.ft; last_f3 = F.f3;
          if (T.stamp) atomicMax(&sLast, e);
        }
        const int q = e - base;
        sF1[q] = lower ? F.f1 : -F.f1;
        sF2[q] = lower ? F.f2 : -F.f2;
        sF3[q] = F.f3;
        sFn[q] = touched ? F.fn : (real)0.;
        sTouched[q] = touched ? 1 : 0;
      }
      if (lost) sFail = 1;
      CH_MARK(4);   // phase 1
      __syncthreads();
      CH_MARK(5);   // its barrier
      slast = sLast;   // (between the round's two barriers: the first wavefront resets it after the second one)
      if (mine) {
        // the entries of this grain in list order; four at a time are fetched from LDS at once and added one after the
        // other (an entry past the end is loaded from a clamped index and NOT added: x + 0 is not x for x = -0)
        const int lo = k0 > base ? k0 : base, hi = k1 < lim ? k1 : lim;
        for (int kb = lo; kb < hi; kb += 4) {
          real u1[4], u2[4], u3[4], un[4];
          unsigned char ut[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int q = (kb + j < hi ? kb + j : hi - 1) - base;
            u1[j] = sF1[q]; u2[j] = sF2[q]; u3[j] = sF3[q]; un[j] = sFn[q]; ut[j] = sTouched[q];
          }
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (kb + j < hi) {
              c1 = c1 + u1[j]; c2 = c2 + u2[j]; c3 = c3 + u3[j];
              if (ut[j]) pr += un[j];
            }
        }
      }
      __syncthreads();
      CH_MARK(6);   // phase 2 + barrier
    }
    if (sFail) break;
    WallHits wh;
    if (tid < DEM_GRAINS && mine) {   // the first wavefront holds the tile's grains
      real ds = 0., df1 = 0., dM11 = 0., dM12 = 0., dM21 = 0., dM22 = 0.;
      int dz = 0;
      walls<false>(me, wf, P, c1, c2, c3, pr, ds, df1, dz, dM11, dM12, dM21, dM22, wh);
      // main.c:1511-1515, then the second half kick main.c:1760-1762
      const real mi = sMI[tid], Iti = sMI[DEM_TILE + tid];
      a1 = c1 / mi + sG[tid];
      a2 = (c2 / mi) + sG[DEM_TILE + tid];
      a3 = c3 / Iti;
      v1 = v1 + P.dt * a1 / 2.;
      v2 = v2 + P.dt * a2 / 2.;
      v3 = v3 + P.dt * a3 / 2.;
    }
    CH_MARK(7);   // walls, accelerations, second half kick
    // the next sub-step's state goes out before this one's book-keeping: the partners' tiles are waiting for it
    const int last_own = last_e >= 0 && last_e == slast ? last_e : -1;
    if (s + 1 < nsteps) drift_publish(stamp + 1);
    CH_MARK(0);   // drift + publish
    if (T.stamp && last_own >= 0) {   // the tile's last grain contact, for the carries
      const long rec = (long)tile * 4 + CARRY_GRAIN;
      T.stamp[rec] = stamp; T.val[2 * rec] = last_ft; T.val[2 * rec + 1] = last_f3;
      const int le = entry_of(last_own);
      T.who[rec] = ((long long)grain_at((int)(((last_own < CH_META) ? sMeta[last_own] : emeta[le]) & 63u)) << 32) | (unsigned)nbr[le];
    }
    if (T.stamp && tid < DEM_GRAINS) {   // per wall, the highest grain of the tile that touched it
#pragma unroll
      for (int kind = CARRY_BOTTOM; kind <= CARRY_RIGHT; ++kind) {
        const unsigned bit = kind == CARRY_BOTTOM ? 1u : (kind == CARRY_LEFT ? 4u : 8u);
        const unsigned long long hit = __ballot((wh.mask & bit) != 0 && (sFlag[tid] & 2));
        if (hit != 0 && tid == 63 - __builtin_clzll(hit)) {
          const long rec = (long)tile * 4 + kind;
          T.stamp[rec] = stamp;
          T.who[rec] = (long long)i << 32;
          T.val[2 * rec] = kind == CARRY_BOTTOM ? wh.ftB : (kind == CARRY_LEFT ? wh.ftL : wh.ftR);
          T.val[2 * rec + 1] = wh.f3B;
        }
      }
    }
#ifdef LBMDEM_CHAIN_TIMING
    t_work += wall_clock64() - t_b;
#endif
  }
#ifdef LBMDEM_CHAIN_TIMING
  if (dbg && tid == 0) {
    long long* d = dbg + (long)tile * 16;
    for (int k = 0; k < 8; ++k) d[8 + k] = ph[k];
    d[0] = t_wait; d[1] = t_work; d[2] = n_spins; d[3] = placed ? 1 : 0; d[4] = hcnt; d[5] = E;
    d[6] = t_begin; d[7] = wall_clock64();
    int nfar = 0;
    for (int h = 0; h < hcnt; ++h) nfar += (int)(sHoff[h] & 1u);
    unsigned hwid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    d[3] |= (long long)nfar << 8 | (long long)(xcc & 0xFu) << 32 | (long long)hwid << 36;
  }
#endif
  if (sFail) {   // (uniform: read after a barrier) -- let the partners know, and the host
    if (have) {
      for (unsigned par = 0; par < 2; ++par) {
        chain_publish<0>(rs, par * line_par + (unsigned)i * 128u, CH_POISON, 0., 0., 0., 0., 0.);
        chain_publish<16>(rs, remote + par * line_par + (unsigned)i * 128u, CH_POISON, 0., 0., 0., 0., 0.);
      }
    }
    if (tid == 0) {
      // the launch's first sequence number (the host's key to the state before the launch), and the stop word: every
      // kernel queued behind this launch returns at once
      *err = (int)((unsigned long long)stamp0 & 0x3FFFFFFFull) + 1;
      if (gate) __hip_atomic_store(gate, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  if (paint.obst) {
    // ---- obst_construction's rasterisation (main.c:1009-1032) of this tile's discs at the positions the run ends with:
    // they are in LDS (the drift of the last sub-step), and so are those of every partner. Four lanes per grain.
    // The stand-alone rasteriser spends its time on its 11 M scattered 4-byte stores (24 of 28 us here too, measured with
    // the stores taken out): with `paint.was` the canvas already holds the picture of two fluid steps ago and only the
    // nodes whose owner changes are written (k_obst_update's rules, lbm_obst.hip) -- the tests cost 4 us.
    __syncthreads();
    if (tid < DEM_GRAINS) {   // the velocities of the fluid-side record are the final ones
      sS[2 * CH_SLOTS + tid] = v1; sS[3 * CH_SLOTS + tid] = v2; sS[4 * CH_SLOTS + tid] = v3;
    }
    __syncthreads();
    const LatticeView& L = paint.L;
    const int g = tid >> 2, hl = tid & 3, gi = grain_at(g);
    const bool there = gi >= 0;
    const bool inplace = paint.was.xc != nullptr;
    const long long stamp_l = stamp0 + nsteps - 1;
    const real gx1 = sS[g], gx2 = sS[CH_SLOTS + g];
    const real rl = there ? paint.rLB[gi] : (real)1.;
    const real xc = (gx1 - L.Mgx) / L.dx, yc = (gx2 - L.Mby) / L.dx, r2 = rl * rl, rbl0 = sR[g] / L.dx;   // main.c:1009-1013
    bool had = false;
    real pxc = 0., pyc = 0., still2 = 0.;
    if (there && inplace) { had = paint.was.mode[gi] != 0; pxc = paint.was.xc[gi]; pyc = paint.was.yc[gi]; still2 = paint.was.still2[gi]; }
    // ... and the picture in the other map buffer (the previous fluid step's): where does this one differ from it?
    const bool chg = inplace && paint.chg.bits != nullptr;
    bool ohad = false;
    real oxc = 0., oyc = 0., ostill2 = 0.;
    if (there && chg) { ohad = paint.other.mode[gi] != 0; oxc = paint.other.xc[gi]; oyc = paint.other.yc[gi]; ostill2 = paint.other.still2[gi]; }
    // bits of the rows x0..x1 in the windows that hold columns y0..y1 (window w = columns [w ww - off, w ww - off + 63])
    auto flag_rows = [&](int x0, int x1, int y0, int y1, int lane_of, int lanes) {
      const ObstChange& C = paint.chg;
      int w0 = (y0 + C.off - 63 + C.ww - 1) / C.ww, w1 = (y1 + C.off) / C.ww;   // (y0 >= 0: the boxes are clamped to the lattice)
      if (y0 + C.off - 63 < 0) w0 = 0;
      if (w1 > paint.windows - 1) w1 = paint.windows - 1;
      if (x0 < L.gx0) x0 = L.gx0;
      if (x1 > L.gx0 + L.nxl - 1) x1 = L.gx0 + L.nxl - 1;
      for (int x = x0 + lane_of; x <= x1; x += lanes) {
        const int xl = x - L.gx0;
        for (int w = w0; w <= w1; ++w) atomicOr(&C.bits[(long)w * C.words + (xl >> 5)], 1u << (xl & 31));
      }
    };
    if (there && hl == 0) {
      paint.xc[gi] = xc; paint.yc[gi] = yc; paint.r2[gi] = r2; paint.rbl0[gi] = rbl0;
      real* o = paint.pk + (long)gi * 8;
      o[0] = gx1; o[1] = gx2; o[2] = sS[2 * CH_SLOTS + g]; o[3] = sS[3 * CH_SLOTS + g]; o[4] = sS[4 * CH_SLOTS + g];
      o[5] = xc; o[6] = yc; o[7] = r2;
      if (inplace) {   // has a grain outrun the pair list? (k_obst_update)
        const real mx = gx1 - paint.xreb[gi], my = gx2 - paint.yreb[gi];
        if (!(mx * mx + my * my <= paint.moved_limit * paint.moved_limit)) *paint.moved_flag = paint.list_generation;
      }
    }
    // the partners' final positions: staged slot, or (beyond the staged halo) the published line
    auto partner_pos = [&](int k, real& jx, real& jy) {
      const int pk = sPre[g] + (k - sK[g]);   // (k: one of THIS grain's list entries)
      const unsigned w = (pk < CH_META) ? sMeta[pk] : emeta[k];
      const unsigned slot = w >> 8;
      if (slot != DEM_CHAIN_DIRECT) { jx = sS[slot]; jy = sS[CH_SLOTS + slot]; return true; }
      const unsigned off = direct_copy(nbr[k]) + ((stamp_l & 1) ? line_par : 0u) + (unsigned)nbr[k] * 128u;
      return chain_read_slot(rs, off, chain_tag(stamp_l), stag, jx) && chain_read_slot(rs, off + 16u, chain_tag(stamp_l), stag, jy);
    };
    // alone: clear of the discs of all partners -- 1.5 nodes for a fresh picture (k_obst_paint's test); in place: this disc
    // has moved less than half a node since it was painted and the circles keep 1.1 nodes apart (k_obst_update's)
    const real moved2 = (xc - pxc) * (xc - pxc) + (yc - pyc) * (yc - pyc);
    bool near = inplace && had && !(moved2 < 0.25);
    if (there) {
      for (int k = sK[g] + hl; k < sK[DEM_TILE + g]; k += 4) {
        real jx = 0., jy = 0.;
        if (!partner_pos(k, jx, jy)) near = true;
        const real ddx = jx - gx1, ddy = jy - gx2, rr = (rl + paint.rLB[nbr[k]] + (inplace ? 1.1 : 1.5)) * L.dx;
        near |= !(ddx * ddx + ddy * ddy >= rr * rr);   // also true for a NaN
      }
    }
    const bool alone = ((__ballot(near) >> (4 * ((tid & 63) >> 2))) & 0xFull) == 0;   // this grain's four lanes
#ifdef LBMDEM_AB
    const bool cut = (flags >> 16) & 1;   // experiment: the rasterisation without its node loops
#else
    const bool cut = false;
#endif
    // Discs that need their nodes looked at are listed in LDS and their lattice rows dealt out to ALL lanes of the
    // workgroup afterwards: with the rows of a grain on its own four lanes, every wavefront walked through every kind
    // of scan while half its lanes (the discs that have not moved enough to change a node) sat idle: 21 us; dealt out: half.
    // (the staging arrays of phase 1 are free now; typed views, the same in the float build)
    real* const recA = sF1;   // [slot][4] xc, yc, rl, rbl0 of a listed disc
    real* const recP = sF2;   // [slot][4] the centre it was painted at in this buffer, in the other buffer
    unsigned long long* const recG = reinterpret_cast<unsigned long long*>(sF3);   // [slot] smallest gap found, as the bits of a double
    long long* const recI = reinterpret_cast<long long*>(sFn);                      // [slot] gi | kind << 40 | had << 44
    static_assert(sizeof(real) * CH_ENTRIES >= 8 * DEM_TILE && CH_ENTRIES >= 4 * DEM_TILE, "LDS views of the rasterisation");
    int* const scan_count = &sScan[0];
    int* const scan_rows = &sScan[1];
    if (tid == 0) { *scan_count = 0; *scan_rows = 0; }
    __syncthreads();
    if (there && !cut) {
      const DiscGeo gn = disc_geo(L, xc, yc, rl, rbl0, true), go = disc_geo(L, pxc, pyc, rl, rbl0, had);
      const bool still = inplace && alone && had && moved2 < still2;
      const bool ring = alone && inplace && had && gn.r2 <= gn.R2;
#ifdef LBMDEM_AB   /* experiment build: which way the grains of the rasterisation go (dbg + tiles * 16: 4 counters) */
      if (dbg && hl == 0)
        atomicAdd(reinterpret_cast<unsigned long long*>(dbg + (long)tiles * 16 + (still ? 0 : (ring ? 1 : (alone ? 2 : 3)))), 1ull);
#endif
      // Against the other buffer's picture: exact (its disc test beside the two others, in the ring scan) when this disc is
      // alone now and has moved less than half a node from BOTH painted centres -- the pictures of two discs that keep 1.1
      // nodes apart now then share no node in either map; a partner that fails this flags every row of its own boxes.
      const real omoved2 = (xc - oxc) * (xc - oxc) + (yc - oyc) * (yc - oyc);
      const DiscGeo gq = disc_geo(L, oxc, oyc, rl, rbl0, ohad);
      const bool oexact = chg && alone && had && ohad && gn.any && gn.r2 <= gn.R2 && omoved2 < 0.25;
      const bool ostill = oexact && omoved2 < ostill2;
      if (chg && !oexact) {   // every row any of the three pictures of this disc reaches, and one around
        int x0 = gn.any ? gn.xi : (1 << 30), x1 = gn.any ? gn.xf : -1, y0 = gn.any ? gn.yi : (1 << 30), y1 = gn.any ? gn.yf : -1;
        if (go.any) { x0 = min(x0, go.xi); x1 = max(x1, go.xf); y0 = min(y0, go.yi); y1 = max(y1, go.yf); }
        if (gq.any) { x0 = min(x0, gq.xi); x1 = max(x1, gq.xf); y0 = min(y0, gq.yi); y1 = max(y1, gq.yf); }
        if (x1 >= x0) flag_rows(x0 - 1, x1 + 1, y0 > 0 ? y0 - 1 : 0, y1 + 1, hl, 4);
      }
      if (still) {
        // nothing can have changed sides: map and record stay (the next comparison is again with the painted centre)
        if (hl == 0) { paint.now.xc[gi] = pxc; paint.now.yc[gi] = pyc; paint.now.still2[gi] = still2; paint.now.mode[gi] = 1; }
        if (oexact && !ostill && hl == 0) {
          // ... in THIS buffer; against the other one the ring is scanned for the bits alone (no node is written: the two
          // tests of this buffer agree on every node)
          const real w = (real)sqrt((double)omoved2) + (sizeof(real) == 4 ? (real)(1e-3 + 5e-7 * (fabs((double)xc) + fabs((double)yc))) : (real)1e-6);
          const int xlo = (int)floor(xc - (rl + w)) - 1, xhi = (int)ceil(xc + (rl + w)) + 1;
          const int slot = atomicAdd(scan_count, 1);
          atomicMax(scan_rows, xhi - xlo + 1);
          recA[slot * 4] = xc; recA[slot * 4 + 1] = yc; recA[slot * 4 + 2] = rl; recA[slot * 4 + 3] = rbl0;
          recP[slot * 4] = pxc; recP[slot * 4 + 1] = pyc; recP[slot * 4 + 2] = oxc; recP[slot * 4 + 3] = oyc;
          recG[slot] = (unsigned long long)__double_as_longlong((double)w);
          recI[slot] = (long long)gi | (1ll << 40) | (1ll << 44) | ((long long)(ohad ? 1 : 0) << 45) | (1ll << 46);
        }
      } else {
        if (hl == 0) { paint.now.xc[gi] = xc; paint.now.yc[gi] = yc; paint.now.still2[gi] = 0.; paint.now.mode[gi] = 1; }
        if (alone && (gn.any || go.any)) {
          if (hl == 0) {
            // Ring: the disc has moved by |D| (< 1/2 node), so a node at distance d from the new centre was at d -+ |D| from
            // the old one: only nodes with |d - r| <= |D| can have changed sides -- an annulus a few hundredths of a node
            // wide that holds a handful of nodes, instead of the (2 r + 3)^2 of the box. (A node inside its inner circle
            // lies in both discs and both boxes, one outside its outer circle in neither disc: the boxes cannot change
            // that.) Box: the union of the two boxes and a node around it (a first picture, or a disc whose box cuts into it).
            // w = |D| + what the rounding of the test's d2 can amount to, in nodes.
            const bool ocmp = ring && oexact && !ostill;   // the other buffer's test rides in the same scan
            const real mv2 = ocmp && omoved2 > moved2 ? omoved2 : moved2;
            const real w = (real)sqrt((double)mv2) + (sizeof(real) == 4 ? (real)(1e-3 + 5e-7 * (fabs((double)xc) + fabs((double)yc))) : (real)1e-6);   // (float: coordinates of a few thousand carry 1e-4 ... 1e-3 themselves)
            int xlo, xhi;
            if (ring) { xlo = (int)floor(xc - (rl + w)) - 1; xhi = (int)ceil(xc + (rl + w)) + 1; }
            else {
              xlo = (!go.any ? gn.xi : (!gn.any ? go.xi : (go.xi < gn.xi ? go.xi : gn.xi))) - 1;
              xhi = (!go.any ? gn.xf : (!gn.any ? go.xf : (go.xf > gn.xf ? go.xf : gn.xf))) + 1;
            }
            const int slot = atomicAdd(scan_count, 1);
            atomicMax(scan_rows, xhi - xlo + 1);
            recA[slot * 4] = xc; recA[slot * 4 + 1] = yc; recA[slot * 4 + 2] = rl; recA[slot * 4 + 3] = rbl0;
            recP[slot * 4] = pxc; recP[slot * 4 + 1] = pyc; recP[slot * 4 + 2] = oxc; recP[slot * 4 + 3] = oyc;
            recG[slot] = (unsigned long long)__double_as_longlong(ring ? (double)w : 1e30);   // ring: its half-width; box: the smallest gap found
            // bit 45: the other buffer holds a picture of this disc, 46: compare with it (else the rows are flagged already,
            // or -- a ring whose other picture has not moved -- need no bits), 47: bits where THIS buffer's node changes
            recI[slot] = (long long)gi | ((long long)(ring ? 1 : 2) << 40) | ((long long)(had ? 1 : 0) << 44) |
                         ((long long)(ohad ? 1 : 0) << 45) | ((long long)(ocmp ? 1 : 0) << 46);
          }
        } else if (!alone) {
          auto partner_geo = [&](int k, bool& ok) {
            real jx = 0., jy = 0.;
            ok = partner_pos(k, jx, jy);
            const int j = nbr[k];
            return disc_geo(L, (jx - L.Mgx) / L.dx, (jy - L.Mby) / L.dx, paint.rLB[j], paint.r[j] / L.dx, ok);
          };
          const int k0g = sK[g], k1g = sK[DEM_TILE + g];
          if (go.any) {   // the nodes this disc has left: to the highest partner that covers them now, else to the fluid
            const int ny = go.yf - go.yi + 1, total = (go.xf - go.xi + 1) * ny;
            for (int k = hl; k < total; k += 4) {
              const int x = go.xi + k / ny, y = go.yi + k % ny;
              if (!disc_has(go, x, y) || disc_has(gn, x, y)) continue;
              int v = -1;
              for (int e = k0g; e < k1g; ++e) {
                bool ok;
                const DiscGeo gj = partner_geo(e, ok);
                if (nbr[e] > v && disc_has(gj, x, y)) v = nbr[e];
              }
              atomicCAS(&paint.obst[(long)(x - L.gx0) * L.sy + y], gi, v);
            }
          }
          if (gn.any) {   // the nodes it covers: highest index wins (main.c:1028); who else covers them, from the partners' discs
            const int ny = gn.yf - gn.yi + 1, total = (gn.xf - gn.xi + 1) * ny;
            for (int k = hl; k < total; k += 4) {
              const int x = gn.xi + k / ny, y = gn.yi + k % ny;
              if (disc_has(gn, x, y)) atomicMax(&paint.obst[(long)(x - L.gx0) * L.sy + y], gi);
            }
            for (int e = k0g; e < k1g; ++e) {
              bool ok;
              const DiscGeo gj = partner_geo(e, ok);
              const int j = nbr[e];
              if (!gj.any || gj.xi > gn.xf || gj.xf < gn.xi || gj.yi > gn.yf || gj.yf < gn.yi) continue;
              for (int k = hl; k < total; k += 4) {
                const int x = gn.xi + k / ny, y = gn.yi + k % ny;
                if (!disc_has(gn, x, y) || !disc_has(gj, x, y)) continue;
                const long node = (long)(x - L.gx0) * L.sy + y;
                paint.touched[gi] = 1; paint.touched[j] = 1;
                if (paint.mincov) {
                  atomicMax(&paint.mincov[node], (paint.epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)gi));
                  atomicMax(&paint.mincov[node], (paint.epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)j));
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();
    {   // ---- the listed discs' rows, one (disc, row) per lane and pass
      const int nscan = *scan_count, nr = *scan_rows;
      for (int t = tid; t < nscan * nr; t += DEM_THREADS) {
        const int slot = t / nr, row = t - slot * nr;
        const real cx = recA[slot * 4], cy = recA[slot * 4 + 1], crl = recA[slot * 4 + 2], crb = recA[slot * 4 + 3];
        const long long bits = recI[slot];
        const int cgi = (int)(bits & 0xFFFFFFFFFFll), kind = (int)((bits >> 40) & 15);
        const bool chad = ((bits >> 44) & 1) != 0;
        const DiscGeo gn = disc_geo(L, cx, cy, crl, crb, true), go = disc_geo(L, recP[slot * 4], recP[slot * 4 + 1], crl, crb, chad);
        const bool ocmp = ((bits >> 46) & 1) != 0;
        const DiscGeo gq = disc_geo(L, recP[slot * 4 + 2], recP[slot * 4 + 3], crl, crb, ((bits >> 45) & 1) != 0);
        const real rm2 = gn.r2 < gn.R2 ? gn.r2 : gn.R2, near2 = (crl + 1.) * (crl + 1.);
        real gap = 1e30;
        auto settle = [&](int x, int y) {   // the reference's own test at both centres (disc_has: box and d2 <= r2, main.c:1027)
          const bool bo = disc_has(go, x, y), bn = disc_has(gn, x, y);
          if (bn != bo) paint.obst[(long)(x - L.gx0) * L.sy + y] = bn ? cgi : -1;
          if (ocmp && disc_has(gq, x, y) != bn) {   // the other map's owner of this node is not this picture's
            const ObstChange& C = paint.chg;
            const int xl = x - L.gx0, w0 = (y + C.off - 63 + C.ww - 1) / C.ww, w1 = (y + C.off) / C.ww;
            for (int w = (y + C.off - 63 < 0 ? 0 : w0); w <= w1 && w < paint.windows; ++w)
              atomicOr(&C.bits[(long)w * C.words + (xl >> 5)], 1u << (xl & 31));
          }
          const real d2 = (x - cx) * (x - cx) + (y - cy) * (y - cy), gg = d2 > rm2 ? d2 - rm2 : rm2 - d2;
          if (d2 <= near2) gap = gg < gap ? gg : gap;   // (nodes farther out than r + 1 stay outside whatever happens within half a node)
        };
        if (kind == 1) {
          // per lattice row two square roots give the annulus' two stretches (one near the poles); the stretches are
          // widened by w once more, which covers the roots' own rounding: most of them hold no node at all
          const real w = (real)__longlong_as_double((long long)recG[slot]);
          const real ro = crl + w, rin = crl > w ? crl - w : 0.;
          const int x = (int)floor(cx - ro) - 1 + row;
          const real dxn = x - cx, o2 = ro * ro - dxn * dxn;
          if (x <= (int)ceil(cx + ro) + 1 && o2 >= 0.) {
            const real yo = sqrt(o2), i2 = rin * rin - dxn * dxn, yn = i2 > 0. ? sqrt(i2) : 0.;
            const int a0 = (int)ceil(cy - yo - w), a1 = (int)floor(cy - yn + w), b0 = (int)ceil(cy + yn - w), b1 = (int)floor(cy + yo + w);
            if (a1 >= b0) { for (int y = a0; y <= b1; ++y) settle(x, y); }
            else { for (int y = a0; y <= a1; ++y) settle(x, y); for (int y = b0; y <= b1; ++y) settle(x, y); }
          }
        } else {
          const int xi = (!go.any ? gn.xi : (!gn.any ? go.xi : (go.xi < gn.xi ? go.xi : gn.xi))) - 1;
          const int xf = (!go.any ? gn.xf : (!gn.any ? go.xf : (go.xf > gn.xf ? go.xf : gn.xf))) + 1;
          const int yi = (!go.any ? gn.yi : (!gn.any ? go.yi : (go.yi < gn.yi ? go.yi : gn.yi))) - 1;
          const int yf = (!go.any ? gn.yf : (!gn.any ? go.yf : (go.yf > gn.yf ? go.yf : gn.yf))) + 1;
          const int x = xi + row;
          if (x <= xf) for (int y = yi; y <= yf; ++y) settle(x, y);
        }
        // box scans: how far the nearest node is from changing sides (positive doubles order like their bit patterns)
        if (kind != 1) atomicMin(&recG[slot], (unsigned long long)__double_as_longlong((double)gap));
      }
      __syncthreads();
      // A node within r + 1 of the centre sees its d2 change by at most (2 (r + 1) + |D|) |D| < (2 r + 3) |D| when the centre moves
      // by |D| < 1/2, a node farther out stays farther than r + 1/2: nothing changes sides while |D| < gap / (2 r + 3).
      if (tid < nscan) {
        const real cx = recA[tid * 4], cy = recA[tid * 4 + 1], crl = recA[tid * 4 + 2], crb = recA[tid * 4 + 3];
        const int cgi = (int)(recI[tid] & 0xFFFFFFFFFFll);
        const DiscGeo gn = disc_geo(L, cx, cy, crl, crb, true);
        if (((recI[tid] >> 40) & 15) != 1 && gn.any && gn.xi > 1 && gn.xf < L.lx - 2 && gn.yi > 1 && gn.yf < L.ly - 2 && gn.xi > L.gx0 && gn.xf < L.gx0 + L.nxl - 1) {
          // (the slack: d2 as the test computes it carries rounding errors of a few ulps of ~100 -- 1e-13 in double, 1e-5 in the float build)
          const real gap = (real)__longlong_as_double((long long)recG[tid]), lim = (gap - (sizeof(real) == 4 ? 1e-3 : 1e-9)) / (2. * crl + 3.);
          paint.now.still2[cgi] = (gn.r2 <= gn.R2 && lim > 0.) ? (lim < 0.45 ? lim * lim : 0.2025) : 0.;
        }
      }
    }
  }
  if (have) {   // grains this rank does not integrate keep their state, in the buffer that is current from now on
    out.x1[i] = x1; out.x2[i] = x2; out.x3[i] = sX3[tid];
    out.v1[i] = v1; out.v2[i] = v2; out.v3[i] = v3;
    out.a1[i] = a1; out.a2[i] = a2; out.a3[i] = a3;
    if (mine) pout[i] = pr;
  }
}

// ---------------------------------------------------------------------------------------------
// Order-dependent contact diagnostics: fr, ice, slip, rw
// ---------------------------------------------------------------------------------------------
//
// The reference updates four file-scope "previous contact" variables while it walks the contacts serially
// (pft, pff, pf, ic: main.c:130-131) and each contact's contribution to slip / rw / fr reads them:
//     slip_i += |ft| (|vt dt| + |ft - pft| / kt);  pft = ft        (force_grains, main.c:782-786)
// so a contribution depends on the contact evaluated just before it -- of whatever grain -- and the
// chain runs on through the four wall loops and into the next sub-step. Reproduced in three steps:
//   k_diag_scan   the sub-step kernel has left (touched, ft, f3, |vt dt|, |v3 dt|) per list entry; the
//                 entries with own < partner, in list order, ARE the reference's contact order. One
//                 workgroup turns "value of the previous touched entry" into a scan (chunk, last-defined
//                 across chunks, chunk again) and stores every entry's contribution;
//   k_diag_accum  one thread per grain adds its contributions in list order (the film law credits both
//                 partners, main.c:1401-1408: the upper one finds the pair's entry by bisection);
//   k_diag_walls  the four wall loops (main.c:1455-1508) replayed by ONE thread in the reference's order
//                 with the carries, `ic`, and the reference's own indexing of g[] by the list POSITION in
//                 the two `fr` updates of acceleration_grains (main.c:1462-1465, 1490-1493) -- a few hundred
//                 wall candidates, once per 4000 sub-steps.
// carry[] = {pft, pff, pf} persists on the device; between diagnostic sub-steps the ordinary sub-step kernel keeps
// per-tile records of its last contacts (CarryTrack, lbmdem_internal.h) from which k_carry_resolve rebuilds the
// carries exactly before the next diagnostic sub-step.

__global__ __launch_bounds__(1024) void k_diag_scan(DiagExtra X, const int* __restrict__ offsets, int n, real kt) {
  LBMDEM_GATE(X.gate);
  __shared__ real sFt[1024], sF3[1024];
  __shared__ unsigned char sHas[1024];
  const int t = threadIdx.x;
  const int E = offsets[n];
  const int chunk = (E + 1023) / 1024;
  const int b = t * chunk, e = b + chunk < E ? b + chunk : E;
  bool has = false;
  real lft = 0., lf3 = 0.;
  for (int k = b; k < e; ++k)
    if (X.e_touched[k]) { has = true; lft = X.e_ft[k]; lf3 = X.e_f3[k]; }
  sHas[t] = has; sFt[t] = lft; sF3[t] = lf3;
  __syncthreads();
  real pft = X.carry[0], pff = X.carry[1];
  for (int u = t - 1; u >= 0; --u)
    if (sHas[u]) { pft = sFt[u]; pff = sF3[u]; break; }
  for (int k = b; k < e; ++k) {
    if (!X.e_touched[k]) continue;
    const real ft = X.e_ft[k], f3 = X.e_f3[k];
    X.e_dslip[k] = fabs(ft) * (X.e_avt[k] + (fabs(ft - pft)) / kt);
    X.e_drw[k] = fabs(f3) * (X.e_av3[k] + (fabs(f3 - pff)) / kt);
    pft = ft; pff = f3;
  }
  __syncthreads();
  if (t == 0) {
    for (int u = 1023; u >= 0; --u)
      if (sHas[u]) { X.carry[0] = sFt[u]; X.carry[1] = sF3[u]; break; }
  }
}

template <bool FILM>
__global__ void k_diag_accum(DiagExtra X, const int* __restrict__ offsets, const int* __restrict__ nbr, int n) {
  LBMDEM_GATE(X.gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  real slip = 0., rw = 0.;
  for (int k = offsets[i]; k < offsets[i + 1]; ++k) {
    const int j = nbr[k];
    int e = k;
    if (j < i) {  // the pair's entry lives in j's list
      if (!FILM) continue;   // force_grains credits the lower grain only
      int lo = offsets[j], hi = offsets[j + 1] - 1;
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (nbr[mid] < i) lo = mid + 1; else hi = mid;
      }
      e = lo;
    }
    if (X.e_touched[e]) { slip += X.e_dslip[e]; rw += X.e_drw[e]; }
  }
  X.slip[i] = slip; X.rw[i] = rw; X.fr[i] = 0.; X.ice[i] = 0.;
}

// wall candidate lists in grain order (VerletWall, main.c:1563-1593): wave w builds list w
__global__ __launch_bounds__(256) void k_diag_wall_lists(DiagExtra X, const unsigned char* __restrict__ wallflags, int n) {
  LBMDEM_GATE(X.gate);
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  int cnt = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    const bool f = i < n && ((wallflags[i] >> w) & 1u);
    const unsigned long long b = __ballot(f);
    if (f) X.wlist[(size_t)w * n + cnt + __popcll(b & lt)] = i;
    cnt += __popcll(b);
  }
  if (lane == 0) X.wcount[w] = cnt;
}

__global__ void k_diag_walls(DiagExtra X, Kin in, const real* __restrict__ r, DemParams P) {
  LBMDEM_GATE(P.gate);
  if (blockIdx.x != 0 || threadIdx.x != 0) return;
  const int n = P.n;
  real pft = X.carry[0], pff = X.carry[1], pf = X.carry[2], ic = 0.0;  // ic: reset every sub-step (main.c:1743)
  const int *LB = X.wlist, *LT = X.wlist + n, *LL = X.wlist + 2 * (size_t)n, *LR = X.wlist + 3 * (size_t)n;
  // g[k].v1 and g[k].a1 of the reference at this point: drifted velocity, force sum so far (k = list position)
  auto v1_of = [&](int g) { return in.v1[g] + P.dt * in.a1[g] / 2.; };
  for (int k = 0; k < X.wcount[0]; ++k) {  // bottom: main.c:1455-1468, force_WallB 809-845
    const int i = LB[k];
    const GrainState me = advance(in, r, i, P);
    const real dn = me.x2 - me.r - P.Mby;
    if (!(dn < 0)) continue;
    const real vn = me.v2, vt = me.v1;
    real fn = -P.km * dn - P.num * vn;
    if (fn < 0) fn = 0.;
    real ft = P.ktm * vt;
    const real ftest = P.mumb * fn;
    if (fabs(ft) > ftest) ft = (ft < 0.0) ? ftest : -ftest;
    const real f1 = ft, f3 = -(ft * me.r * P.murf);
    X.rw[i] += fabs(f3) * (fabs(me.v3 * P.dt) + (fabs(f3 - pff)) / P.kt);
    X.fr[i] += fabs(ft) * (fabs(vt * P.dt) + (fabs(ft - pft)) / P.kt);
    pff = f3; pft = ft;
    X.a1gc[i] = X.a1gc[i] + f1;
    X.fr[i] += fabs(f1) * (fabs(P.dt * v1_of(k)) + fabs(P.dt2 * X.a1gc[k]) + (fabs(f1 - pf)) / P.kt);
    pf = f1;
  }
  for (int k = 0; k < X.wcount[1]; ++k) {  // top: main.c:1470-1480, force_WallT 846-887
    const int i = LT[k];
    const GrainState me = advance(in, r, i, P);
    const real dn = -me.x2 - me.r + P.Mhy;
    if (!(dn < 0)) continue;
    const real vn = me.v2;
    real fn = P.km * dn - P.num * vn;
    ic += P.num * vn * vn * P.dt;
    if (fn > 0.) fn = 0.;
    const real vt = me.v1 + me.v3 * me.r - P.wallT_vel;
    real ft = fabs(P.ktm * vt);
    real ftmax;
    if (vt >= 0) ftmax = P.mumb * fn - P.nugt * vt; else ftmax = P.mumb * fn + P.nugt * vt;
    if (ft > ftmax) ft = ftmax;
    if (vt > 0) ft = -ft;
    X.a1gc[i] = X.a1gc[i] + ft;
  }
  for (int k = 0; k < X.wcount[2]; ++k) {  // left: main.c:1482-1496, force_WallL 888-921
    const int i = LL[k];
    const GrainState me = advance(in, r, i, P);
    const real dn = me.x1 - me.r - P.Mgx;
    if (!(dn < 0)) continue;
    const real vn = me.v1;
    real fn = -P.km * dn + P.num * vn;
    ic += P.num * vn * vn * P.dt;
    if (fn < 0.) fn = 0.;
    const real vt = me.v2;
    real ft = P.mum * fn;
    if (vt > 0) ft = -ft;
    const real f1 = fn, f2 = ft, f3 = ft * me.r * P.murf;
    X.ice[i] += ic;
    X.rw[i] += fabs(f3) * fabs(me.v3 * P.dt);
    X.fr[i] += fabs(ft) * (fabs(vt * P.dt) + (fabs(ft - pft)) / P.kt);
    pft = ft;
    X.a1gc[i] = X.a1gc[i] + f1;
    X.fr[i] += fabs(f2) * (fabs(P.dt * v1_of(k)) + fabs(P.dt2 * X.a1gc[k]) + (fabs(f2 - pf)) / P.kt);
    pf = f2;
  }
  for (int k = 0; k < X.wcount[3]; ++k) {  // right: main.c:1498-1508, force_WallR 923-951
    const int i = LR[k];
    const GrainState me = advance(in, r, i, P);
    const real dn = -me.x1 - me.r + P.Mdx;
    if (!(dn < 0)) continue;
    const real vn = me.v1;
    const real fn = P.km * dn - P.num * vn;
    const real vt = me.v2;
    real ft = P.mum * fn;
    if (vt > 0) ft = -ft;
    pft = ft;
    X.a1gc[i] = X.a1gc[i] + ((fn > 0.) ? 0. : fn);
  }
  X.carry[0] = pft; X.carry[1] = pff; X.carry[2] = pf;
}

// The carries as the reference holds them now: per carry the youngest record among the kinds that assign it, in
// program order: sub-step, then grain contacts < bottom < left < right wall, then the grain the record is about (every tile
// leaves the LAST of its contacts in the reference's order, i.e. of its highest grain that has one: the youngest of all is the
// record with the highest grain -- whatever grains the tiles are made of, and whichever kernel's tiling wrote the record).
__global__ __launch_bounds__(256) void k_carry_resolve(CarryTrack T, long long min_stamp) {
  LBMDEM_GATE(T.gate);
  __shared__ unsigned long long best[3];
  if (threadIdx.x < 3) best[threadIdx.x] = 0ull;
  __syncthreads();
  // 0 = nothing; grains < 2^24 (dem_chain_alloc), kinds 2 bits, the rest for the sub-step
  auto key_of = [&](long rec, int kind) -> unsigned long long {
    const long long st = T.stamp[rec];
    if (st < min_stamp) return 0ull;
    return ((unsigned long long)(st + 1) << 26) | ((unsigned long long)kind << 24) | ((unsigned long long)(T.who[rec] >> 32) & 0xFFFFFFull);
  };
  unsigned long long b_ft = 0ull, b_ff = 0ull, b_f = 0ull;
  for (int t = threadIdx.x; t < T.tiles; t += 256) {
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) {
      const unsigned long long key = key_of((long)t * 4 + kind, kind);
      if (key > b_ft) b_ft = key;                                                    // pft: all four
      if ((kind == CARRY_GRAIN || kind == CARRY_BOTTOM) && key > b_ff) b_ff = key;   // pff
      if ((kind == CARRY_BOTTOM || kind == CARRY_LEFT) && key > b_f) b_f = key;      // pf
    }
  }
  if (b_ft) atomicMax(&best[0], b_ft);
  if (b_ff) atomicMax(&best[1], b_ff);
  if (b_f) atomicMax(&best[2], b_f);
  __syncthreads();
  if (threadIdx.x < 3) { T.best_key[2 * threadIdx.x] = 0; T.best_key[2 * threadIdx.x + 1] = 0; }
  __syncthreads();
  // what won, in rank-independent terms, for the cross-rank resolve of a strip decomposition:
  // best_key[c] = {(stamp + 1) * 4 + kind, (grain << 32) | partner}; {0, 0} = this handle has no record
  for (int t = threadIdx.x; t < T.tiles; t += 256) {
#pragma unroll
    for (int kind = 0; kind < 4; ++kind) {
      const long rec = (long)t * 4 + kind;
      const unsigned long long key = key_of(rec, kind);
      if (key == 0ull) continue;
#pragma unroll
      for (int c = 0; c < 3; ++c)
        if (key == best[c]) {   // (a key names one record: a grain is in one tile)
          T.carry[c] = T.val[2 * rec + (c == 1 ? 1 : 0)];   // pft, pf: ft; pff: f3
          T.best_key[2 * c] = (long long)((key >> 26) * 4 + ((key >> 24) & 3u));
          T.best_key[2 * c + 1] = T.who[rec];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Verlet list
// ---------------------------------------------------------------------------------------------

__device__ __forceinline__ int cell_coord(real x, real o, real cs, int nc) {
  int c = (int)floor((x - o) / cs);
  return c < 0 ? 0 : (c >= nc ? nc - 1 : c);
}

// Grains by cell, as a counting sort = a radix sort with the whole cell index as its one digit (round 4; before: hipCUB radix sort of (cell, grain) pairs = 7 launches at this size,
// two memsets and a bounds kernel): k_cell_count leaves every grain's cell and its arrival rank within the cell,
// an exclusive scan of the per-cell counts gives cell_start[0 .. ncell] (cell c = [cell_start[c], cell_start[c + 1])), and
// k_cell_scatter writes the grain to its place and returns the count to zero for the next rebuild. The order of the grains
// WITHIN a cell is whatever order the atomics arrived in -- k_verlet_scan<1> sorts every grain's partners anyway, so the
// list does not depend on it.
__global__ void k_cell_count(int n, const real* __restrict__ x1, const real* __restrict__ x2, real ox,
                             real oy, real cs, int ncx, int ncy, unsigned int* __restrict__ keys,
                             int* __restrict__ rank, int* __restrict__ cell_cnt, real* __restrict__ xreb,
                             real* __restrict__ yreb, const int* __restrict__ gate) {
  LBMDEM_GATE(gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  xreb[i] = x1[i]; yreb[i] = x2[i];   // where the list found the grain: how far it has moved since decides whether the list still holds every close pair (k_obst_update)
  const unsigned c = (unsigned)(cell_coord(x2[i], oy, cs, ncy) * ncx + cell_coord(x1[i], ox, cs, ncx));
  keys[i] = c;
  rank[i] = atomicAdd(&cell_cnt[c], 1);
}

__global__ void k_cell_scatter(int n, const unsigned int* __restrict__ keys, const int* __restrict__ rank,
                               const int* __restrict__ cell_start, int* __restrict__ cell_cnt,
                               int* __restrict__ sorted, const int* __restrict__ gate) {
  LBMDEM_GATE(gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const unsigned c = keys[i];
  sorted[cell_start[c] + rank[i]] = i;
  cell_cnt[c] = 0;   // (every occupied cell is visited by its grains; nobody reads the counts in this launch)
}

// the reference's candidate test (main.c:1527-1532), lo < hi
__device__ __forceinline__ bool verlet_pair(real x1l, real x2l, real rl, real x1h, real x2h,
                                            real rh, real dV) {
  const real ddx = x1l - x1h;
  const real ddy = x2l - x2h;
  // main.c:1529-1532: fabs() and sqrt() make the left-hand sides double sums in either build
  if (((fabs((double)ddx) - rl - rh) <= dV) && ((fabs((double)ddy) - rl - rh) <= dV))
    return (sqrt((double)(ddx * ddx + ddy * ddy)) - rl - rh) <= dV;
  return false;
}

// wall candidate lists as per-grain flags: main.c:1563-1593
__device__ __forceinline__ unsigned char wall_flags_of(real x1, real x2, real r, const DemParams& P) {
  unsigned f = 0;
  if (x2 - r - P.Mby < P.distVerlet) f |= 1u;
  if (-x2 - r + P.Mhy < P.distVerlet) f |= 2u;
  if (x1 - r - P.Mgx < P.distVerlet) f |= 4u;
  if (-x1 - r + P.Mdx < P.distVerlet) f |= 8u;
  return (unsigned char)f;
}

// MODE 0: count partners; MODE 1: write them (then sort ascending), clamp the offsets, wall flags -- one thread per grain
template <int MODE>
__global__ void k_verlet_scan(int n, const real* __restrict__ x1, const real* __restrict__ x2,
                              const real* __restrict__ r, real ox, real oy, real cs, int ncx, int ncy,
                              const int* __restrict__ cell_start,
                              const int* __restrict__ sorted, real dV, int* __restrict__ counts,
                              int* __restrict__ offsets, int* __restrict__ nbr, int* __restrict__ own,
                              long cap, int* __restrict__ overflow, DemParams P, unsigned char* __restrict__ wallflags) {
  LBMDEM_GATE(P.gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const real xi = x1[i], yi = x2[i], ri = r[i];
  const int cx = cell_coord(xi, ox, cs, ncx), cy = cell_coord(yi, oy, cs, ncy);
  int cnt = 0;
  int base = 0;
  if (MODE) {
    // offsets[] arrives as the plain exclusive sum of the counts. A list longer than the allocation is truncated (and
    // flagged): no kernel may index nbr[] / own[] past `cap`, so every grain clamps its own offset, and the last grain
    // writes the total (before: two more launches, k_set_last_offset and k_clamp_offsets)
    const long raw = offsets[i];
    if (i == n - 1) {
      const long total = raw + counts[i];
      if (total > cap) *overflow = 1;
      offsets[n] = (int)(total > cap ? cap : total);
    }
    if (raw > cap) { offsets[i] = (int)cap; *overflow = 1; }
    base = (int)(raw > cap ? cap : raw);
    wallflags[i] = wall_flags_of(xi, yi, ri, P);   // VerletWall's four candidate lists (main.c:1563-1593)
  }
  for (int yy = cy - 1; yy <= cy + 1; ++yy) {
    if (yy < 0 || yy >= ncy) continue;
    for (int xx = cx - 1; xx <= cx + 1; ++xx) {
      if (xx < 0 || xx >= ncx) continue;
      const int c = yy * ncx + xx;
      const int s = cell_start[c], e = cell_start[c + 1];
      for (int k = s; k < e; ++k) {
        const int j = sorted[k];
        if (j == i) continue;
        const bool hit = (i < j) ? verlet_pair(xi, yi, ri, x1[j], x2[j], r[j], dV)
                                 : verlet_pair(x1[j], x2[j], r[j], xi, yi, ri, dV);
        if (!hit) continue;
        if (MODE) {
          if ((long)base + cnt < cap) { nbr[base + cnt] = j; own[base + cnt] = i; } else *overflow = 1;
        }
        ++cnt;
      }
    }
  }
  if (!MODE) {
    counts[i] = cnt;
  } else {
    // ascending partner order (insertion sort; lists are ~6 long)
    const long end = ((long)base + cnt < cap) ? base + cnt : cap;
    for (long a = base + 1; a < end; ++a) {
      const int v = nbr[a];
      long b = a - 1;
      while (b >= base && nbr[b] > v) { nbr[b + 1] = nbr[b]; --b; }
      nbr[b + 1] = v;
    }
  }
}

// owner of every list entry, from the offsets (after a checkpoint load; k_verlet_scan<1> writes it directly)
__global__ void k_fill_own(int n, const int* __restrict__ offsets, int* __restrict__ own, const int* __restrict__ gate) {
  LBMDEM_GATE(gate);
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  for (int k = offsets[i]; k < offsets[i + 1]; ++k) own[k] = i;
}

// What k_dem_chain needs of a freshly built list, one workgroup per tile: the distinct partners outside the tile (an LDS
// hash set, then numbered by a scan over the table) and every entry's emeta word. Which halo slot a grain gets depends on
// the order the atomics arrive in -- only the LDS placement, never a result.
__global__ __launch_bounds__(256) void k_tile_halo(int n, const int* __restrict__ offsets, const int* __restrict__ nbr,
                                                   const int* __restrict__ tile_grains, const int* __restrict__ where,
                                                   int* __restrict__ halo_ids,
                                                   int* __restrict__ halo_cnt, unsigned* __restrict__ emeta,
                                                   unsigned char* __restrict__ tile_far, int tiles_per_xcd,
                                                   const int* __restrict__ gate) {
  LBMDEM_GATE(gate);
  constexpr int CELLS = 1024;
  __shared__ int table[CELLS], cidx[CELLS];
  __shared__ int wsum[4];
  __shared__ int sK0[DEM_TILE], sPre[DEM_TILE + 1];   // first list entry of the tile's grains; their entries' positions in the tile
  const int tile = blockIdx.x, tid = threadIdx.x;
  for (int c = tid; c < CELLS; c += 256) table[c] = -1;
  if (tid < 64) {   // the tile's entries: its grains' list ranges one after the other
    const int g = tile_grains[(long)tile * DEM_TILE + tid];
    const int k0 = g >= 0 ? offsets[g] : 0, cnt = g >= 0 ? offsets[g + 1] - k0 : 0;
    int incl = cnt;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int up = __shfl_up(incl, d);
      if (tid >= d) incl += up;
    }
    sK0[tid] = k0; sPre[tid + 1] = incl;
    if (tid == 0) sPre[0] = 0;
  }
  __syncthreads();
  const int E = sPre[DEM_TILE];
  auto entry_of = [&](int p, int& li) {   // position in the tile -> the grain's place in the tile and the list entry
    int lo = 0, hi = DEM_TILE - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (sPre[mid] <= p) lo = mid; else hi = mid - 1; }
    li = lo;
    return sK0[lo] + (p - sPre[lo]);
  };
  bool farp = false;
  for (int p = tid; p < E; p += 256) {
    int li;
    const int gj = nbr[entry_of(p, li)];
    const int tj = where[gj] >> 6;
    if (tj == tile) continue;
    farp = farp || tj / tiles_per_xcd != tile / tiles_per_xcd;
    unsigned hsh = ((unsigned)gj * 2654435761u) >> 22;   // 10 bits
    for (int probe = 0; probe < CELLS; ++probe) {
      const int old = atomicCAS(&table[hsh], -1, gj);
      if (old == -1 || old == gj) break;
      hsh = (hsh + 1) & (CELLS - 1);
    }
  }
  __syncthreads();
  // number the occupied cells: four consecutive cells per thread, wave scan, wave totals
  int cnt = 0;
#pragma unroll
  for (int k = 0; k < 4; ++k) cnt += table[4 * tid + k] >= 0 ? 1 : 0;
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int up = __shfl_up(incl, d);
    if ((tid & 63) >= d) incl += up;
  }
  if ((tid & 63) == 63) wsum[tid >> 6] = incl;
  __syncthreads();
  int basei = incl - cnt;
  for (int w = 0; w < (tid >> 6); ++w) basei += wsum[w];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = 4 * tid + k;
    if (table[c] >= 0) {
      cidx[c] = basei;
      if (basei < DEM_CHAIN_HALO) halo_ids[(long)tile * DEM_CHAIN_HALO + basei] = table[c];
      ++basei;
    }
  }
  if (tid == 255) halo_cnt[tile] = basei < DEM_CHAIN_HALO ? basei : DEM_CHAIN_HALO;
  // does this tile have a partner in a tile of another XCD's eighth? (then it publishes through memory, for everybody)
  const int anyfar = __syncthreads_or(farp ? 1 : 0);
  if (tid == 0) tile_far[tile] = (unsigned char)(anyfar ? 1 : 0);
  for (int p = tid; p < E; p += 256) {
    int li;
    const int e = entry_of(p, li);
    const int gi = tile_grains[(long)tile * DEM_TILE + li], gj = nbr[e];
    unsigned slot = DEM_CHAIN_DIRECT;
    if ((where[gj] >> 6) == tile) slot = (unsigned)(where[gj] & 63);
    else {
      unsigned hsh = ((unsigned)gj * 2654435761u) >> 22;
      for (int probe = 0; probe < CELLS; ++probe) {
        const int t = table[hsh];
        if (t == gj) { if (cidx[hsh] < DEM_CHAIN_HALO) slot = (unsigned)(DEM_TILE + cidx[hsh]); break; }
        if (t == -1) break;
        hsh = (hsh + 1) & (CELLS - 1);
      }
    }
    emeta[e] = (unsigned)li | (gi < gj ? 64u : 0u) | (slot << 8);
  }
}

}  // namespace

int carry_track_alloc(CarryTrack& T, int n) {
  T = CarryTrack{};
  T.tiles = (n + DEM_TILE - 1) / DEM_TILE;
  if (T.tiles >= (1 << 24)) return -1;
  const size_t recs = (size_t)T.tiles * 4;
  if (hipMalloc((void**)&T.stamp, sizeof(long long) * recs) != hipSuccess) return -1;
  if (hipMalloc((void**)&T.val, sizeof(real) * (2 * recs + 3)) != hipSuccess) { carry_track_free(T); return -1; }
  if (hipMalloc((void**)&T.who, sizeof(long long) * (recs + 6)) != hipSuccess) { carry_track_free(T); return -1; }
  if (hipMemset(T.who, 0, sizeof(long long) * (recs + 6)) != hipSuccess) { carry_track_free(T); return -1; }
  T.best_key = T.who + recs;
  T.carry = T.val + 2 * recs;
  if (hipMemset(T.stamp, 0xFF, sizeof(long long) * recs) != hipSuccess ||
      hipMemset(T.val, 0, sizeof(real) * (2 * recs + 3)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    carry_track_free(T);
    return -1;
  }
  return 0;
}

void carry_track_free(CarryTrack& T) {
  if (T.stamp) (void)hipFree(T.stamp);
  if (T.who) (void)hipFree(T.who);
  if (T.val) (void)hipFree(T.val);
  T = CarryTrack{};
}

void launch_carry_resolve(const CarryTrack& T, long long min_stamp, hipStream_t st) {
  hipLaunchKernelGGL(k_carry_resolve, dim3(1), dim3(256), 0, st, T, min_stamp);
}


int verlet_alloc(VerletDevice& V, int n, real cs, real ox, real oy, real wx, real wy) {
  V = VerletDevice{};
  V.cs = cs; V.ox = ox; V.oy = oy;
  V.ncx = (int)ceil(wx / cs) + 1; if (V.ncx < 1) V.ncx = 1;
  V.ncy = (int)ceil(wy / cs) + 1; if (V.ncy < 1) V.ncy = 1;
  const size_t ncell = (size_t)V.ncx * V.ncy;
  V.cap = (long)n * 32;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
  A((void**)&V.keys_in, sizeof(unsigned) * n);                                  // cell of grain i
  A((void**)&V.vals_in, sizeof(int) * n); A((void**)&V.vals_out, sizeof(int) * n);   // rank within its cell; grains by cell
  // cell_start[0 .. ncell] (exclusive scan of the counts), cell_end[0 .. ncell] = the per-cell counts (zero between rebuilds)
  A((void**)&V.cell_start, sizeof(int) * (ncell + 1)); A((void**)&V.cell_end, sizeof(int) * (ncell + 1));
  A((void**)&V.counts, sizeof(int) * n); A((void**)&V.offsets, sizeof(int) * (n + 1));
  A((void**)&V.nbr, sizeof(int) * V.cap); A((void**)&V.own, sizeof(int) * V.cap); A((void**)&V.wallflags, n);
  A((void**)&V.overflow, sizeof(int));
  {
    const size_t tiles = ((size_t)n + DEM_TILE - 1) / DEM_TILE;
    A((void**)&V.halo_ids, sizeof(int) * tiles * DEM_CHAIN_HALO); A((void**)&V.halo_cnt, sizeof(int) * tiles);
    A((void**)&V.emeta, sizeof(unsigned) * V.cap);
    A((void**)&V.tile_far, tiles);
    A((void**)&V.tile_grains, sizeof(int) * tiles * DEM_TILE); A((void**)&V.where, sizeof(int) * (size_t)n);
    A((void**)&V.xreb, sizeof(real) * 2 * (size_t)n);
  }
  if (e != hipSuccess) return -1;
  V.yreb = V.xreb + n;
  V.scan_tmp_bytes = 0;
  size_t cells_tmp = 0;
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, V.scan_tmp_bytes, V.counts, V.offsets, n) != hipSuccess) return -1;
  if (hipcub::DeviceScan::ExclusiveSum(nullptr, cells_tmp, V.cell_end, V.cell_start, (int)(ncell + 1)) != hipSuccess) return -1;
  if (cells_tmp > V.scan_tmp_bytes) V.scan_tmp_bytes = cells_tmp;
  A(&V.scan_tmp, V.scan_tmp_bytes);
  if (e != hipSuccess) return -1;
  if (hipMemset(V.cell_end, 0, sizeof(int) * (ncell + 1)) != hipSuccess) return -1;
  if (hipMemset(V.offsets, 0, sizeof(int) * (n + 1)) != hipSuccess) return -1;
  if (hipMemset(V.wallflags, 0, n) != hipSuccess) return -1;
  if (hipMemset(V.overflow, 0, sizeof(int)) != hipSuccess) return -1;
  {   // tiles by index until somebody says otherwise (verlet_set_tiles)
    const size_t tiles = ((size_t)n + DEM_TILE - 1) / DEM_TILE;
    std::vector<int> tg(tiles * DEM_TILE);
    for (size_t k = 0; k < tg.size(); ++k) tg[k] = k < (size_t)n ? (int)k : -1;
    if (verlet_set_tiles(V, n, tg.data()) != 0) return -1;
  }
  // the consumers run on a non-blocking stream that does not order itself after the NULL stream's memsets
  if (hipDeviceSynchronize() != hipSuccess) return -1;
  return 0;
}

int verlet_set_tiles(VerletDevice& V, int n, const int* tg) {
  const size_t tiles = ((size_t)n + DEM_TILE - 1) / DEM_TILE;
  std::vector<int> where((size_t)n, -1);
  for (size_t k = 0; k < tiles * DEM_TILE; ++k) {
    const int g = tg[k];
    if (g < 0) continue;
    if (g >= n || where[g] != -1) return -1;                                      // every grain exactly once
    if ((k % DEM_TILE) != 0 && tg[k - 1] >= 0 && tg[k - 1] >= g) return -1;       // ascending within a tile
    where[g] = (int)k;                                                            // = tile << 6 | position (DEM_TILE = 64)
  }
  for (int g = 0; g < n; ++g) if (where[g] < 0) return -1;
  static_assert(DEM_TILE == 64, "where[] packs the position into six bits");
  if (hipMemcpy(V.tile_grains, tg, sizeof(int) * tiles * DEM_TILE, hipMemcpyHostToDevice) != hipSuccess) return -1;
  if (hipMemcpy(V.where, where.data(), sizeof(int) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) return -1;
  return 0;
}

void verlet_free(VerletDevice& V) {
  void* ps[] = {V.keys_in, V.vals_in, V.vals_out, V.cell_start, V.cell_end,
                V.scan_tmp, V.counts, V.offsets, V.nbr, V.own, V.wallflags, V.overflow, V.halo_ids, V.halo_cnt, V.emeta, V.tile_far, V.xreb, V.tile_grains, V.where};
  for (void* p : ps) if (p) (void)hipFree(p);
  V = VerletDevice{};
}

int launch_verlet_rebuild(VerletDevice& V, const Kin& K, const real* r, const DemParams& P,
                          hipStream_t st) {
  hipError_t e = hipSuccess;
  const int n = P.n;
  const int nb = (n + 255) / 256;
  const size_t ncell = (size_t)V.ncx * V.ncy;
  // 9 dependent launches (20 before the counting sort and the folded offset / wall-flag kernels: a rebuild is launch
  // latency, 130 us per 100 DEM steps for a few us of work)
  hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, st, n, K.x1, K.x2, V.ox, V.oy, V.cs, V.ncx, V.ncy,
                     V.keys_in, V.vals_in, V.cell_end, V.xreb, V.yreb, V.gate);
  e = hipcub::DeviceScan::ExclusiveSum(V.scan_tmp, V.scan_tmp_bytes, V.cell_end, V.cell_start, (int)(ncell + 1), st);
  if (e != hipSuccess) {   // k_cell_scatter, which returns the counts to zero for the next rebuild, will not run
    (void)hipMemsetAsync(V.cell_end, 0, sizeof(int) * (ncell + 1), st);
    return (int)e;
  }
  hipLaunchKernelGGL(k_cell_scatter, dim3(nb), dim3(256), 0, st, n, V.keys_in, V.vals_in, V.cell_start, V.cell_end,
                     V.vals_out, V.gate);
  hipLaunchKernelGGL(k_verlet_scan<0>, dim3(nb), dim3(256), 0, st, n, K.x1, K.x2, r, V.ox, V.oy, V.cs, V.ncx,
                     V.ncy, V.cell_start, V.vals_out, P.distVerlet, V.counts, V.offsets, V.nbr,
                     V.own, V.cap, V.overflow, P, V.wallflags);
  e = hipcub::DeviceScan::ExclusiveSum(V.scan_tmp, V.scan_tmp_bytes, V.counts, V.offsets, n, st);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(k_verlet_scan<1>, dim3(nb), dim3(256), 0, st, n, K.x1, K.x2, r, V.ox, V.oy, V.cs, V.ncx,
                     V.ncy, V.cell_start, V.vals_out, P.distVerlet, V.counts, V.offsets, V.nbr,
                     V.own, V.cap, V.overflow, P, V.wallflags);
  launch_tile_halo(V, n, st);
  return (int)hipGetLastError();
}

int diag_extra_alloc(DiagExtra& X, int n, long cap, real* carry) {
  X = DiagExtra{};
  real* d = nullptr;
  const size_t nd = 5 * (size_t)n + 6 * (size_t)cap;
  if (hipMalloc((void**)&d, sizeof(real) * nd) != hipSuccess) return -1;
  X.fr = d;   // from here on diag_extra_free() releases whatever was allocated
  if (hipMemset(d, 0, sizeof(real) * nd) != hipSuccess) { diag_extra_free(X); return -1; }
  X.fr = d; X.ice = d + n; X.slip = d + 2 * (size_t)n; X.rw = d + 3 * (size_t)n; X.a1gc = d + 4 * (size_t)n;
  real* e = d + 5 * (size_t)n;
  X.e_ft = e; X.e_f3 = e + cap; X.e_avt = e + 2 * cap; X.e_av3 = e + 3 * cap; X.e_dslip = e + 4 * cap; X.e_drw = e + 5 * cap;
  X.carry = carry;   // lives with the handle's CarryTrack
  if (hipMalloc((void**)&X.e_touched, (size_t)cap) != hipSuccess) { diag_extra_free(X); return -1; }
  if (hipMemset(X.e_touched, 0, (size_t)cap) != hipSuccess) { diag_extra_free(X); return -1; }
  if (hipMalloc((void**)&X.wlist, sizeof(int) * (4 * (size_t)n + 4)) != hipSuccess) { diag_extra_free(X); return -1; }
  X.wcount = X.wlist + 4 * (size_t)n;
  // the consumers run on a non-blocking stream that does not order itself after the NULL stream's memsets
  if (hipDeviceSynchronize() != hipSuccess) { diag_extra_free(X); return -1; }
  return 0;
}

void diag_extra_free(DiagExtra& X) {
  if (X.fr) (void)hipFree(X.fr);
  if (X.e_touched) (void)hipFree(X.e_touched);
  if (X.wlist) (void)hipFree(X.wlist);
  X = DiagExtra{};
}

void launch_diag_extra(const DiagExtra& X, const Kin& in, const real* r, const VerletDevice& V,
                       const DemParams& P, int film, hipStream_t st) {
  const int n = P.n;
  hipLaunchKernelGGL(k_diag_scan, dim3(1), dim3(1024), 0, st, X, V.offsets, n, P.kt);
  if (film)
    hipLaunchKernelGGL(k_diag_accum<true>, dim3((n + 255) / 256), dim3(256), 0, st, X, V.offsets, V.nbr, n);
  else
    hipLaunchKernelGGL(k_diag_accum<false>, dim3((n + 255) / 256), dim3(256), 0, st, X, V.offsets, V.nbr, n);
  hipLaunchKernelGGL(k_diag_wall_lists, dim3(1), dim3(256), 0, st, X, V.wallflags, n);
  hipLaunchKernelGGL(k_diag_walls, dim3(1), dim3(1), 0, st, X, in, r, P);
}

void launch_fill_own(const VerletDevice& V, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_fill_own, dim3((n + 255) / 256), dim3(256), 0, st, n, V.offsets, V.own, V.gate);
}

void launch_dem_substep(const Kin& in, const Kin& out, const real* r, const real* m,
                        const real* It, const real* fhf, const VerletDevice& V, real* pout,
                        const DemParams& P, int film, real* diag, const DiagExtra* X, const unsigned char* active,
                        const CarryTrack* track, long long stamp, const unsigned char* owner, const ObstFillJob& fill,
                        hipStream_t st) {
  const int n = P.n;
  const CarryTrack T = track ? *track : CarryTrack{};
#ifdef LBMDEM_AB
  static const int variant = getenv("LBMDEM_DEM_VARIANT") ? atoi(getenv("LBMDEM_DEM_VARIANT")) : 1;
#else
  const int variant = 1;
#endif
  if (!diag && variant == 1) {  // one lane per list entry
    const int nbe = (n + DEM_GRAINS - 1) / DEM_GRAINS;
    int extra = 0;
    if (fill.map && fill.row1 > fill.row0) {   // ~8 16-byte stores per thread, at most one workgroup per CU
      const long v4 = (long)(fill.row1 - fill.row0) * fill.L.sy / 4;
      extra = (int)((v4 + DEM_THREADS * 8 - 1) / (DEM_THREADS * 8));
      if (extra > 256) extra = 256;
    }
    if (film)
      hipLaunchKernelGGL(k_dem_entries<true>, dim3(DEM_GRID(nbe) + extra), dim3(DEM_THREADS), 0, st, in, out, r, m, It, fhf, V.offsets, V.nbr,
                         V.own, V.wallflags, pout, P, active, T, stamp, owner, fill, nbe);
    else
      hipLaunchKernelGGL(k_dem_entries<false>, dim3(DEM_GRID(nbe) + extra), dim3(DEM_THREADS), 0, st, in, out, r, m, It, fhf, V.offsets, V.nbr,
                         V.own, V.wallflags, pout, P, active, T, stamp, owner, fill, nbe);
    return;
  }
  if (fill.map && fill.row1 > fill.row0) launch_obst_fill_rows(fill.map, fill.L, fill.row0, fill.row1, st);   // (the diagnostic sub-step has its own kernel)
  const int nb = (P.n + 127) / 128;
  DiagOut D{};
  if (diag) {  // [8][n] doubles then [2][n] ints
    D.s = diag; D.f1 = diag + n; D.f2 = diag + 2 * n; D.ifm = diag + 3 * n;
    D.M11 = diag + 4 * n; D.M12 = diag + 5 * n; D.M21 = diag + 6 * n; D.M22 = diag + 7 * n;
    D.z = reinterpret_cast<int*>(diag + 8 * (size_t)n); D.zz = D.z + n;
    D.X = *X;
  }
#define LBM_DEM_LAUNCH(FILM, DIAG)                                                                        \
  hipLaunchKernelGGL((k_dem_substep<FILM, DIAG>), dim3(nb), dim3(128), 0, st, in, out, r, m, It, fhf,    \
                     V.offsets, V.nbr, V.wallflags, pout, D, P, active)
  if (film) { if (diag) LBM_DEM_LAUNCH(true, true); else LBM_DEM_LAUNCH(true, false); }
  else { if (diag) LBM_DEM_LAUNCH(false, true); else LBM_DEM_LAUNCH(false, false); }
#undef LBM_DEM_LAUNCH
}


void launch_tile_halo(const VerletDevice& V, int n, hipStream_t st) {
  const int tiles = (n + DEM_TILE - 1) / DEM_TILE;
  hipLaunchKernelGGL(k_tile_halo, dim3(tiles), dim3(256), 0, st, n, V.offsets, V.nbr, V.tile_grains, V.where, V.halo_ids,
                     V.halo_cnt, V.emeta, V.tile_far, dem_chain_tslots(n) >> 3, V.gate);
}

int dem_chain_tslots(int n) { return DEM_GRID((n + DEM_GRAINS - 1) / DEM_GRAINS); }

int dem_chain_alloc(DemChain& C, int n) {
  C = DemChain{};
  const size_t bytes = (size_t)4 * n * 128;   // two parities x (local copy, remote copy)
  if (bytes >= ((size_t)1 << 31)) return 0;   // 32-bit buffer offsets: the chain stays off (capacity 0)
  if (hipMalloc(&C.pub, bytes ? bytes : 128) != hipSuccess) return -1;
  C.pub_bytes = bytes;
  if (hipMalloc((void**)&C.census, 4 * sizeof(int)) != hipSuccess) { dem_chain_free(C); return -1; }   // counter, pad, 64-bit placement map
  if (hipMalloc((void**)&C.gate, sizeof(int)) != hipSuccess || hipMemset(C.gate, 0, sizeof(int)) != hipSuccess) { dem_chain_free(C); return -1; }
  if (hipHostMalloc((void**)&C.err_host, sizeof(int), hipHostMallocDefault) != hipSuccess) { dem_chain_free(C); return -1; }
  *C.err_host = 0;
  if (hipHostGetDevicePointer((void**)&C.err, (void*)C.err_host, 0) != hipSuccess) { dem_chain_free(C); return -1; }
  if (hipMemset(C.pub, 0, bytes ? bytes : 128) != hipSuccess || hipMemset(C.census, 0, 4 * sizeof(int)) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) { dem_chain_free(C); return -1; }
  return 0;
}

void dem_chain_free(DemChain& C) {
  if (C.pub) (void)hipFree(C.pub);
  if (C.census) (void)hipFree(C.census);
  if (C.gate) (void)hipFree(C.gate);
  if (C.err_host) (void)hipHostFree((void*)C.err_host);
  C = DemChain{};
}

// All workgroups of the tile slots have to run at the same time (a tile waits for the tiles of its partners). The
// occupancy query bounds it; a census launch of the SAME kernel (same registers, same LDS) with exactly that many
// workgroups, each waiting until all have checked in, proves it on this GPU. Synchronises the stream; once per handle.
int dem_chain_census(DemChain& C, int tslots, hipStream_t st) {
  C.capacity = 0;
  if (!C.pub) return 0;
  int per_cu = 0, dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_dem_chain, DEM_THREADS, 0) != hipSuccess) return 0;
  if ((long)per_cu * prop.multiProcessorCount < tslots) return 0;
  if (hipMemsetAsync(C.census, 0, 4 * sizeof(int), st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 0;
  *C.err_host = 0;
  hipLaunchKernelGGL(k_dem_chain, dim3(tslots), dim3(DEM_THREADS), 0, st, Kin{}, Kin{}, nullptr, nullptr, nullptr, nullptr,
                     nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, DemParams{}, nullptr, CarryTrack{}, 0ll,
                     nullptr, ObstFillJob{}, tslots, -1, C.pub, (unsigned)C.pub_bytes, C.err, C.census, nullptr, 0, ChainPaint{}, nullptr);
  int seen[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(seen, C.census, sizeof(seen), hipMemcpyDeviceToHost, st) != hipSuccess) return 0;
  if (hipStreamSynchronize(st) != hipSuccess) return 0;
  // placement: every residue class of the block index on exactly one XCD, the eight classes on eight different ones
  unsigned long long map = 0;
  memcpy(&map, seen + 2, sizeof(map));
  unsigned used = 0;
  bool regular = true;
  for (int c = 0; c < 8; ++c) {
    const unsigned byte = (unsigned)(map >> (8 * c)) & 0xFFu;
    if (byte == 0 || (byte & (byte - 1)) != 0 || (used & byte)) regular = false;
    used |= byte;
  }
  if (*C.err_host == 0 && seen[0] == tslots && regular) C.capacity = tslots;
  *C.err_host = 0;
  return C.capacity;
}

void launch_dem_chain(const Kin& in, const Kin& out, const real* r, const real* m, const real* It, const real* fhf,
                      const VerletDevice& V, real* pout, const DemParams& P, const unsigned char* active,
                      const CarryTrack* track, long long stamp0, const unsigned char* owner, const ObstFillJob& fill,
                      const DemChain& C, int nsteps, const ChainPaint& paint, hipStream_t st) {
  const int n = P.n;
  const CarryTrack T = track ? *track : CarryTrack{};
  const int nbe = (n + DEM_GRAINS - 1) / DEM_GRAINS;
  int extra = 0;
  if (fill.map && fill.row1 > fill.row0) {
    const long v4 = (long)(fill.row1 - fill.row0) * fill.L.sy / 4;
    extra = (int)((v4 + DEM_THREADS * 8 - 1) / (DEM_THREADS * 8));
    if (extra > 128) extra = 128;   // next to a latency chain: few workgroups, many stores each
  }
  int one_xcd = 0;
#ifdef LBMDEM_AB
  static const bool want_one = getenv("LBMDEM_CHAIN_ONE_XCD") != nullptr;
  one_xcd = want_one && nbe <= 128 ? 1 : 0;
  static const int extra_sleep = getenv("LBMDEM_CHAIN_SLEEP") ? atoi(getenv("LBMDEM_CHAIN_SLEEP")) : 0;
  one_xcd |= (extra_sleep & 0xFF) << 8;
  static const bool tail_cut = getenv("LBMDEM_CHAIN_TAILCUT") != nullptr;
  if (tail_cut) one_xcd |= 1 << 16;
  if (C.capacity < 0) one_xcd |= 1 << 17;   // lbmdem_debug_chain_giveup: a tile of this launch gives up half way
#endif
  hipLaunchKernelGGL(k_dem_chain, dim3(((one_xcd & 1) ? nbe * 8 : DEM_GRID(nbe)) + extra), dim3(DEM_THREADS), 0, st, in, out, r, m, It, fhf, V.offsets,
                     V.nbr, V.emeta, V.halo_ids, V.halo_cnt, V.tile_grains, V.where, V.tile_far, V.wallflags, pout, P, active, T, stamp0, owner, fill, nbe, nsteps,
                     C.pub, (unsigned)C.pub_bytes, C.err, C.census, C.dbg, one_xcd, paint, C.gate);
}
