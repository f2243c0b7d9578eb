// lbm_obst.hip -- the obstacle map (obst_construction, main.c:991-1065): reset and rasterisation of the reduced discs.
// `act` and `delta` are not stored: the fused kernel recomputes them from the map and the grain centres.

#include "lbm_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// small kernels
// ---------------------------------------------------------------------------------------------

// obst = -1 in the interior, nbgrains on the four lattice edges (main.c:669-683, 997-999): obst_fill_range (lbmdem_internal.h)
__global__ void k_obst_fill(int* __restrict__ obst, LatticeView L, int row0, int row1) {
  LBMDEM_GATE(L.gate);
  obst_fill_range(obst, L, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)gridDim.x * blockDim.x, row0, row1);
}

// Rasterise the reduced discs (main.c:1016-1032). A quarter of a wavefront per grain (round 4; a wavefront before): every lane derives the grain's lattice
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
// GL = lanes per grain: 16 (four grains per wavefront) when the pair list tells which discs are alone -- nearly all, and
// they take the plain-store sweep --, 64 (one grain per wavefront, the round-3 form) when every grain takes the atomic path
// (no list: strips with distributed grains, the first step): its batched atomics want the whole box in one sweep.
constexpr int PAINT_LANES = 16;
template <int GL>
__global__ void k_obst_paint(int* __restrict__ obst, LatticeView L, int n, const real* __restrict__ x1,
                             const real* __restrict__ x2, const real* __restrict__ r,
                             const real* __restrict__ rLB, const real* __restrict__ v1,
                             const real* __restrict__ v2, const real* __restrict__ v3,
                             real* __restrict__ oxc, real* __restrict__ oyc, real* __restrict__ or2,
                             real* __restrict__ orbl0, real* __restrict__ pk,
                             unsigned char* __restrict__ touched, const unsigned char* __restrict__ mask,
                             unsigned* __restrict__ mincov, unsigned epoch, const int* __restrict__ list,
                             const int* __restrict__ list_count, int list_cap, const int* __restrict__ voff,
                             const int* __restrict__ vnbr, ObstSnap snap_out) {
  LBMDEM_GATE(L.gate);
  // FOUR grains per wavefront, PAINT_LANES = 16 lanes each, sweeping their boxes (11 ... 19 columns) in chunks of 16 columns:
  // the per-grain set-up -- geometry, three divisions, the partner test -- is paid once per four grains and the sweep's lanes
  // are mostly busy. One wavefront per grain 46 us (the kernel was bound by its 50 000 short waves), 32 lanes per grain 32.5,
  // 16: 27, 8: 29 us (A/B, round 4).
  const int lane = threadIdx.x & 63, half = lane / GL, hl = lane % GL;
  // XCD k (workgroups b % 8 == k) takes the k-th contiguous eighth of the grains, as the DEM sub-step does: the positions
  // and the partner lists a grain's lanes read were written through that XCD's L2 (36 -> 32.5 us, A/B)
  // (not with a list: its entries fill only the head of the launch's positions, which would all land on the first XCDs)
  const int bslots = (int)gridDim.x >> 3;
  const int blk = list ? (int)blockIdx.x : ((int)blockIdx.x & 7) * bslots + ((int)blockIdx.x >> 3);
  int i = (int)(((long)blk * blockDim.x + threadIdx.x) / GL);
  if (list) {                    // strip decomposition: only the grains that can reach this rank's rows
    if (i >= *list_count || i >= list_cap) return;   // (an overflowing list is flagged by its producer)
    i = list[i];
  }
  if (i >= n) return;
  if (mask && !mask[i]) return;
  const real gx1 = x1[i], gx2 = x2[i];
  const real xc = (gx1 - L.Mgx) / L.dx, yc = (gx2 - L.Mby) / L.dx, r2 = rLB[i] * rLB[i], rbl0 = r[i] / L.dx;
  if (hl == 0) {
    oxc[i] = xc; oyc[i] = yc; or2[i] = r2; orbl0[i] = rbl0;
    real* o = pk + (long)i * 8;
    o[0] = gx1; o[1] = gx2; o[2] = v1[i]; o[3] = v2[i]; o[4] = v3[i]; o[5] = xc; o[6] = yc; o[7] = r2;
    if (snap_out.xc) { snap_out.xc[i] = xc; snap_out.yc[i] = yc; snap_out.still2[i] = 0.; snap_out.mode[i] = 1; }   // what k_obst_update starts from
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
    for (int k = k0 + hl; k < k1; k += GL) {
      const int j = vnbr[k];
      const real ddx = (x1[j] - gx1) / L.dx, ddy = (x2[j] - gx2) / L.dx, rr = ri + rLB[j] + 1.5;
      near |= !(ddx * ddx + ddy * ddy >= rr * rr);   // also true for a NaN
    }
    alone = ((__ballot(near) >> ((GL * half) & 63)) & (GL >= 64 ? ~0ull : (1ull << (GL & 63)) - 1)) == 0;   // this grain's lanes
  }
  if (alone) {
    for (int y = yi + hl; y <= yf; y += GL)          // the box in chunks of GL columns, y fastest: no integer divisions
      for (int x = xi; x <= xf; ++x)
        if (in_disc(x, y)) obst[(long)(x - L.gx0) * L.sy + y] = i;
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
  // the box in one sweep: XR rows at a time, y fastest (no integer divisions); all atomics of the grain's lanes are issued
  // before the first returned value is looked at: one round trip, not SWEEPS
  constexpr int XR = GL >= 64 ? 2 : 1, YL = GL / XR, SWEEPS = GL >= 64 ? 12 : 24;
  if (ny <= YL && xf - xi + 1 <= XR * SWEEPS) {
    const int y = yi + hl % YL, xsub = hl / YL;
    const bool col = hl % YL < ny;
    int old[SWEEPS];
#pragma unroll
    for (int s_ = 0; s_ < SWEEPS; ++s_) {
      const int x = xi + XR * s_ + xsub;
      old[s_] = -1;
      if (col && x <= xf && in_disc(x, y)) old[s_] = atomicMax(&obst[(long)(x - L.gx0) * L.sy + y], i);
    }
#pragma unroll
    for (int s_ = 0; s_ < SWEEPS; ++s_) {
      const int x = xi + XR * s_ + xsub;
      if (old[s_] >= 0) overlap((long)(x - L.gx0) * L.sy + y, old[s_]);
    }
  } else {
    const int total = (xf - xi + 1) * ny;
    for (int k = hl; k < total; k += GL) {
      const int x = xi + k / ny, y = yi + k % ny;
      if (in_disc(x, y)) {
        const long node = (long)(x - L.gx0) * L.sy + y;
        overlap(node, atomicMax(&obst[node], i));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// The map updated in place
// ---------------------------------------------------------------------------------------------
//
// obst_construction (main.c:991-1065) clears the map and paints every disc again because it is serial C. Between two
// fluid steps a grain moves by a fraction of a node, so the canvas this rasterisation writes to -- the map of TWO steps
// ago, the other buffer is the "previous map" of reinit_obst_density -- already holds nearly every node right. Each map
// buffer remembers the centres its discs were painted at (ObstSnap); a grain compares its footprint THEN with its
// footprint NOW, node by node with the reference's own test (main.c:1027), and writes only the nodes that differ: no reset
// of the canvas (67 MB per step at 4096^2), a few thousand stores instead of 11 M.
//   * A disc that has moved less than half a node since and keeps clear (1.1 nodes) of the discs of all its partners in
//     the pair list shares no node with another disc in either picture: plain stores.
//   * Any other disc settles every node with the grains that can reach it, whatever order the grains are processed in:
//     a node it leaves goes to the highest-index partner whose NEW disc covers it (else to the fluid) by compare-and-swap
//     on its own id; a node it covers takes atomicMax (main.c:1028: the highest index wins). A stale higher owner that
//     has moved away hands the node over the same way. Overlap flags and the lowest-cover records (`touched`, `mincov`)
//     come from the partners' disc tests, not from what the atomics return (which may be a stale owner here).
// Same maps, bit for bit, as clear + repaint (tests: every golden runs through it; tests/test_gpu_obst_update.py).
template <int GL>
__global__ void k_obst_update(int* __restrict__ obst, LatticeView L, int n, const real* __restrict__ x1,
                              const real* __restrict__ x2, const real* __restrict__ r,
                              const real* __restrict__ rLB, const real* __restrict__ v1,
                              const real* __restrict__ v2, const real* __restrict__ v3,
                              real* __restrict__ oxc, real* __restrict__ oyc, real* __restrict__ or2,
                              real* __restrict__ orbl0, real* __restrict__ pk,
                              unsigned char* __restrict__ touched, unsigned* __restrict__ mincov, unsigned epoch,
                              const int* __restrict__ voff, const int* __restrict__ vnbr, ObstSnap was, ObstSnap now,
                              const real* __restrict__ xreb, const real* __restrict__ yreb, real moved_limit,
                              int* __restrict__ moved_flag, int list_generation) {
  LBMDEM_GATE(L.gate);
  const int lane = threadIdx.x & 63, half = lane / GL, hl = lane % GL;
  const int bslots = (int)gridDim.x >> 3;   // XCD k takes the k-th contiguous eighth of the grains (k_obst_paint)
  const int blk = ((int)blockIdx.x & 7) * bslots + ((int)blockIdx.x >> 3);
  const int i = (int)(((long)blk * blockDim.x + threadIdx.x) / GL);
  if (i >= n) return;
  const real gx1 = x1[i], gx2 = x2[i], ri = rLB[i];
  const real xc = (gx1 - L.Mgx) / L.dx, yc = (gx2 - L.Mby) / L.dx, r2 = ri * ri, rbl0 = r[i] / L.dx;
  const bool had = was.mode[i] != 0;
  const real pxc = was.xc[i], pyc = was.yc[i];
  {
    // The list holds every pair within distVerlet of touching WHEN IT WAS BUILT; a disc can only have met one that is not
    // in its list if a grain has travelled more than half that distance since. Said to the host (pinned word, the list's
    // generation): it clears and repaints with atomics until the next rebuild.
    const real mx = gx1 - xreb[i], my = gx2 - yreb[i];
    if (hl == 0 && !(mx * mx + my * my <= moved_limit * moved_limit)) *moved_flag = list_generation;
  }
  if (hl == 0) {
    oxc[i] = xc; oyc[i] = yc; or2[i] = r2; orbl0[i] = rbl0;
    real* o = pk + (long)i * 8;
    o[0] = gx1; o[1] = gx2; o[2] = v1[i]; o[3] = v2[i]; o[4] = v3[i]; o[5] = xc; o[6] = yc; o[7] = r2;
  }
  const DiscGeo gn = disc_geo(L, xc, yc, ri, rbl0, true), go = disc_geo(L, pxc, pyc, ri, rbl0, had);
  const int k0 = voff[i], k1 = voff[i + 1];
  // Alone = this disc has moved less than half a node since it was painted into this buffer, and its circle keeps 1.1
  // nodes clear of the NEW circles of all its partners (two discs share a node only if their circles meet; touching grains
  // leave 0.15 (r_i + r_j) = 1.5 ... 2.4 nodes between their reduced discs). That is enough without looking at where the
  // partners were: one that has moved less than half a node left the same gap open in the old picture; one that has moved
  // more is not alone itself and hands over by compare-and-swap whatever it owned near this disc -- to the fluid, or to
  // this disc where it covers the node (nothing else can: a third disc there would be within 1.1 nodes of this one).
  // (Physical units: no division per partner.)
  bool near = had && !((xc - pxc) * (xc - pxc) + (yc - pyc) * (yc - pyc) < 0.25);
  for (int k = k0 + hl; k < k1; k += GL) {
    const int j = vnbr[k];
    const real ddx = x1[j] - gx1, ddy = x2[j] - gx2, rr = (ri + rLB[j] + 1.1) * L.dx;
    near |= !(ddx * ddx + ddy * ddy >= rr * rr);   // also true for a NaN
  }
  const bool alone = ((__ballot(near) >> ((GL * half) & 63)) & (GL >= 64 ? ~0ull : (1ull << (GL & 63)) - 1)) == 0;   // this grain's lanes
  // ---- the new picture's record. A disc that has stayed where it was painted, near enough that no node can have changed
  // sides (see `still2` below), leaves the map AND the record as they are: the next comparison is again with the centre
  // the nodes were really painted at.
  const real moved2 = (xc - pxc) * (xc - pxc) + (yc - pyc) * (yc - pyc);
  if (alone && had && moved2 < was.still2[i]) {
    if (hl == 0) { now.xc[i] = pxc; now.yc[i] = pyc; now.still2[i] = was.still2[i]; now.mode[i] = 1; }
    return;
  }
  if (hl == 0) { now.xc[i] = xc; now.yc[i] = yc; now.still2[i] = 0.; now.mode[i] = 1; }
  if (!gn.any && !go.any) return;
  if (alone && had && gn.r2 <= gn.R2) {
    // The disc has moved by |D| (< 1/2 node): only nodes with |d - r| <= |D| can have changed sides (a node at distance d
    // from the new centre was at d -+ |D| from the old one) -- an annulus a few hundredths of a node wide. Per lattice row
    // two square roots give its two stretches, most of which hold no node; every candidate is settled by the reference's
    // own test at both centres. w = |D| + what the rounding of the test's d2 can amount to, in nodes. (k_dem_chain's
    // rasterisation does the same from LDS.) The record keeps still2 = 0: the next step scans again, which is cheap.
    const real w = (real)sqrt((double)moved2) + (sizeof(real) == 4 ? (real)(1e-3 + 5e-7 * (fabs((double)xc) + fabs((double)yc))) : (real)1e-6);
    const real ro = ri + w, rin = ri > w ? ri - w : 0.;
    const int xlo = (int)floor(xc - ro) - 1, xhi = (int)ceil(xc + ro) + 1;
    auto settle = [&](int x, int y) {
      const bool bo = disc_has(go, x, y), bn = disc_has(gn, x, y);
      if (bn != bo) obst[(long)(x - L.gx0) * L.sy + y] = bn ? i : -1;
    };
    for (int x = xlo + hl; x <= xhi; x += GL) {
      const real dxn = x - xc, o2 = ro * ro - dxn * dxn;
      if (!(o2 >= 0.)) continue;
      const real yo = sqrt(o2), i2 = rin * rin - dxn * dxn, yn = i2 > 0. ? sqrt(i2) : 0.;
      const int a0 = (int)ceil(yc - yo - w), a1 = (int)floor(yc - yn + w), b0 = (int)ceil(yc + yn - w), b1 = (int)floor(yc + yo + w);
      if (a1 >= b0) { for (int y = a0; y <= b1; ++y) settle(x, y); }
      else { for (int y = a0; y <= a1; ++y) settle(x, y); for (int y = b0; y <= b1; ++y) settle(x, y); }
    }
    return;
  }
  if (alone) {
    // the union of the two boxes and one node around it: the nodes whose owner changes are written; on the way, how far
    // the nearest node is from changing sides, as a gap in d2 = |P - C|^2 against r2 -- a node within r + 1 of the centre sees
    // its d2 change by less than (2 r + 3) |D| when the centre moves by |D| < 1/2, one farther out stays outside: nothing
    // changes sides while |D| < gap / (2 r + 3)
    int xi = !go.any ? gn.xi : (!gn.any ? go.xi : (go.xi < gn.xi ? go.xi : gn.xi));
    int xf = !go.any ? gn.xf : (!gn.any ? go.xf : (go.xf > gn.xf ? go.xf : gn.xf));
    int yi = !go.any ? gn.yi : (!gn.any ? go.yi : (go.yi < gn.yi ? go.yi : gn.yi));
    int yf = !go.any ? gn.yf : (!gn.any ? go.yf : (go.yf > gn.yf ? go.yf : gn.yf));
    --xi; ++xf; --yi; ++yf;
    const real rm2 = gn.r2 < gn.R2 ? gn.r2 : gn.R2, near2 = (ri + 1.) * (ri + 1.);
    real gap = 1e30;
    for (int y = yi + hl; y <= yf; y += GL)
      for (int x = xi; x <= xf; ++x) {
        const bool bo = disc_has(go, x, y), bn = disc_has(gn, x, y);
        if (bn != bo) obst[(long)(x - L.gx0) * L.sy + y] = bn ? i : -1;
        const real d2 = (x - xc) * (x - xc) + (y - yc) * (y - yc), g = d2 > rm2 ? d2 - rm2 : rm2 - d2;
        if (d2 <= near2) gap = g < gap ? g : gap;
      }
#pragma unroll
    for (int d = 1; d < GL && d < 64; d <<= 1) {   // over this grain's lanes (GL = a power of two)
      const real o = __shfl_xor(gap, d);
      gap = o < gap ? o : gap;
    }
    if (hl == 0 && gn.any && gn.xi > 1 && gn.xf < L.lx - 2 && gn.yi > 1 && gn.yf < L.ly - 2 && gn.xi > L.gx0 &&
        gn.xf < L.gx0 + L.nxl - 1) {   // (not next to a clamp of the box: there the argument above does not cover the nodes it cuts off)
      const real lim = (gap - (sizeof(real) == 4 ? 1e-3 : 1e-9)) / (2. * ri + 3.);   // (slack for the rounding of d2: k_dem_chain)
      now.still2[i] = (gn.r2 <= gn.R2 && lim > 0.) ? (lim < 0.45 ? lim * lim : 0.2025) : 0.;
    }
    return;
  }
  auto partner_geo = [&](int j) {
    const real jr = rLB[j];
    return disc_geo(L, (x1[j] - L.Mgx) / L.dx, (x2[j] - L.Mby) / L.dx, jr, r[j] / L.dx, true);
  };
  if (go.any) {   // the nodes this disc has left: to the highest partner that covers them now, else to the fluid
    const int ny = go.yf - go.yi + 1, total = (go.xf - go.xi + 1) * ny;
    for (int k = hl; k < total; k += GL) {
      const int x = go.xi + k / ny, y = go.yi + k % ny;
      if (!disc_has(go, x, y) || disc_has(gn, x, y)) continue;
      int v = -1;
      for (int e = k0; e < k1; ++e) {
        const int j = vnbr[e];
        if (j > v && disc_has(partner_geo(j), x, y)) v = j;
      }
      atomicCAS(&obst[(long)(x - L.gx0) * L.sy + y], i, v);
    }
  }
  if (gn.any) {   // the nodes it covers: highest index wins (main.c:1028) ...
    const int ny = gn.yf - gn.yi + 1, total = (gn.xf - gn.xi + 1) * ny;
    for (int k = hl; k < total; k += GL) {
      const int x = gn.xi + k / ny, y = gn.yi + k % ny;
      if (disc_has(gn, x, y)) atomicMax(&obst[(long)(x - L.gx0) * L.sy + y], i);
    }
    // ... and who else covers them, from the partners' own disc tests (one partner at a time: its geometry once)
    for (int e = k0; e < k1; ++e) {
      const int j = vnbr[e];
      const DiscGeo gj = partner_geo(j);
      if (!gj.any || gj.xi > gn.xf || gj.xf < gn.xi || gj.yi > gn.yf || gj.yf < gn.yi) continue;
      for (int k = hl; k < total; k += GL) {
        const int x = gn.xi + k / ny, y = gn.yi + k % ny;
        if (!disc_has(gn, x, y) || !disc_has(gj, x, y)) continue;
        const long node = (long)(x - L.gx0) * L.sy + y;
        touched[i] = 1; touched[j] = 1;
        if (mincov) {
          atomicMax(&mincov[node], (epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)i));
          atomicMax(&mincov[node], (epoch & 0xFFFu) << 20 | (0xFFFFFu - (unsigned)j));
        }
      }
    }
  }
}

// test aid: values that differ between two lattices (the fused kernel run with and without the change bits)
__global__ void k_count_differences(const real* __restrict__ a, const real* __restrict__ b, long n, int* __restrict__ bad) {
  int mine = 0;
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long)gridDim.x * blockDim.x)
    mine += __double_as_longlong((double)a[k]) != __double_as_longlong((double)b[k]) ? 1 : 0;
  if (mine) atomicAdd(bad, mine);
}

// test aid (lbmdem_set_change_mask(h, 2)): one wavefront per (row, window) -- does a clear bit hide a difference?
__global__ __launch_bounds__(64) void k_change_verify(const int* __restrict__ ob_old, const int* __restrict__ ob_new,
                                                      LatticeView L, ObstChange C, int windows, int* __restrict__ bad) {
  const int xl = blockIdx.x / windows, w = blockIdx.x - xl * windows;
  const int y = w * C.ww - C.off + (int)threadIdx.x;
  const bool set = (C.bits[(long)w * C.words + (xl >> 5)] >> (xl & 31)) & 1u;
  bool differ = false;
  if (y >= 0 && y < L.ly) differ = ob_old[(long)xl * L.sy + y] != ob_new[(long)xl * L.sy + y];
  if (__any(differ) && !set && threadIdx.x == 0) atomicAdd(bad, 1);
}

}  // namespace

void launch_obst_fill(int* obst, const LatticeView& L, hipStream_t st) { launch_obst_fill_rows(obst, L, 0, L.nxl, st); }

void launch_obst_fill_rows(int* obst, const LatticeView& L, int row0, int row1, hipStream_t st) {
  if (row1 <= row0) return;
  hipLaunchKernelGGL(k_obst_fill, dim3(grid_for((long)(row1 - row0) * L.sy / 4)), dim3(256), 0, st, obst, L, row0, row1);
}

void launch_obst_paint(int* obst, const LatticeView& L, int n, const real* x1, const real* x2, const real* r,
                       const real* rLB, const real* v1, const real* v2, const real* v3, real* xc,
                       real* yc, real* r2, real* rbl0, real* pk, unsigned char* touched,
                       const unsigned char* mask, unsigned* mincov, unsigned epoch, const int* list,
                       const int* list_count, int list_cap, const int* voff, const int* vnbr, const ObstSnap& snap_out,
                       hipStream_t st) {
  const int gl = voff ? PAINT_LANES : 64;
  const long threads = (long)(list ? list_cap : n) * gl;
  const unsigned pgrid = (unsigned)(((threads + 255) / 256 + 7) / 8 * 8);   // a multiple of the 8 XCDs (see the kernel)
  if (voff)
    hipLaunchKernelGGL(k_obst_paint<PAINT_LANES>, dim3(pgrid), dim3(256), 0, st, obst, L, n, x1, x2, r, rLB, v1, v2, v3, xc, yc, r2,
                       rbl0, pk, touched, mask, mincov, epoch, list, list_count, list_cap, voff, vnbr, snap_out);
  else
    hipLaunchKernelGGL(k_obst_paint<64>, dim3(pgrid), dim3(256), 0, st, obst, L, n, x1, x2, r, rLB, v1, v2, v3, xc, yc, r2, rbl0, pk,
                       touched, mask, mincov, epoch, list, list_count, list_cap, voff, vnbr, snap_out);
}


void launch_obst_update(int* obst, const LatticeView& L, int n, const real* x1, const real* x2, const real* r,
                        const real* rLB, const real* v1, const real* v2, const real* v3, real* xc, real* yc, real* r2,
                        real* rbl0, real* pk, unsigned char* touched, unsigned* mincov, unsigned epoch, const int* voff,
                        const int* vnbr, const ObstSnap& was, const ObstSnap& now, const real* xreb, const real* yreb,
                        real moved_limit, int* moved_flag, int list_generation, hipStream_t st) {
  const long threads = (long)n * PAINT_LANES;
  const unsigned pgrid = (unsigned)(((threads + 255) / 256 + 7) / 8 * 8);
  hipLaunchKernelGGL(k_obst_update<PAINT_LANES>, dim3(pgrid), dim3(256), 0, st, obst, L, n, x1, x2, r, rLB, v1, v2, v3, xc, yc, r2,
                     rbl0, pk, touched, mincov, epoch, voff, vnbr, was, now, xreb, yreb, moved_limit, moved_flag, list_generation);
}

void launch_change_verify(const int* ob_old, const int* ob_new, const LatticeView& L, const ObstChange& chg, int windows, int* bad,
                          hipStream_t st) {
  hipLaunchKernelGGL(k_change_verify, dim3((unsigned)(L.nxl * windows)), dim3(64), 0, st, ob_old, ob_new, L, chg, windows, bad);
}

void launch_count_differences(const real* a, const real* b, long n, int* bad, hipStream_t st) {
  hipLaunchKernelGGL(k_count_differences, dim3(2048), dim3(256), 0, st, a, b, n, bad);
}
