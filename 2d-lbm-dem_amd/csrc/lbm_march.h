// lbm_march.h -- what the marching forms of the fused fluid kernel share (k_cs_march in lbm_fused.hip; the experiment
// build's k_cs_march3 in lbm_fused_ab.hip): obstacle-id rows, the `act` test, the wave-private ring of grain records, the
// register context of a pull, the classification of a row's pulls, the full-wave DPP shifts.
#pragma once

#include "lbm_device.h"

namespace {

// ---------------------------------------------------------------------------------------------
// the fused fluid kernel, register-resident "marching" form
// ---------------------------------------------------------------------------------------------
//
// One WAVEFRONT owns a window of 64 consecutive y (MARCH_WW = 60 produce output -- a window's stores then start on a
// 32-byte sector boundary, lbm_fused.hip -- the two edge lanes on either side only feed their neighbours) and walks along x
// over its rows. Each lane keeps the f* of its column for three consecutive rows in registers; row x+1 is
// re-initialised/collided and rotated in while row x is produced. The six diagonal/vertical neighbours a pull needs live in
// the adjacent lanes and are fetched with full-wave DPP shifts. A lane loads ONE obstacle id per row; what a row's nodes are
// (grain or fluid, active, the end of a bounce-back link) is kept as wave-uniform 64-bit lane masks (below), and a node's
// column neighbours are the same masks shifted by one. No barriers: waves are independent, every load of a row is a full
// 512-byte coalesced request through a buffer resource, and the loads of the next two rows are in flight while the current
// one is computed. LDS is used only wave-privately: a ring of grain records (RecRing), the slots of the compacted bounce-back
// evaluation and a copy of the lattice's constants. Redundant work: 4 of 64 lanes and 3 of LX+3 rows.
// DESIGN.md section 4.1 and LABBOOK.md (rounds 1-6) list what was measured on the way (in-order vmcnt, no loads under
// branches, ...).

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

// the id of (xl, y) alone, same rules; the two column neighbours of a lane are its neighbour lanes' ids (ids3_of)
// (EDGE = false: the column is on the lattice and the row is one of the slab's, or a row whose ids nobody reads -- the
// clamp only keeps the address valid)
template <bool EDGE = true>
__device__ __forceinline__ int load_id(const int* __restrict__ ob, const LatticeView& L, int xl, int y) {
  const bool rok = xl >= 0 && xl < L.nxl;
  const int xc = xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl);
  if (!EDGE) return ob[(long)xc * L.sy + y];
  const int cc = y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y);
  const int vc = ob[(long)xc * L.sy + cc];
  return (rok && cc == y) ? vc : L.n;
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
  int* ids;     // this wave's [REC_RING][64]: the grain the record belongs to (the owner of the node)
  __device__ __forceinline__ void put(int row, int lane, const GP& g, int id) const {
    real2* p = base + (row & (REC_RING - 1)) * 4 * 64 + lane;
    p[0] = make_real2(g.x1, g.x2);
    p[64] = make_real2(g.v1, g.v2);
    p[128] = make_real2(g.v3, g.xc);
    p[192] = make_real2(g.yc, g.r2);
    ids[(row & (REC_RING - 1)) * 64 + lane] = id;
  }
  // position and velocities of that record (what reinit_obst_density needs)
  __device__ __forceinline__ GPv getv(int row, int lane) const {
    const real2* p = base + (row & (REC_RING - 1)) * 4 * 64 + lane;
    const real2 a = p[0], b = p[64];
    return GPv{a.x, a.y, b.x, b.y, p[128].x};
  }
  __device__ __forceinline__ int get_id(int row, int lane) const { return ids[(row & (REC_RING - 1)) * 64 + lane]; }
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

// Buffer resources. The lattices and the maps are addressed as {descriptor in four scalar registers, 32-bit per-lane
// offset, scalar offset, immediate}: a lane's column offset is loop-invariant (ONE register for all nine populations and all
// rows), the row is the scalar offset, the direction the immediate -- where a flat 64-bit address per access costs a
// register pair and three vector instructions. A descriptor starts at the work item's first row, so the offsets stay below
// 2^31 whatever the size of the lattice.
typedef __attribute__((ext_vector_type(2))) unsigned int buf_u2;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rs(const void* p, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes < 0x7FFFFFF0L ? (int)bytes : 0x7FFFFFF0, 0x00020000);
}
template <int IMM>
__device__ __forceinline__ real buf_load_real(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
#ifdef LBMDEM_SINGLE_PRECISION
  return __int_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, voff + IMM, soff, 0));
#else
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, voff + IMM, soff, 0));
#endif
}
template <int IMM>
__device__ __forceinline__ void buf_store_real(real v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
#ifdef LBMDEM_SINGLE_PRECISION
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_int(v), rs, voff + IMM, soff, 0);
#else
  __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(buf_u2, v), rs, voff + IMM, soff, 0);
#endif
}
// byte offset of column y inside a row of populations (direction 0), layout f[x][y / TILE][q][y % TILE]
__device__ __forceinline__ int fcol_bytes(int y) {
  return ((y / LBMDEM_TILE_Y) * (9 * LBMDEM_TILE_Y) + (y % LBMDEM_TILE_Y)) * (int)sizeof(real);
}
constexpr int F_QBYTES = LBMDEM_TILE_Y * (int)sizeof(real);   // from one direction to the next

// Lane masks. A per-lane boolean of a whole row is ONE wave-uniform 64-bit value in a scalar register pair (bit i = lane
// i = column y0 + i), and the same boolean of the node one column up or down is that value shifted by one -- so the
// tests "is the source of this pull a grain node", "is that node active", "has this solid node a fluid neighbour" of a
// whole row are a handful of scalar instructions (which issue beside the other wavefront's vector instructions) instead
// of a vector compare per lane, direction and neighbour. lane_of() hands a mask back to the vector unit as a condition
// (v_cndmask / s_and_saveexec take the register pair as it is: no instruction). Bits shifted in at the window's two end
// lanes are 0: those lanes only feed their neighbours' populations and ids, nothing reads their flags.
typedef unsigned long long lmask;
__device__ __forceinline__ bool lane_of(lmask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// the mask of the nodes at lane + DY (column y + DY), aligned to this lane
template <int DY>
__device__ __forceinline__ lmask m_at(lmask m) { return DY > 0 ? m >> 1 : (DY < 0 ? m << 1 : m); }
// of three rows x-1, x, x+1 the one at x + DX
#define MARCH_ROW(DX, B, C, D) ((DX) < 0 ? (B) : ((DX) > 0 ? (D) : (C)))
// nodes whose eight neighbours are all solid, from the solid masks of the row before, the row itself and the row after
__device__ __forceinline__ lmask m_enclosed(lmask sa, lmask sb, lmask sc) {
  const lmask ha = (sa >> 1) & sa & (sa << 1), hc = (sc >> 1) & sc & (sc << 1);
  return ha & hc & (sb >> 1) & (sb << 1);
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
// (the same shifts with bound_ctrl: the end lane gets 0 instead of keeping its value -- the destination then needs no
// copy of the source first; for the populations, whose end lanes' shifted values nobody reads)
__device__ __forceinline__ int dpp_up1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }
__device__ __forceinline__ int dpp_dn1z(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x130, 0xf, 0xf, true); }
#ifdef LBMDEM_SINGLE_PRECISION
__device__ __forceinline__ real shfl_up1z(real v) { return __int_as_float(dpp_up1z(__float_as_int(v))); }
__device__ __forceinline__ real shfl_dn1z(real v) { return __int_as_float(dpp_dn1z(__float_as_int(v))); }
#else
__device__ __forceinline__ real shfl_up1z(real v) {
  return __hiloint2double(dpp_up1z(__double2hiint(v)), dpp_up1z(__double2loint(v)));
}
__device__ __forceinline__ real shfl_dn1z(real v) {
  return __hiloint2double(dpp_dn1z(__double2hiint(v)), dpp_dn1z(__double2loint(v)));
}
#endif
__device__ __forceinline__ real shfl_up1(real v) { return dpp_up1(v); }
__device__ __forceinline__ real shfl_dn1(real v) { return dpp_dn1(v); }
__device__ __forceinline__ int shfl_up1(int v) { return dpp_up1(v); }
__device__ __forceinline__ int shfl_dn1(int v) { return dpp_dn1(v); }
// a row's ids at (y-1, y, y+1) from the lanes' own ids (wave-uniform control flow only); the window's two end lanes see
// their own id beyond the window -- they only feed their neighbours, nothing reads what is derived from it
__device__ __forceinline__ Ids3 ids3_of(int c) { return Ids3{dpp_up1(c), c, dpp_dn1(c)}; }

}  // namespace
