// lbm_fused.hip -- the fused fluid kernels of the MI355X LBM-DEM stepper (gfx950, wave64): the PRODUCT kernels only.
// (The experiment build adds k_cs_march3 and the per-phase timers from lbm_fused_ab.hip / lbm_march_timing.h.)
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


#include "lbm_march.h"
#include <type_traits>
#if defined(LBMDEM_AB) && (defined(MARCH_TIMING) || defined(MARCH_TRACE))
#include "lbm_march_timing.h"
#else
#define MT_DECL
#define MT(i)
#define MT_FLUSH
#endif

#ifdef LBMDEM_AB
void launch_march3_ab(int which, int lx_template, const real* fin, real* fout, const int* obst_old, const int* obst_new,
                      const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int nstrips, int nwork, int remap,
                      int seg_rows, int seg_stride, int grid, hipStream_t st);   // lbm_fused_ab.hip
#endif

namespace {

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
  LBMDEM_GATE(L.gate);
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

// How the rows [xo0, xo1) of one launch are cut into work items (a window of WW columns x a segment of rows, one per
// wavefront). nlev == 0: uniform segments (seg_rows / seg_stride arguments of the kernel). nlev > 0: the rows are split into
// eight bands of band_rows rows, one per XCD (workgroup b runs on XCD b % 8), and every band is cut into rows[0] rows in
// segments of seg[0] rows, then rows[1] rows in segments of seg[1] rows, ...: LONG SEGMENTS FIRST, SHORT ONES LAST. The
// launch is bound by the memory system while all wave slots are busy (scripts/march_trace.py: ~390 row-equivalents per us with
// 1 800 or with 2 040 of the 2 048 slots busy), so what a schedule can lose is its tail: with uniform 32-row segments a slot
// idles for the last 125 us of 920 on average (14 % of the slot time); with the last rows of each band in 16- and 8-row
// segments 4.5 %.
// The band of XCD k is not one contiguous row range but the chunks k, k + 8, k + 16, ... of `chunk` rows each (a multiple of
// every segment length), so that a packing that fills only part of the lattice's length loads all eight XCDs alike.
struct MarchPlan {
  int nlev, band_rows, chunk;
  int seg[4];     // rows per segment of level l
  int off[4];     // first row of level l within the band
  int first[5];   // first item of level l within the band's item list; first[nlev] = items per band
};

// One work item of the marching kernel: window `strip`, rows [xs, xe). EDGE = false is the instantiation for items whose
// 64 columns and rows xs - 2 .. xe + 1 are all interior nodes of the lattice and rows of this slab (94 % of a 4096^2
// lattice's items): no clamp, no bounds test, no lattice-edge bounce-back anywhere in its loop -- and none of the registers
// they hold. The kernel picks the instantiation per wavefront.
constexpr int LINK_SLOTS = 64;   // link slots of a wavefront's compacted bounce-back evaluation

template <int WW, bool CHG, bool EDGE>
__device__ __forceinline__ void march_item(const real* __restrict__ fin, real* __restrict__ fout,
                                           const int* __restrict__ ob_old, const int* __restrict__ ob_new,
                                           const LatticeView& L, const LatticeView& Lk, const GrainFluidView& G,
                                           const ForceSlots& S, const ObstChange& CH, int strip, int xs, int xe,
                                           const RecRing ring,
                                           real* const pay, int* const desc, int lane, int w) {
  (void)w;   // (the experiment build's row trace indexes its buffer with it)
  // WW producing lanes in the middle of the window, (64 - WW) / 2 feeding lanes on either side
  constexpr int OFF = (64 - WW) / 2;
  const int y = strip * WW - OFF + lane;
  const bool yin = !EDGE || (y >= 0 && y < L.ly);
  const bool writer = lane >= OFF && lane < OFF + WW && yin;
  const bool deep_y = !EDGE || (strip * WW >= 2 && strip * WW + WW - 1 <= L.ly - 3);  // the producing lanes

  // Software pipeline. In iteration x (producing row x) the wave issues, in this order,
  //   (1) small gathers: new ids of row x+4, previous-map id of row x+4, the record of the grain that
  //       owned (x+2, y) before (reinit of row x+2), the record of the grain that owns (x+3, y) now
  //   (2) the nine populations of row x+3 into the row buffer it has just consumed (two buffers,
  //       ping-pong, loop unrolled by two: no register copies, so the loads stay in flight for two
  //       iterations)
  //   (3) the nine stores of row x.
  // gfx9 retires vector-memory operations in issue order (one vmcnt counter), so data must be consumed
  // in the order it was requested; no other global load exists inside the loop.
  auto row_ok = [&](int xl) { return !EDGE || (yin && xl >= 0 && xl < L.nxl); };
  const int ycl = !EDGE ? y : (y < 0 ? 0 : (y >= L.ly ? L.ly - 1 : y));
  // unconditional (clamped address): every use is guarded by interior(xl), and grain_rec clamps the
  // id. (A `row_ok ? v : -1` select here makes the compiler sink the load into a branch followed by
  // s_waitcnt vmcnt(0), which drains the whole prefetch pipeline once per iteration.)
  // The previous map's ids. With an ObstChange (CHG: the launch of a whole single-domain step after a rasterisation in
  // place) the wave knows the rows of its window in which the two maps differ -- a few per cent of them -- and takes the id
  // from the current map, which it holds anyway, everywhere else: the second map's 4 bytes per node were 2.2 % of the
  // kernel's traffic. The load stays in the instruction stream for every row (the in-order vmcnt bookkeeping of the
  // prefetch pipeline must not depend on data; under a branch it cost more than it saved): a buffer load whose offset
  // lies beyond the resource's records returns 0 and moves nothing. The rows a wave asks for are consecutive (xs - 1,
  // xs, ...): their bits sit in a 64-bit shift register in scalar registers, lowest bit = the next row, ones shifted in
  // (rows beyond the 64th are read from both maps).
  unsigned long long cm = ~0ull;
  // The four streams of the loop through buffer resources that start at the item's first row (lbm_march.h): a lane keeps
  // ONE column offset per stream, rows are scalar offsets.
  const int row0 = __builtin_amdgcn_readfirstlane(xs - 2 < 0 ? 0 : xs - 2);
  const int frow = L.sy * 9 * (int)sizeof(real), irow = L.sy * 4;   // bytes per row of populations / of ids
  const __amdgpu_buffer_rsrc_t fin_rs = make_rs(fin + (long)row0 * L.sy * 9, (long)(L.nxl - row0) * frow);
  const __amdgpu_buffer_rsrc_t fout_rs = make_rs(fout + (long)row0 * L.sy * 9, (long)(L.nxl - row0) * frow);
  const __amdgpu_buffer_rsrc_t id_rs = make_rs(ob_new + (long)row0 * L.sy, (long)(L.nxl - row0) * irow);
  const __amdgpu_buffer_rsrc_t old_rs = make_rs(ob_old + (long)row0 * L.sy, (long)(L.nxl - row0) * irow);
  const int fcol = fcol_bytes(ycl), icol = ycl * 4;
  auto rel_row = [&](int xl) { return (xl < 0 ? 0 : (xl >= L.nxl ? L.nxl - 1 : xl)) - row0; };   // clamped: always a valid row
  if (CHG) {
    const int r0 = xs - 1, first = r0 < 0 ? 0 : r0;
    const unsigned* wp = CH.bits + (long)strip * CH.words + (first >> 5);   // (a window's words are padded by four)
    const unsigned w0 = __builtin_amdgcn_readfirstlane(wp[0]), w1 = __builtin_amdgcn_readfirstlane(wp[1]);
    const unsigned w2 = __builtin_amdgcn_readfirstlane(wp[2]);
    const int sh = __builtin_amdgcn_readfirstlane(first & 31);
    cm = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
    if (sh) cm = (cm >> sh) | ((unsigned long long)w2 << (64 - sh));
    if (r0 < 0) cm <<= 1;   // row -1 is never used
  }
  auto load_old = [&](int xl, int same) -> int {
    if (!CHG) return __builtin_amdgcn_raw_buffer_load_b32(old_rs, icol, rel_row(xl) * irow, 0);
    const int keep = (cm & 1ull) ? 0 : -1;   // scalar: -1 = the maps agree in this row
    cm = (cm >> 1) | (1ull << 63);
    // (offset bit 31: beyond the resource's < 2^31 bytes, and far from wrapping round in the range check)
    const int v = __builtin_amdgcn_raw_buffer_load_b32(old_rs, (rel_row(xl) * irow + icol) | (keep & (int)0x80000000), 0, 0);
    return v | (same & keep);
  };
  // the current map's id of (xl, y); off the lattice (EDGE items only) reads as "wall": never fluid
  auto load_id = [&](int xl) -> int {
    const int v = __builtin_amdgcn_raw_buffer_load_b32(id_rs, icol, rel_row(xl) * irow, 0);
    if (!EDGE) return v;
    return (xl >= 0 && xl < L.nxl && ycl == y) ? v : L.n;
  };
#if defined(LBMDEM_AB) && defined(MARCH_NO_OLD)   /* timing experiment (wrong where the maps differ): the second map is not read */
#define MARCH_OLD(xl, same) (same)
#else
#define MARCH_OLD(xl, same) load_old(xl, same)
#endif
  // off-lattice positions load a clamped neighbour's values; they are never used (pull_one tests the
  // bounds of the source node before touching its populations)
  auto load_raw = [&](int xl, real (&raw)[9]) {
    const int so = rel_row(xl) * frow;
    raw[0] = buf_load_real<0 * F_QBYTES>(fin_rs, fcol, so); raw[1] = buf_load_real<1 * F_QBYTES>(fin_rs, fcol, so);
    raw[2] = buf_load_real<2 * F_QBYTES>(fin_rs, fcol, so); raw[3] = buf_load_real<3 * F_QBYTES>(fin_rs, fcol, so);
    raw[4] = buf_load_real<4 * F_QBYTES>(fin_rs, fcol, so); raw[5] = buf_load_real<5 * F_QBYTES>(fin_rs, fcol, so);
    raw[6] = buf_load_real<6 * F_QBYTES>(fin_rs, fcol, so); raw[7] = buf_load_real<7 * F_QBYTES>(fin_rs, fcol, so);
    raw[8] = buf_load_real<8 * F_QBYTES>(fin_rs, fcol, so);
  };
  auto interior = [&](int xl) {
    const int gx = L.gx0 + xl;
    return !EDGE || (row_ok(xl) && gx >= 1 && gx <= L.lx - 2 && y >= 1 && y <= L.ly - 2);
  };
  // f* of one node: reinit (previous map) + collide (current map)
  auto make_fstar = [&](int xl, real (&f)[9], int oo, const GPv& g, int on) {
    const bool in = interior(xl);
    if (in && oo != -1) grain_equilibrium(L, g, L.gx0 + xl, y, f);
    if (in && on == -1) mrt_collide(L, f);
  };
  auto grain_rec = [&](int id) { return load_gp(G, (id < 0 || id >= L.n) ? 0 : id); };
  auto reinit_rec = [&](int id) { return load_gpv(G, (id < 0 || id >= L.n) ? 0 : id); };

  real Fm[9], F0[9], Fp[9], bufA[9], bufB[9];

  // the current map's ids: one per lane and row; a node's column neighbours are the neighbour lanes
  const int iA = load_id(xs - 2);  // row x-2 (only needed for act of row x-1)
  int iB = load_id(xs - 1);        // row x-1
  int iC = load_id(xs);            // row x
  int iD = load_id(xs + 1);        // row x+1
  int iE = load_id(xs + 2);        // row x+2
  {
    int oo = MARCH_OLD(xs - 1, iB);
#if defined(LBMDEM_AB) && defined(MARCH_NO_HALO_ROWS)   /* timing experiment (wrong results): the two rows a segment shares with its neighbours are not read */
    load_raw(xs, Fm);
#else
    load_raw(xs - 1, Fm);
#endif
    make_fstar(xs - 1, Fm, oo, reinit_rec(oo), iB);
    oo = MARCH_OLD(xs, iC);
    load_raw(xs, F0);
    make_fstar(xs, F0, oo, reinit_rec(oo), iC);
  }
  // records of the current owners of rows x-1 .. x+2 into the ring
  ring.put(xs - 1, lane, grain_rec(iB), iB);
  ring.put(xs, lane, grain_rec(iC), iC);
  ring.put(xs + 1, lane, grain_rec(iD), iD);
  ring.put(xs + 2, lane, grain_rec(iE), iE);
  int inext = load_id(xs + 3);
  int oo1 = MARCH_OLD(xs + 1, iD);   // previous-map ids of rows x+1, x+2
  int oo2 = MARCH_OLD(xs + 2, iE);
  GP rec_next = grain_rec(inext);  // owner record of row x+3, goes into the ring next iteration
  GPv gv_next = ring.getv(xs + 1, lane);
  load_raw(xs + 1, bufA);
  load_raw(xs + 2, bufB);
  // The flags of a row as lane masks (lbm_march.h): sol* = the node belongs to a grain or a lattice-edge wall (id != -1),
  // act* = `act` (main.c:1039-1052). hm* = the highest id among a node and its two column neighbours.
  auto sol_of = [&](int c) -> lmask { return __ballot(c != -1); };
  auto hmax_of = [&](int c) { const int m = dpp_up1(c), p = dpp_dn1(c), t = m > c ? m : c; return t > p ? t : p; };
  // `act` of the row with ids b (a, c = the rows before and after): a solid node with a fluid neighbour -- all masks --
  // or, where reduced discs of different grains touch, with a neighbour that a HIGHER-index grain took from the fluid
  // (node_active; asked only when some solid node without a fluid neighbour has a higher id next to it, wave-uniform)
  auto act_of = [&](lmask sa, lmask sb, lmask sc, int hma, int hmb, int hmc, int a, int b, int c, int xl, int ycol) -> lmask {
    const lmask enclosed = sb & m_enclosed(sa, sb, sc);
    lmask act = sb & ~enclosed;
    const int t = hma > hmb ? hma : hmb;
    const lmask rare = __ballot((t > hmc ? t : hmc) > b) & enclosed;
    if (rare != 0) {
      const bool on = node_active(L, G, ids3_of(a), ids3_of(b), ids3_of(c), L.gx0 + xl, ycol, [&] { return ring.get(xl, lane); });
      act |= __ballot(on) & rare;
    }
    return act;
  };
  lmask solB = sol_of(iB), solC = sol_of(iC), solD = sol_of(iD), solE = sol_of(iE);
  int hmC = hmax_of(iC), hmD = hmax_of(iD), hmE = hmax_of(iE);
  lmask actM = act_of(sol_of(iA), solB, solC, hmax_of(iA), hmax_of(iB), hmC, iA, iB, iC, xs - 1, y);
  lmask act0 = act_of(solB, solC, solD, hmax_of(iB), hmC, hmD, iB, iC, iD, xs, y);
  const lmask wmask = EDGE ? __ballot(writer) : (((lmask)1 << WW) - 1) << OFF;

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
    // f* of row x+1. The record of a node's PREVIOUS owner (reinit_obst_density, main.c:966-986) is the ring's record of
    // its current one except at the few nodes that have changed hands since the previous map: only a row that has such a
    // node (wave-uniform) fetches records from memory.
    {
      const bool in = interior(x + 1), re = in && oo1 != -1, moved = re && oo1 != iD;
      GPv g = gv_next;   // (the ring's record of (x + 1, y), requested at the end of the iteration before)
      // (opaque: otherwise the compiler selects between the two ADDRESSES and reads through a flat pointer, under the
      // branch, with the wait that drains the row prefetch)
      asm volatile("" : "+v"(g.x1), "+v"(g.x2), "+v"(g.v1), "+v"(g.v2), "+v"(g.v3));
      if (__ballot(moved) != 0) {
        if (moved) g = reinit_rec(oo1);
      }
#if !(defined(LBMDEM_AB) && defined(MARCH_ABL_NOREINIT))   /* timing experiments (wrong results): a phase left out */
      if (re) grain_equilibrium(Lk, g, L.gx0 + x + 1, y, Fp);
#endif
#if !(defined(LBMDEM_AB) && defined(MARCH_ABL_NOCOLLIDE))
      if (in && iD == -1) mrt_collide(Lk, Fp);
#endif
    }
    MT(1)
    const int iF = inext;  // row x+3
    // ---- (1) small gathers
    oo1 = oo2;
    inext = load_id(x + 4);
    oo2 = MARCH_OLD(x + 3, iF);
    __builtin_amdgcn_sched_barrier(0);
    // ---- (2) the big loads: populations of row x+3
    // (rows beyond xe are never consumed: the last two prefetches of a segment re-request row xe, which the wave loaded
    // one or two iterations ago, instead of two new rows -- 2 of the 36 rows a 32-row segment would read, 2 of 12 for 8 rows;
    // the loads stay unconditional so that the in-order vmcnt bookkeeping of the pipeline does not change)
#if defined(LBMDEM_AB) && defined(MARCH_NO_HALO_ROWS)
    load_raw(x + 3 < xe ? x + 3 : xe - 1, buf);
#else
    load_raw(x + 3 < xe ? x + 3 : xe, buf);
#endif
    __builtin_amdgcn_sched_barrier(0);
    MT(2)
    // (the column re-declared opaque: otherwise (double)(y - 1), (double)y, (double)(y + 1) of node_active's rare path are
    // hoisted out of the loop into six registers the kernel does not have, and one pair ends up in scratch -- whose reload
    // drains the row prefetch)
    int y_act = y;
    asm volatile("" : "+v"(y_act));
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOACT)
    const lmask actP = solD;
#else
    const lmask actP = act_of(solC, solD, solE, hmC, hmD, hmE, iC, iD, iE, x + 1, y_act);
#endif
    MT(3)

    // the populations around P = (x, y): own row from the registers, the two column neighbours by DPP shifts
    real Fo[9], In[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) Fo[q] = F0[q];
    In[0] = 0.0;
    In[2] = Fm[6];            // (-1, 0): same lane, row x-1, slot opp(2) = 6
    In[6] = Fp[2];            // ( 1, 0)
    In[1] = shfl_dn1z(Fm[5]);  // (-1, 1): lane+1, row x-1, slot 5
    In[8] = shfl_dn1z(F0[4]);  // ( 0, 1)
    In[7] = shfl_dn1z(Fp[3]);  // ( 1, 1)
    In[3] = shfl_up1z(Fm[7]);  // (-1,-1): lane-1
    In[4] = shfl_up1z(F0[8]);  // ( 0,-1)
    In[5] = shfl_up1z(Fp[1]);  // ( 1,-1)
    {
      const int gx = L.gx0 + x;
      const int so_row = (x - row0) * frow;   // this row in the output lattice's resource
      // wave-uniform: is every producing lane of this row at least two nodes away from all edges?
      const bool deep = !EDGE || (deep_y && gx >= 2 && gx <= L.lx - 3);
      // (a) Per direction q three lane masks: lk = (P, q) is a bounce-back link (P fluid, its source S = P - e_q a grain
      // node), hz = the node two links out, NN = P + e_q, is solid and its own update precedes S's (q <= 4), rs = both ends
      // solid and the source active: the slot takes the weight w_q (main.c:1161-1162). Formed where they are used -- three
      // scalar instructions each -- instead of being carried: two dozen register pairs would not fit.
      const lmask wm = x < xe ? wmask : 0;
#define MARCH_SRC(Q) m_at<-EYq(Q)>(MARCH_ROW(-EXq(Q), solB, solC, solD))   /* S = P - e_q is solid */
#define MARCH_LINK(Q) (wm & ~solC & MARCH_SRC(Q))
#define MARCH_HAZ(Q) (Q <= 4 ? MARCH_LINK(Q) & m_at<EYq(Q)>(MARCH_ROW(EXq(Q), solB, solC, solD)) : 0)
#define MARCH_RESET(Q) (solC & MARCH_SRC(Q) & m_at<-EYq(Q)>(MARCH_ROW(-EXq(Q), actM, act0, actP)))
      unsigned ibb = 0, nnm = 0, hzm = 0;   // the same per lane, bit q, where the masks do not apply (edge rows)
      int T = 0;
      if (deep) {
        T = __popcll(MARCH_LINK(1)) + __popcll(MARCH_LINK(2)) + __popcll(MARCH_LINK(3)) + __popcll(MARCH_LINK(4)) +
            __popcll(MARCH_LINK(5)) + __popcll(MARCH_LINK(6)) + __popcll(MARCH_LINK(7)) + __popcll(MARCH_LINK(8));
      } else if constexpr (EDGE) {
        // rows and windows next to a lattice edge: the general classification, per lane; everything but the links is stored
        RegCtx C;
        C.ring = ring;
        C.row = x;
        C.lane = lane;
#pragma unroll
        for (int q = 0; q < 9; ++q) { C.Fo[q] = Fo[q]; C.In[q] = In[q]; }
        const Ids3 b3 = ids3_of(iB), c3 = ids3_of(iC), d3 = ids3_of(iD);
        C.o0 = iC;
        C.onb[0] = 0;
        C.onb[1] = b3.p; C.onb[2] = b3.c; C.onb[3] = b3.m; C.onb[4] = c3.m;
        C.onb[5] = d3.m; C.onb[6] = d3.c; C.onb[7] = d3.p; C.onb[8] = c3.p;
        C.act = 0;
#define MARCH_ACTBIT(D) C.act |= lane_of(m_at<EYq(D)>(MARCH_ROW(EXq(D), actM, act0, actP))) ? 1u << D : 0u;
        MARCH_ACTBIT(1) MARCH_ACTBIT(2) MARCH_ACTBIT(3) MARCH_ACTBIT(4) MARCH_ACTBIT(5) MARCH_ACTBIT(6) MARCH_ACTBIT(7) MARCH_ACTBIT(8)
#undef MARCH_ACTBIT
        if (writer && x < xe) classify_store_row<true>(C, L, gx, y, fout, fbase_xy(L, x, y), ibb, nnm, hzm);
#pragma unroll
        for (int q = 1; q < 9; ++q) T += __popcll(__ballot((ibb >> q) & 1u));
      }
      MT(4)
      // (b) The bounce-back links of the row (typically ~20, spread over all eight directions and a few lanes) are compacted
      // into dense lanes through LDS and evaluated in ONE pass with the direction as data, instead of ~3.5
      // direction-specific divergent passes of ~130 instructions. slot of link (lane, q) = number of links in directions
      // < q + number in direction q on lower lanes. In a deep row with at most 64 links the pass leaves its results in the
      // links' slots and the lanes pick them up for the row's nine coalesced stores (`merged`); else every population but
      // the links' is stored first and the pass stores each result to the node it belongs to.
#if defined(LBMDEM_AB) && defined(MARCH_SCATTER)   /* experiment: every result stored by the lane that evaluated it (+13 us) */
      const bool merged = false;
#else
      const bool merged = deep && T <= LINK_SLOTS;
#endif
      auto store_row = [&](auto with_results) {   // the nine stores of a deep row
        if (lane_of(wm)) {
          buf_store_real<0>(Fo[0], fout_rs, fcol, so_row);
          // (all eight results requested first, then ONE wait: a read and its wait per direction cost the row a
          // thousand cycles)
          real res[9];
          if constexpr (decltype(with_results)::value) {
            int before = 0;
#define MARCH_RES(Q)                                                                               \
            {                                                                                      \
              const lmask b = MARCH_LINK(Q);                                                       \
              res[Q] = pay[((before + (int)mbcnt(b)) & (LINK_SLOTS - 1)) * 4];                     \
              before += __popcll(b);                                                               \
            }
            MARCH_RES(1) MARCH_RES(2) MARCH_RES(3) MARCH_RES(4) MARCH_RES(5) MARCH_RES(6) MARCH_RES(7) MARCH_RES(8)
#undef MARCH_RES
          }
#define MARCH_STORE(Q)                                                                             \
          {                                                                                        \
            real v = lane_of(MARCH_RESET(Q)) ? Wq(Q) : In[OPPq(Q)];                                \
            if constexpr (decltype(with_results)::value) v = lane_of(MARCH_LINK(Q)) ? res[Q] : v;  \
            buf_store_real<Q * F_QBYTES>(v, fout_rs, fcol, so_row);                                \
          }
          MARCH_STORE(1) MARCH_STORE(2) MARCH_STORE(3) MARCH_STORE(4) MARCH_STORE(5) MARCH_STORE(6) MARCH_STORE(7) MARCH_STORE(8)
#undef MARCH_STORE
        }
      };
      if (deep && !merged) store_row(std::false_type{});   // (a link's slot gets the streamed value as a placeholder; the stores of one
                                               // wavefront to one address keep their order, and a link whose wall distance
                                               // fires neither formula keeps exactly this value, main.c:1166-1217)
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOPASS)
      T = 0;
#endif
      MT(12)
      for (int base = 0; base < T; base += LINK_SLOTS) {  // wave-uniform; a second round only if > 64 links
        int before = 0;
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOCOMPACT)
        if (false) {
#else
        if (deep) {
#endif
#define MARCH_COMPACT(Q)                                                                           \
          {                                                                                        \
            const lmask b = MARCH_LINK(Q);                                                         \
            const int t = before + (int)mbcnt(b) - base;                                           \
            before += __popcll(b);                                                                 \
            if (lane_of(b) && (unsigned)t < (unsigned)LINK_SLOTS) {                                \
              desc[t] = lane | (Q << 8) | (1 << 12) | (lane_of(MARCH_HAZ(Q)) ? 1 << 13 : 0);       \
              pay[t * 4 + 0] = Fo[OPPq(Q)];                                                        \
              pay[t * 4 + 1] = Fo[Q];                                                              \
              pay[t * 4 + 2] = In[Q];                                                              \
              pay[t * 4 + 3] = In[OPPq(Q)];                                                        \
            }                                                                                      \
          }
          MARCH_COMPACT(1) MARCH_COMPACT(2) MARCH_COMPACT(3) MARCH_COMPACT(4)
          MARCH_COMPACT(5) MARCH_COMPACT(6) MARCH_COMPACT(7) MARCH_COMPACT(8)
#undef MARCH_COMPACT
        } else if constexpr (EDGE) {
#pragma unroll
          for (int q = 1; q < 9; ++q) {
            const lmask b = __ballot((ibb >> q) & 1u);
            const int t = before + (int)mbcnt(b) - base;
            before += __popcll(b);
            if (((ibb >> q) & 1u) && (unsigned)t < (unsigned)LINK_SLOTS) {
              desc[t] = lane | (q << 8) | (((nnm >> q) & 1u) << 12) | (((hzm >> q) & 1u) << 13);
              pay[t * 4 + 0] = Fo[OPPq(q)];
              pay[t * 4 + 1] = Fo[q];
              pay[t * 4 + 2] = In[q];
              pay[t * 4 + 3] = In[OPPq(q)];
            }
          }
        }
        __builtin_amdgcn_wave_barrier();  // LDS operations of one wave execute in order
        MT(10)
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOEVALPASS)
        if (false) {
#else
        if (base + lane < T) {
#endif
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
          const int ex = (k.q >= 1 && k.q <= 3) ? -1 : ((k.q >= 5 && k.q <= 7) ? 1 : 0);
          const int ey = (k.q == 1 || k.q >= 7) ? 1 : ((k.q >= 3 && k.q <= 5) ? -1 : 0);
          // the record of the grain that owns S = P - e_q, and which grain that is (one round of LDS reads)
          const GP gS = ring.get(x - ex, src - ey);
          const int owner = ring.get_id(x - ex, src - ey);
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOEVAL)
          const real out = k.in_qo;
#else
          const real out = ibb_eval_rt(Lk, k, Lk.wc_diag, Lk.wc_axis, gS, [&] { return ring.get(x + ex, src + ey); });
#endif
          if (merged) pay[lane * 4] = out;   // picked up by the lane of node `src` for plane q of the row's stores
#if !(defined(LBMDEM_AB) && defined(MARCH_ABL_NOFSTORE))
          else buf_store_real<0>(out, fout_rs, fcol_bytes(k.gy) + k.q * F_QBYTES, so_row);
#endif
          // ... and the link's momentum-exchange sum f_new[S][opp q] + f_new[P][q] (main.c:1313-1316; the first
          // is f*[P][opp q], streamed unchanged into the solid node) to the slot table of the grain that owns S
#if defined(LBMDEM_AB) && defined(MARCH_ABL_NOTAB)
          if (false) {
#else
          if (S.tab != nullptr) {
#endif
            const int rel = slot_line(k.gx - ex, k.gy - ey, ex, ey, gS.xc, gS.yc) + S.half;
            if ((unsigned)rel < (unsigned)S.spd)
#if defined(LBMDEM_AB) && defined(MARCH_TAB_NT)   /* experiment: the link sums past the L2's retention */
              __builtin_nontemporal_store(k.own_qo + out, &S.tab[((long)owner * 8 + (k.q - 1)) * S.spd + rel]);
#else
              S.tab[((long)owner * 8 + (k.q - 1)) * S.spd + rel] = k.own_qo + out;
#endif
          }
        }
        __builtin_amdgcn_wave_barrier();
        MT(11)
      }
      if (merged) {
        if (T > 0) store_row(std::true_type{});
        else store_row(std::false_type{});
      }
#undef MARCH_RESET
#undef MARCH_HAZ
#undef MARCH_LINK
#undef MARCH_SRC
      MT(5)
    }
    // row x-1 is no longer needed: its ring slot takes the owner records of row x+3; then request
    // those of row x+4 (consumed at this point of the next iteration)
    ring.put(x + 3, lane, rec_next, iF);
    rec_next = grain_rec(inext);
    gv_next = ring.getv(x + 2, lane);   // position and velocities of the owner of (x + 2, y): the next iteration's re-initialisation
    // rotate
#pragma unroll
    for (int q = 0; q < 9; ++q) { Fm[q] = F0[q]; F0[q] = Fp[q]; }
    iB = iC; iC = iD; iD = iE; iE = iF;
    solB = solC; solC = solD; solD = solE; solE = sol_of(iF);
    hmC = hmD; hmD = hmE; hmE = hmax_of(iF);
    actM = act0; act0 = actP;
    MT(6)
  };

  // Both halves run unconditionally (a row >= xe stores nothing): with `if (x + 1 < xe)` around the second one the
  // compiler cannot count its loads as younger than the first half's when it places s_waitcnt at the loop head, and
  // waits for more of the pipeline than the data it needs (vmcnt(4) instead of vmcnt(9); 1.5-2 % of the kernel).
  for (int x = xs; x < xe; x += 2) {
    iterate(x, bufA);
    iterate(x + 1, bufB);
  }
  MT_FLUSH
}

// wavefronts per workgroup of k_cs_march (they share nothing: the workgroup is the unit the dispatcher places and retires)
#ifndef MARCH_WPB
#define MARCH_WPB 4
#endif
constexpr int WPB = MARCH_WPB;

template <int LX, int MINW, int WW, bool CHG = false>
__global__ __launch_bounds__(64 * WPB, MINW) void k_cs_march(const real* __restrict__ fin, real* __restrict__ fout,
                                                  const int* __restrict__ ob_old,
                                                  const int* __restrict__ ob_new, LatticeView L,
                                                  GrainFluidView G, ForceSlots S, int nstrips, int nwork,
                                                  int xcd_remap, int seg_rows, int seg_stride, MarchPlan P, ObstChange CH) {
  LBMDEM_GATE(L.gate);
  const int lane = threadIdx.x & 63;
  int blk = blockIdx.x;
  int w, strip, xs, rows_per_wave;
  if (LX == 0 && P.nlev > 0) {
    const int band = blk & 7, local = (blk >> 3) * WPB + (threadIdx.x >> 6);
    if (local >= P.first[P.nlev]) return;  // whole wave
    int l = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k)
      if (k < P.nlev && local >= P.first[k]) l = k;
    const int j = local - P.first[l], sg = j / nstrips;
    strip = j - sg * nstrips;
    rows_per_wave = P.seg[l];
    const int v = P.off[l] + sg * rows_per_wave, c = v / P.chunk;   // row within the band -> chunk c of this XCD
    xs = L.xo0 + (c * 8 + band) * P.chunk + (v - c * P.chunk);
    if (xs >= L.xo1) return;               // a band that ends beyond the row range (row counts that are no multiple of 8 bands)
    w = band * P.first[P.nlev] + local;
  } else {
    if (xcd_remap) {  // XCD k (blocks b % 8 == k) walks the k-th contiguous eighth of the work list
      const int per = gridDim.x >> 3;
      blk = (blk & 7) * per + (blk >> 3);
    }
    w = blk * WPB + (threadIdx.x >> 6);
    if (w >= nwork) return;  // whole wave
    const int seg = w / nstrips;
    strip = w - seg * nstrips;
    // rows per wave: the template value, or (LX == 0) a run-time value
    rows_per_wave = LX > 0 ? LX : seg_rows;
    // segment k starts seg_stride rows after segment k-1: = rows_per_wave for a contiguous row range; larger when one
    // launch covers the two edge-row ranges of a strip
    xs = L.xo0 + seg * seg_stride;
  }
  // (the item is the wavefront's: window, rows and everything derived from them live in scalar registers)
  strip = __builtin_amdgcn_readfirstlane(strip);
  xs = __builtin_amdgcn_readfirstlane(xs);
  const int xe = __builtin_amdgcn_readfirstlane(xs + rows_per_wave < L.xo1 ? xs + rows_per_wave : L.xo1);
  __shared__ real2 sRec[WPB * REC_RING * 4 * 64];
  __shared__ int sRid[WPB * REC_RING * 64];
  const RecRing ring{sRec + (threadIdx.x >> 6) * (REC_RING * 4 * 64), sRid + (threadIdx.x >> 6) * (REC_RING * 64)};
  // wave-private scratch for the compacted bounce-back evaluation: 64 link slots
  // (Round 3 measured a spare 65th slot that lanes without a link write to, instead of sitting out the writes under an
  // exec mask: eight mask round trips per row less, but 12 B of scratch and twice the scalar spill reloads: +1.7 %.)
  __shared__ real sPay[WPB * LINK_SLOTS * 4];
  __shared__ int sDesc[WPB * LINK_SLOTS];
  real* const pay = sPay + (threadIdx.x >> 6) * (LINK_SLOTS * 4);
  int* const desc = sDesc + (threadIdx.x >> 6) * LINK_SLOTS;
  // The lattice's constants once more in LDS, wave-private: the arithmetic routines (collision, grain equilibrium,
  // bounce-back) read their reals from there, at the point of use, through the vector unit -- thirteen doubles that would
  // otherwise sit in 26 scalar registers for the whole loop, in a kernel whose scalar registers are spilled to vector lanes.
  __shared__ LatticeView sLat[WPB];
  LatticeView* const Lk = sLat + (threadIdx.x >> 6);
  if (lane == 0) *Lk = L;
  __builtin_amdgcn_wave_barrier();
  // Is the whole item -- its 64 columns, its rows and the two rows either side -- interior? (wave-uniform)
  constexpr int OFF = (64 - WW) / 2;
  const int y_lo = strip * WW - OFF;
  const bool inner = y_lo >= 1 && y_lo + 63 <= L.ly - 2 && L.gx0 + xs >= 2 && L.gx0 + xe <= L.lx - 2 && xs >= 2 &&
                     xe + 2 <= L.nxl;
  if (inner) march_item<WW, CHG, false>(fin, fout, ob_old, ob_new, L, *Lk, G, S, CH, strip, xs, xe, ring, pay, desc, lane, w);
  else march_item<WW, CHG, true>(fin, fout, ob_old, ob_new, L, *Lk, G, S, CH, strip, xs, xe, ring, pay, desc, lane, w);
}

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

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
  seg_rows &= ~1;
  if (seg_rows < 8) seg_rows = 8;
  if (seg_rows > 32) seg_rows = 32;
  return seg_rows;
}

// The tapered schedule of a LARGE row range (MarchPlan above): eight XCD bands, each cut into long main segments followed by
// `tail` levels of shorter ones. Returns a plan with nlev == 0 when the range is too small for it (short ranges run uniform
// short segments: march_segment_rows).
static MarchPlan march_plan(int rows, int nstrips, int main_seg, const int (*tail)[2], int ntail, int chunk) {
  MarchPlan P{};
  if (rows < 8 * (main_seg + 64)) return P;
  if (chunk <= 0) chunk = 1 << 30;                         // one contiguous band per XCD
  int band_rows = (((rows + 7) / 8) + 7) & ~7;             // a multiple of 8: every level below is cut into whole segments
  if (chunk < band_rows) band_rows = (band_rows + chunk - 1) / chunk * chunk;
  else chunk = band_rows;
  P.chunk = chunk;
  int tail_rows = 0;
  for (int k = 0; k < ntail; ++k) tail_rows += tail[k][1];
  if (band_rows - tail_rows < main_seg) return P;
  const int main_rows = (band_rows - tail_rows) / main_seg * main_seg;
  int rem = band_rows - tail_rows - main_rows;            // < main_seg, a multiple of 8: joins the tail levels it fits into
  P.band_rows = band_rows;
  int off = 0, first = 0, lev = 0;
  bool fits = true;   // every segment inside one chunk
  auto add = [&](int seg, int r) {
    if (r <= 0) return;
    if (chunk % seg != 0 || off % seg != 0) fits = false;
    P.seg[lev] = seg; P.off[lev] = off; P.first[lev] = first;
    off += r; first += r / seg * nstrips; ++lev;
  };
  add(main_seg, main_rows);
  for (int k = 0; k < ntail; ++k) {
    int r = tail[k][1];
    const int seg = tail[k][0];
    const int take = k + 1 < ntail ? rem / seg * seg : rem;   // the last (shortest) level takes what is left
    r += take; rem -= take;
    add(seg, r);
  }
  if (rem != 0 || !fits) return MarchPlan{};   // (no tail level to take the remainder, ...: uniform segments instead)
  P.nlev = lev;
  P.first[lev] = first;
  return P;
}

// The work order the product uses for `rows` rows of `nstrips` windows: tapered for a large range (uniform segments would be
// 32 rows long), else none (nlev == 0: uniform segments of seg_rows rows).
static MarchPlan product_plan(int rows, int nstrips, int seg_rows) {
  MarchPlan P{};
  if (seg_rows < 32) return P;
  {
    // Round 5: 64-row main segments and a 32-row level in front of the 16- and 8-row ones. A wavefront's rows do not get
    // dearer with the segment length (profiles/r05_segment_length_counters.txt: wave-cycles +0.7 % at 64 rows, +2 % at 128),
    // and every segment re-reads two prologue rows: -3 % of the read requests, fused kernel -0.4 % (interleaved A/B);
    // 128-row segments lose 19 % to the end of the launch.
    int main_seg = 64, ntail = 3;
    int tail[3][2] = {{32, 64}, {16, 64}, {8, 64}};   // {rows per segment, rows of every band cut that way}
#ifdef LBMDEM_AB   // LBMDEM_PLAN="seg:rows,seg:rows[,seg:rows]" (tail levels), LBMDEM_CS_ROWS = the main segment length
    if (getenv("LBMDEM_CS_ROWS")) main_seg = atoi(getenv("LBMDEM_CS_ROWS"));
    if (const char* e = getenv("LBMDEM_PLAN")) {
      ntail = sscanf(e, "%d:%d,%d:%d,%d:%d", &tail[0][0], &tail[0][1], &tail[1][0], &tail[1][1], &tail[2][0], &tail[2][1]) / 2;
    }
#endif
    bool ok = main_seg >= 8 && main_seg % 8 == 0;
    for (int k = 0; k < ntail; ++k) ok = ok && tail[k][0] >= 8 && tail[k][0] % 8 == 0 && tail[k][1] % tail[k][0] == 0;
    int chunk = 64;
#ifdef LBMDEM_AB
    if (getenv("LBMDEM_CHUNK")) chunk = atoi(getenv("LBMDEM_CHUNK"));
#endif
    if (chunk > 0 && chunk % main_seg != 0) chunk = 0;
    if (ok) P = march_plan(rows, nstrips, main_seg, tail, ntail, chunk);
  }
  return P;
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

// Producing lanes per 64-lane window. 60: a window's stores start at y = 60 k, i.e. on a 32-byte sector boundary of the
// [y/16][q][y%16] tile rows, and cover 15 whole sectors -- with 62 (rounds 1-4) the two sectors at every window seam were
// written half by one wavefront and half by its neighbour, at different times (two byte-masked writes per sector for the L2
// to merge: 57.4 M written sectors per launch against 56.2 M, profiles/r04_g_sq_counters.txt). Interleaved A/B on one
// GPU, product work order: 62: 0.7197 / 0.7209 ms, 60: 0.7128 / 0.7113, 56 (64-byte aligned): 0.7172 / 0.7176 (the wider
// halo costs more waves than the alignment saves).
constexpr int MARCH_WW = 60;

template <int LX, int MINW, int WW = MARCH_WW>
static void launch_march(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                         const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int remap,
                         hipStream_t st, ObstChange chg = ObstChange{nullptr, 0, 0, 0}) {
  const int rows = L.xo1 - L.xo0;
  const int nstrips = (L.ly + WW - 1) / WW;
  if (chg.ww != WW || chg.off != (64 - WW) / 2) chg = ObstChange{nullptr, 0, 0, 0};   // (bits cut for other windows)
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
  int nwork = nstrips * nseg;
  int grid = (nwork + WPB - 1) / WPB;
  if (remap) grid = ((grid + 7) / 8) * 8;
  MarchPlan P{};
  if (LX == 0 && remap) P = product_plan(rows, nstrips, seg_rows);   // a large row range: the tapered work order
  if (P.nlev > 0) {
    nwork = 8 * P.first[P.nlev];
    grid = 8 * ((P.first[P.nlev] + WPB - 1) / WPB);
  }
#ifdef LBMDEM_AB   // k_cs_march3 (lbm_fused_ab.hip) only exists in the experiment build
  if constexpr (WW == 62) {
    const int mk = march_kernel();
    if (mk == 3 || mk == 21 || mk == 22) {
      launch_march3_ab(mk, LX, fin, fout, obst_old, obst_new, L, G, S, nstrips, nwork, remap, seg_rows, seg_rows, grid, st);
      return;
    }
  }
#endif
  unsigned dyn_lds = 0;
#ifdef LBMDEM_AB   // occupancy experiment: extra (unused) dynamic LDS so that only ONE workgroup fits a CU
  static const int env_lds = getenv("LBMDEM_MARCH_DYNLDS") ? atoi(getenv("LBMDEM_MARCH_DYNLDS")) : 0;
  dyn_lds = (unsigned)env_lds;
#endif
  if (LX == 0 && chg.bits != nullptr) {
    hipLaunchKernelGGL((k_cs_march<LX, MINW, WW, LX == 0>), dim3(grid), dim3(64 * WPB), dyn_lds, st, fin, fout, obst_old, obst_new, L, G,
                       S, nstrips, nwork, remap, seg_rows, seg_rows, P, chg);
    return;
  }
  hipLaunchKernelGGL((k_cs_march<LX, MINW, WW>), dim3(grid), dim3(64 * WPB), dyn_lds, st, fin, fout, obst_old, obst_new, L, G,
                     S, nstrips, nwork, remap, seg_rows, seg_rows, P, chg);
}

// Two row ranges of equal width w <= 32 (the rows next to the two cuts of a strip) in ONE launch: two segments of w rows,
// the second `stride` rows after the first.
static void launch_march_two_ranges(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                                    const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int lo0, int w,
                                    int hi0, hipStream_t st) {
  constexpr int WW = MARCH_WW;
  LatticeView Ls = L;
  Ls.xo0 = lo0; Ls.xo1 = hi0 + w;
  const int nstrips = (L.ly + WW - 1) / WW;
  const int nwork = nstrips * 2;
  const int grid = (nwork + WPB - 1) / WPB;
#ifdef LBMDEM_AB
  if (march_kernel() == 3) {   // (k_cs_march3 is laid out for 62-column windows)
    const int ns3 = (L.ly + 61) / 62, nw3 = ns3 * 2;
    launch_march3_ab(3, 0, fin, fout, obst_old, obst_new, Ls, G, S, ns3, nw3, 0, w, hi0 - lo0, (nw3 + 3) / 4, st);
    return;
  }
#endif
  hipLaunchKernelGGL((k_cs_march<0, 2, WW>), dim3(grid), dim3(64 * WPB), 0, st, fin, fout, obst_old, obst_new, Ls, G, S, nstrips,
                     nwork, 0, w, hi0 - lo0, MarchPlan{}, ObstChange{nullptr, 0, 0, 0});
}

// The marching kernel assumes reductionR < 1 (always true in the reference); other configurations run the
// LDS-tile kernel, which does not fill the slot table.
bool collide_stream_fills_slots(const LatticeView& L) { return L.reduced_lt1 != 0; }

void collide_stream_work_order(const LatticeView& L, int* info) {
  for (int k = 0; k < 12; ++k) info[k] = 0;
  if (!L.reduced_lt1) return;   // the LDS-tile kernel: no marching work items
  const int rows = L.xo1 - L.xo0, nstrips = (L.ly + MARCH_WW - 1) / MARCH_WW;
  const int seg_rows = march_segment_rows(rows, nstrips);
  const MarchPlan P = product_plan(rows, nstrips, seg_rows);
  info[0] = P.nlev;
  if (P.nlev == 0) {
    info[3] = seg_rows;
    info[7] = rows;
    info[11] = nstrips * ((rows + seg_rows - 1) / seg_rows);
    return;
  }
  info[1] = P.band_rows; info[2] = P.chunk;
  for (int l = 0; l < P.nlev; ++l) {
    info[3 + l] = P.seg[l];
    info[7 + l] = (l + 1 < P.nlev ? P.off[l + 1] : P.band_rows) - P.off[l];
  }
  info[11] = 8 * P.first[P.nlev];
}

void collide_stream_windows(int* ww, int* off) { *ww = MARCH_WW; *off = (64 - MARCH_WW) / 2; }

void launch_collide_stream(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                           const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, hipStream_t st,
                           const ObstChange& chg) {
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
      case 22: launch_march<0, 2, 62>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;   // rounds 1-4: 62 producing lanes, product work order
      case 23: launch_march<0, 2, 56>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;   // 64-byte aligned stores, product work order
      default: launch_march<32, 2>(fin, fout, obst_old, obst_new, L, G, S, remap, st); return;
    }
  }
#endif
  if (L.reduced_lt1) {
#ifdef LBMDEM_AB
    if (march_kernel() != 2) {   // k_cs_march3 (62-column windows only)
      launch_march<0, 2, 62>(fin, fout, obst_old, obst_new, L, G, S, /*xcd remap*/ 1, st);
      return;
    }
#endif
    launch_march<0, 2>(fin, fout, obst_old, obst_new, L, G, S, /*xcd remap*/ 1, st, chg);
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

