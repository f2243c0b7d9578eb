// lbm_fused_ab.hip -- EXPERIMENT BUILD ONLY (make AB=1 -> liblbmdem_hip_ab.so; nothing of this file is in the product):
// k_cs_march3, the three-waves-per-SIMD form of the marching kernel (round 3: measured, not faster -- DESIGN.md section 4),
// with its compile-time switches M3_DMAPOP / M3_UNIFORM / M3_MANUAL / M3_OVERFLOW, reached through LBMDEM_MARCH=3 | 21 | 22.
#ifdef LBMDEM_AB

#include "lbm_march.h"

namespace {

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
        const real out = ibb_eval_rt(L, k, wc_diag, wc_axis, g, [&] {
          // the hazard partner: the grain that owns NN = P + e_q (rare; its loads stay inside this branch)
          return load_gp(G, ob_new[(long)(x + ex) * L.sy + (k.gy + ey)]);
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

}  // namespace

// called by launch_march / launch_march_two_ranges (lbm_fused.hip) for LBMDEM_MARCH = 3 (three waves per SIMD), 21, 22
void launch_march3_ab(int which, int lx_template, const real* fin, real* fout, const int* obst_old, const int* obst_new,
                      const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int nstrips, int nwork, int remap,
                      int seg_rows, int seg_stride, int grid, hipStream_t st) {
#define M3_LAUNCH(LXT, MINW, NBUF)                                                                                  \
  hipLaunchKernelGGL((k_cs_march3<LXT, MINW, 62, NBUF>), dim3(grid), dim3(256), 0, st, fin, fout, obst_old, obst_new, L, G, \
                     S, nstrips, nwork, remap, seg_rows, seg_stride)
#define M3_PICK(MINW, NBUF)                                                              \
  switch (lx_template) {                                                                 \
    case 16: M3_LAUNCH(16, MINW, NBUF); break;                                           \
    case 32: M3_LAUNCH(32, MINW, NBUF); break;                                           \
    case 64: M3_LAUNCH(64, MINW, NBUF); break;                                           \
    default: M3_LAUNCH(0, MINW, NBUF); break;                                            \
  }
  if (which == 3) { M3_PICK(3, 1) }
  else if (which == 21) { M3_PICK(2, 1) }
#ifndef M3_DMAPOP
  else if (which == 22) { M3_PICK(2, 2) }
#endif
#undef M3_PICK
#undef M3_LAUNCH
}

#endif  // LBMDEM_AB
