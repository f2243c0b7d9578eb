// lbm_forces.hip -- hydrodynamic force and torque on the grains (forces_fluid, main.c:1285-1333): the link-sum table route
// (k_forces_table + k_forces_gather_queue), the gather routes (k_forces_parity, k_forces_fast), the table's reset.

#include "lbm_device.h"

namespace {

__global__ void k_fill_u64(unsigned long long* __restrict__ p, long count, unsigned long long v) {
  for (long k = (long)blockIdx.x * blockDim.x + threadIdx.x; k < count; k += (long)gridDim.x * blockDim.x) p[k] = v;
}


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
  LBMDEM_GATE(L.gate);
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

#ifndef LBMDEM_FT_WAVES
#define LBMDEM_FT_WAVES 4
#endif
constexpr int FT_WAVES = LBMDEM_FT_WAVES;  // waves per workgroup of k_forces_table

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
                                                                const unsigned char* __restrict__ mask, int list_cap,
                                                                ObstFillJob fill, int nfill) {
  LBMDEM_GATE(L.gate);
  extern __shared__ real sDyn[];
  // the first `nfill` workgroups reset the obstacle map the next rasterisation starts from (the fused kernel has just
  // finished with it; this kernel is bound by its arithmetic, the 67 MB of stores disappear beside it)
  if ((int)blockIdx.x < nfill) {
    obst_fill_range(fill.map, fill.L, (long)blockIdx.x * blockDim.x + threadIdx.x, (long)nfill * blockDim.x, fill.row0, fill.row1);
    return;
  }
  const int bx = (int)blockIdx.x - nfill;
  __shared__ __attribute__((aligned(16))) real sZero[8];   // phase B: what a lane adds once its own list has ended
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
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // (uniform: the grain's scalars come through the scalar unit)
  const size_t lists_doubles = (size_t)FT_WAVES * GW * per_grain;
  unsigned long long* const bm = reinterpret_cast<unsigned long long*>(sDyn + lists_doubles) + (size_t)wave * nw64 * 3;
  int* const pw = reinterpret_cast<int*>(bm + nw64);  // 3 * nw64 ints = 1.5 * nw64 words: the wave's 3 * nw64 words hold both
  int* const counts = reinterpret_cast<int*>(reinterpret_cast<unsigned long long*>(sDyn + lists_doubles) +
                                             (size_t)FT_WAVES * nw64 * 3);
  const int lane = threadIdx.x & 63;
  const int gslot0 = wave * GW;                       // this wave's grains within the workgroup
  const int g0 = (bx * FT_WAVES + wave) * GW; // position in the list (or the grain index itself without a list)
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
  // ---- per grain: everything the wave needs from memory in ONE round -- the disc and the table rows. (The disc used to be
  // re-read in phase A: the compiler cannot carry G.xc[i] across the table's reset stores, which may alias it for all it
  // knows, so every wave paid a second dependent round trip before its first square root.) The table loads do not wait
  // for `local` either: any grain's rows are valid addresses; only the reset stores are conditional.
  real gxc[GW], gyc[GW], gr2[GW], grb[GW];
  bool gtouched[GW];
#pragma unroll
  for (int g = 0; g < GW; ++g) {
    gid[g] = g0 + g < ntodo ? grain_at(g0 + g) : -1;
    const int i = gid[g], ii = i >= 0 ? i : 0;
    gxc[g] = G.xc[ii]; gyc[g] = G.yc[ii]; gr2[g] = G.r2[ii]; grb[g] = G.rbl0[ii];
    gtouched[g] = S.touched[ii] != 0;
    unsigned long long* tg = reinterpret_cast<unsigned long long*>(S.tab) + (long)ii * 8 * spd;
    unsigned long long tf[PASSES], tb[PASSES];
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int fm = j * FPP + lane / LPF;
      const int q = fam_q(fm), qo = q - 4;
      const int rb = S.half - c;  // the same line seen from the opposite direction
      tf[j] = tb[j] = LBMDEM_SLOT_EMPTY;
      if (lane_has_line) {
        tf[j] = tg[(q - 1) * spd + rel];
        if (rb >= 0 && rb < spd) tb[j] = tg[(qo - 1) * spd + rb];
      }
    }
    bool local = false;
    if (i >= 0 && mask) {
      local = mask[i] != 0;      // strip decomposition: the grains the rasteriser saw (the others' geometry is stale)
      if (!local) gid[g] = -1;
    } else if (i >= 0) {
      local = gxc[g] + grb[g] + 2.0 >= (real)L.gx0 && gxc[g] - grb[g] - 2.0 <= (real)(L.gx0 + L.nxl);
    }
#pragma unroll
    for (int j = 0; j < PASSES; ++j) {
      const int fm = j * FPP + lane / LPF;
      const int q = fam_q(fm), qo = q - 4;
      const int rb = S.half - c;
      fw[g][j] = local ? tf[j] : LBMDEM_SLOT_EMPTY;
      bw[g][j] = local ? tb[j] : LBMDEM_SLOT_EMPTY;
      if (local && lane_has_line) {
        tg[(q - 1) * spd + rel] = LBMDEM_SLOT_EMPTY;
        if (rb >= 0 && rb < spd) tg[(qo - 1) * spd + rb] = LBMDEM_SLOT_EMPTY;
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
    const real xc = gxc[g], yc = gyc[g], r2 = gr2[g];
    own[g] = grain_owned(L, xc);
    const bool was_touched = gtouched[g];
    if (was_touched && lane == 0 && consume) S.touched[i] = 0;  // the rasteriser sets it again while it applies
    int xi, xf, yi, yf;
    {
      const real rbl0 = grb[g];   // grain_box() on the values already here
      xi = (int)(xc - rbl0); if (xi < 1) xi = 1;                 // int max(real->int, 1): main.c:1300
      xf = (int)(xc + rbl0); if (xf > L.lx - 2) xf = L.lx - 2;   // main.c:1301
      yi = (int)(yc - rbl0); if (yi < 1) yi = 1;
      yf = (int)(yc + rbl0); if (yf > L.ly - 2) yf = L.ly - 2;
    }
    const bool todo = (consume ? own[g] : !own[g]) && xi <= xf && yi <= yf;
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
#if defined(LBMDEM_AB) && defined(FT_ABLATE)
    if (FT_ABLATE >= 2) continue;
#endif
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
      int r1 = excl & 1023, r2_ = (excl >> 10) & 1023;
      const int tot = __builtin_amdgcn_readlane(incl, 63);
      for (int u = 0; u < wpl; ++u) {
        const int w = lane * wpl + u;
        if (w < nw64) {
          const unsigned long long v = bm[w];
          pw[w] = r1; pw[nw64 + w] = r2_;
          const int d = __popcll(v & M_DIAG), x = __popcll(v & M_XDIR), y = __popcll(v & M_YDIR);
          r1 += d + x; r2_ += d + y;
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
      const int r1 = pw[w] + bd + bx, r2_ = pw[nw64 + w] + bd + by, r3 = r1 + r2_;   // (2 d + x + y = (d + x) + (d + y))
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
    if (bx == 0 && threadIdx.x == 0) packbuf[0] = (real)ntodo;
    return;
  }
  if (threadIdx.x < 8) sZero[threadIdx.x] = 0.0;
  __syncthreads();
#if defined(LBMDEM_AB) && defined(FT_ABLATE)   /* timing experiment (wrong results): 1 = no replay, 2 = no ranking / addends either */
  if (FT_ABLATE >= 1) return;
#endif
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
    // The additions are one dependent chain per lane; what can be hidden is the LDS latency of the addends. Two-stage
    // software pipeline: the block of 8 addends after the one being added is already in flight. No exec masks in the loop:
    // a lane whose list has ended reads a block of zeros instead (h + 0.0 == h: an accumulator is never -0.0, see above).
    // (Before: load 8, wait, add 8 under `if (t < mine)` -- every block paid the full LDS round trip plus a mask round trip,
    // ~2.5 us for the ~300 addends of a torque; the kernel ran 84 us against 61 us with the cross-lane reduction.)
    longest = __builtin_amdgcn_readfirstlane(longest);
    // (s_setprio(3) for the replaying wavefront: 73.5 against 73.8 us -- the replay costs the workgroup's LIFETIME, during which
    // its 26 KB of lists keep the next workgroup out, not issue slots: LABBOOK round 5, note 4)
    auto block_at = [&](int t) { return reinterpret_cast<const real2*>(t < mine ? tl + t : sZero); };
    auto fetch = [&](real2 (&v)[4], int t) {
      const real2* p = block_at(t);
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = p[u];
    };
    auto chain = [&](const real2 (&v)[4]) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { h = h + v[u].x; h = h + v[u].y; }
    };
    real2 va[4], vb[4];
    // (the empty asm statements keep the compiler from merging the two fetches of `va` into one load at the loop head --
    // load(phi) for phi(load, load) -- which puts the LDS round trip back in front of the chain)
#define FT_STAGE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
    fetch(va, 0);
    FT_STAGE();
    for (int t = 0; t < longest; t += 16) {  // wave-uniform trip count
      fetch(vb, t + 8);
      FT_STAGE();
      chain(va);
      FT_STAGE();
      fetch(va, t + 16);
      FT_STAGE();
      chain(vb);
      FT_STAGE();
    }
#undef FT_STAGE
    if (g < NG && gi >= 0 && replayed) fhf[a * L.n + gi] = mine_own ? h * (a == 2 ? scale3 : scale12) : 0.0;
  }
}

// The queued grains, gathered from obst and f: one wavefront per grain, a fixed grid strides over the queue.
__global__ __launch_bounds__(64) void k_forces_gather_queue(const real* __restrict__ f, const int* __restrict__ obst,
                                                            LatticeView L, GrainFluidView G, ForceSlots S,
                                                            double scale12, double scale3, real* __restrict__ fhf,
                                                            unsigned* __restrict__ clear, int nclear) {
  LBMDEM_GATE(L.gate);
  __shared__ ForceLds sh;
  const int lane = threadIdx.x;
  for (int k = blockIdx.x * 64 + lane; k < nclear; k += gridDim.x * 64) clear[k] = 0u;   // (ObstFillJob::clear)
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
  LBMDEM_GATE(L.gate);
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

}  // namespace

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

void launch_slots_clear(const ForceSlots& S, int n, hipStream_t st) {
  const long count = (long)n * 8 * S.spd;
  hipLaunchKernelGGL(k_fill_u64, dim3(grid_for(count)), dim3(256), 0, st, reinterpret_cast<unsigned long long*>(S.tab),
                     count, (unsigned long long)LBMDEM_SLOT_EMPTY);
}

void launch_forces_parity(const real* f, const int* obst, const LatticeView& L,
                          const GrainFluidView& G, double scale12, double scale3, real* fhf,
                          unsigned char* owner, hipStream_t st) {
  hipLaunchKernelGGL(k_forces_parity, dim3(L.n), dim3(64), 0, st, f, obst, L, G, scale12, scale3, fhf, owner);
}

template <int GW, int PASSES>
static void launch_forces_table_t(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                                  const ForceSlots& S, double scale12, double scale3, real* fhf, unsigned char* owner,
                                  int fast, const ObstFillJob& fill, hipStream_t st) {
  const int HB = S.hb, B = 2 * HB + 1;
  const int nw64 = (B * B * 8 + 63) / 64;
  // addends per accumulator: one link per direction and lattice line meeting the disc; fhf3 takes two addends per diagonal
  // link. A disc of reduced radius R <= hb - 1 meets at most 2 ceil(R sqrt 2) + 1 = 2 half - 3 lines of a diagonal family and
  // 2 hb - 1 of an axis family (the table itself has spd >= 2 half + 1 and 2 hb + 1 rows: 25 % more, which at 26 KB of lists
  // per workgroup was a seventh workgroup per CU); a grain that exceeds the lists anyway is gathered (n1 = -1 below).
  const int nd = 2 * S.half - 3, na = 2 * HB - 1;
  const int cap1 = (4 * nd + 2 * na + 7) & ~7, cap3 = (8 * nd + 4 * na + 7) & ~7;
  const size_t lists_doubles = (size_t)FT_WAVES * GW * (2 * cap1 + cap3);
  const size_t lds = lists_doubles * 8 + (size_t)FT_WAVES * nw64 * 24 + (size_t)FT_WAVES * GW * 20;
  const int per_block = FT_WAVES * GW;
  const int ntodo = S.local_list ? S.local_cap : L.n;   // strips: the compacted list of local grains bounds the launch
  int nfill = 0;
  if (fill.map && fill.row1 > fill.row0) {   // ~64 16-byte stores per thread
    const long v4 = (long)(fill.row1 - fill.row0) * fill.L.sy / 4;
    nfill = (int)((v4 + 64L * FT_WAVES * 64 - 1) / (64L * FT_WAVES * 64));
    if (nfill > 512) nfill = 512;
  }
  hipLaunchKernelGGL((k_forces_table<GW, PASSES>), dim3(nfill + (ntodo + per_block - 1) / per_block), dim3(64 * FT_WAVES), lds, st,
                     f, obst, L, G, S, cap1, cap3, nw64, scale12, scale3, fhf, owner, fast ? (int)FT_FAST : (int)FT_CONSUME, S.local_list,
                     S.local_count, PackSides{}, S.mask, S.local_cap, fill, nfill);
  const int grid = L.n < 256 ? L.n : 256;
  hipLaunchKernelGGL(k_forces_gather_queue, dim3(grid), dim3(64), 0, st, f, obst, L, G, S, scale12, scale3, fhf, fill.clear,
                     fill.clear ? fill.nclear : 0);
}

void launch_forces_table_pack(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                              const ForceSlots& S, const int* const list[2], const int* const list_count[2], int cap,
                              real* const buf[2], hipStream_t st) {
  const int HB = S.hb, B = 2 * HB + 1;
  const int nw64 = (B * B * 8 + 63) / 64;
  const int nd = 2 * S.half - 3, na = 2 * HB - 1;   // as in launch_forces_table_t
  const int cap1 = (4 * nd + 2 * na + 7) & ~7, cap3 = (8 * nd + 4 * na + 7) & ~7;
  const size_t lds = (size_t)FT_WAVES * (2 * cap1 + cap3) * 8 + (size_t)FT_WAVES * nw64 * 24 + (size_t)FT_WAVES * 20;
  const int blocks = (cap + FT_WAVES - 1) / FT_WAVES;   // the list lengths are only known on the device
  const PackSides P{{list[0], list[1]}, {list_count[0], list_count[1]}, {buf[0], buf[1]}};
  if (S.spd <= 32)
    hipLaunchKernelGGL((k_forces_table<1, 2>), dim3(blocks, 2), dim3(64 * FT_WAVES), lds, st, f, obst, L, G, S, cap1, cap3,
                       nw64, 0.0, 0.0, (real*)nullptr, (unsigned char*)nullptr, (int)FT_PACK, (const int*)nullptr,
                       (const int*)nullptr, P, S.mask, cap, ObstFillJob{}, 0);
  else
    hipLaunchKernelGGL((k_forces_table<1, 4>), dim3(blocks, 2), dim3(64 * FT_WAVES), lds, st, f, obst, L, G, S, cap1, cap3,
                       nw64, 0.0, 0.0, (real*)nullptr, (unsigned char*)nullptr, (int)FT_PACK, (const int*)nullptr,
                       (const int*)nullptr, P, S.mask, cap, ObstFillJob{}, 0);
}

void launch_forces_slots(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                         const ForceSlots& S, double scale12, double scale3, real* fhf, unsigned char* owner,
                         int fast, const ObstFillJob& fill, hipStream_t st) {
  // (several grains per wavefront sharing the serial replay were measured in round 2 -- 138-351 us against 92: the lists'
  // LDS limits the occupancy -- and are no longer built)
  if (S.spd <= 32) launch_forces_table_t<1, 2>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, fill, st);
  else launch_forces_table_t<1, 4>(f, obst, L, G, S, scale12, scale3, fhf, owner, fast, fill, st);
}

void launch_forces_fast(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                        double scale12, double scale3, real* fhf, unsigned char* owner,
                        hipStream_t st) {
  const long threads = (long)L.n * 64;
  hipLaunchKernelGGL(k_forces_fast, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, st, f, obst, L,
                     G, scale12, scale3, fhf, owner);
}

