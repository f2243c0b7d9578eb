// lbmdem_internal.h -- shared between the HIP translation units of liblbmdem_hip.so.
// Not part of the public ABI (that is include/lbmdem_hip.h).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

// ---------------------------------------------------------------------------------------------
// `real` -- the reference's arithmetic type (main.c:34-40): double, or float when the reference is compiled
// -DSINGLE_PRECISION. The library is built once per type (liblbmdem_hip.so / liblbmdem_hip_sp.so, csrc/Makefile SP=1);
// every device formula is written with the reference's own literals and declarations, so that the usual arithmetic
// conversions promote exactly where the reference's C does (an `1.`, `0.5` or `4.5` literal, sqrt(), fabs() make a
// sub-expression double even in the float build; an int literal does not). Host buffers at the C ABI are double in
// both builds (every float is a double).
// ---------------------------------------------------------------------------------------------
#ifdef LBMDEM_SINGLE_PRECISION
typedef float real;
typedef float2 real2;
#define LBMDEM_REAL_MANT 24          /* significand bits incl. the hidden one */
#define LBMDEM_TILE_Y 32             /* nodes per 128-byte line */
#else
typedef double real;
typedef double2 real2;
#define LBMDEM_REAL_MANT 53
#define LBMDEM_TILE_Y 16
#endif
__host__ __device__ __forceinline__ real2 make_real2(real a, real b) { real2 v; v.x = a; v.y = b; return v; }

// ---------------------------------------------------------------------------------------------
// Device-side views (passed by value to kernels)
// ---------------------------------------------------------------------------------------------

// Local lattice slab: global rows [gx0, gx0 + nxl), all ly columns. Device layout of the populations:
// f[xl][y / 16][q][y % 16] (row pitch sy, a multiple of 16; see fbase() in lbm_device.h) -- the reference's
// fast axis (main.c:56) is kept as the fast axis so that an x-strip is one contiguous slab and halo rows
// are contiguous. The obstacle map is plain obst[xl][y] with the same pitch.
struct LatticeView {
  int lx, ly;    // global lattice size
  int gx0;       // global x of local row 0
  int nxl;       // local rows (owned + halo)
  int xo0, xo1;  // owned local rows [xo0, xo1)
  int sy;        // row pitch in elements
  long plane;    // nxl * sy: nodes of the slab (a lattice is 9 * plane reals)
  int n;         // nbgrains; also the obst code of the lattice-edge walls (main.c:676,681)
  real dx, c, Mgx, Mby;
  real s2, s3, s5, s7, s8, s9;
  int reduced_lt1;  // phys.reductionR < 1: reduced discs lie strictly inside the grains (always, in the reference)
  // x / c and x / (c*c) as correctly rounded quotients from the reciprocals (round-1 form of the quotients, no longer used);
  // recip_ok = 0 when a divisor's significand is all ones (the one case the construction does not cover)
  real rc, rcc;
  int recip_ok;
  // The stop word of the handle (device memory, 0 = go): a kernel that finds it set returns at once. Raised by a launch of the
  // multi-sub-step DEM kernel that could not finish (k_dem_chain), it turns everything queued behind that launch into
  // nothing, so that the host finds the state exactly as it was before the launch and replays from there (lbmdem_capi.hip,
  // chain_recover). null: no such word.
  const int* gate;
  real lid6;               // EXTENSION: uw_h / 6 of the lid terms commented out at main.c:1129-1130; 0 = off
  real cc;                 // c * c
  real wc_diag, wc_axis;   // w_q / c for the diagonal and the axis directions (main.c:1174,1184)
};

// obst = -1 in the interior, nbgrains on the four lattice edges (main.c:669-683, 997-999); 16-byte stores (the row pitch
// is a multiple of 16 elements). Thread `first` of `stride` threads.
__device__ __forceinline__ void obst_fill_range(int* __restrict__ obst, const LatticeView& L, long first, long stride,
                                                int row0 = 0, int row1 = -1) {   // local rows [row0, row1); -1: to the end
  const long total4 = (long)(row1 < 0 ? L.nxl : row1) * L.sy / 4;
  int4* o4 = reinterpret_cast<int4*>(obst);
  for (long k = (long)row0 * L.sy / 4 + first; k < total4; k += stride) {
    const long e = k * 4;
    const int xl = (int)(e / L.sy), y0 = (int)(e % L.sy);
    const int gx = L.gx0 + xl;
    const bool xedge = (gx == 0 || gx == L.lx - 1);
    int4 v;
    v.x = (xedge || y0 == 0 || y0 >= L.ly - 1) ? L.n : -1;
    v.y = (xedge || y0 + 1 >= L.ly - 1) ? L.n : -1;
    v.z = (xedge || y0 + 2 >= L.ly - 1) ? L.n : -1;
    v.w = (xedge || y0 + 3 >= L.ly - 1) ? L.n : -1;
    o4[k] = v;
  }
}

// Grain state used by the fluid kernels. xc, yc, r2, rbl0 are the lattice-unit centre, squared
// reduced radius and unreduced radius (main.c:1009-1013), refreshed by every obst_construction.
struct GrainFluidView {
  const real* x1; const real* x2;
  const real* v1; const real* v2; const real* v3;
  const real* xc; const real* yc; const real* r2; const real* rbl0;
  // the same eight values packed per grain, [n][8] = {x1, x2, v1, v2, v3, xc, yc, r2}: one 64-byte
  // record, fetched with four 16-byte loads by the fluid kernels
  const real* pk;
  // Lowest grain index covering a node, for nodes covered by MORE than one disc (written by the rasteriser when its
  // atomicMax finds a previous owner): (epoch & 0xFFF) << 20 | (0xFFFFF - index), valid when the epoch matches the
  // last rasterisation. Decides `act` where discs overlap (main.c:1039-1052 sees the map as it was when the OWNER was
  // painted: a neighbour counts as fluid iff no grain of index <= owner covers it). null: fewer than 2^20 grains only.
  const unsigned* mincov;
  unsigned epoch;
};
#define LBMDEM_MINCOV_IDS (1 << 20)

// Per-grain table of link momentum-exchange sums, written by the fused fluid kernel where it evaluates the
// interpolated bounce-back links and consumed by the hydrodynamic-force kernel (main.c:1313-1316:
// f[P][opp q] + f[N][q] of a link from solid node P of grain i to fluid node N = P + e_q).
// tab[i][q - 1][rel]: the nodes of a reduced disc on one lattice line form one interval, so grain i has at most
// ONE link of direction q into a fluid node per lattice line parallel to e_q; rel = c(P) - c(centre) + half with
// c(x, y) = ey * x - ex * y numbers those lines (centre = the truncated lattice coordinates of the grain centre).
// A slot holds LBMDEM_SLOT_EMPTY unless the fused kernel of this step wrote it. `touched[i]` = some node of grain
// i's disc is also covered by another grain's disc (set by the rasteriser). tab == nullptr: feature off.
struct ForceSlots {
  real* tab;
  unsigned char* touched;
  int spd;   // slots per direction (2 * half + 1 rounded up)
  int half;
  int hb;    // >= largest reduced radius in nodes + 1: a grain's nodes lie within +-hb of its truncated centre
  int* gathered;  // device counter: grains the table could not serve ... (one of two: the queue kernel zeroes the
                  // OTHER one for the next fluid step, so no memset sits between the kernels)
  int* gathered_next;
  int* queue;     // ... and their indices, for the gather kernel that follows
  int* error;     // device flag: a grain cut by a strip boundary that neither the table nor a local gather can serve
  const unsigned char* mask;  // strip decomposition with distributed grains: the grains the rasteriser handled; else null
  const int* local_list;      // ... the same set as a list (+ its device-side length and its capacity), else null
  const int* local_count;
  int local_cap;
};
#define LBMDEM_SLOT_EMPTY 0x7FF8C0DE5107E117ull  /* a quiet NaN no arithmetic produces (double build; the float build has no table) */
constexpr int LBMDEM_SPD_MAX = 64;               // larger grains (reduced radius > ~20 nodes): feature off

#define LBMDEM_GATE(g) do { if ((g) != nullptr && *(g) != 0) return; } while (0)   /* first statement of a gated kernel */

struct DemParams {
  int n;
  const int* gate;                       // the handle's stop word (LatticeView::gate)
  real dt, dt2;
  real kg, nug, kt, mu, murf;            // grain-grain (main.c:104-113)
  real km, num, ktm, mumb, mum, nugt;    // walls
  real Mgx, Mdx, Mby, Mhy;               // DEM wall positions (main.c:201-204,1555-1561)
  double wallT_vel;                      // amp*freq*cos(freq*t) (main.c:855): cos() makes it a double in either build
  real xG, yG;
  real distVerlet;
};

// kinematic state, SoA; two copies ping-pong across DEM sub-steps
struct Kin {
  real *x1, *x2, *x3, *v1, *v2, *v3, *a1, *a2, *a3;
};

// ---------------------------------------------------------------------------------------------
// Launchers (host functions defined in the .hip files)
// ---------------------------------------------------------------------------------------------

// lbm_fused.hip, lbm_forces.hip, lbm_obst.hip, lbm_lattice.hip
void launch_obst_fill(int* obst, const LatticeView& L, hipStream_t st);
// fills xc, yc, r2, rbl0, pk (per-grain lattice geometry, main.c:1009-1013) and paints the reduced discs
// The lattice-unit centres a map buffer's discs were painted at (mode: 0 = the grain left nothing in that buffer):
// what the in-place update of the map (k_obst_update) compares the new centres with. Null pointers: not recorded.
// `still2`: while the centre stays within sqrt(still2) of (xc, yc) no node of the disc changes sides (0: not known).
struct ObstSnap { real* xc; real* yc; real* still2; unsigned char* mode; };
// Where the two obstacle maps of a fluid step differ (reinit_obst_density needs the previous owner only there): one bit per
// lattice row and 64-column window of the fused kernel -- bits[window * words + (row >> 5)] bit (row & 31), window w = columns
// [w * ww - off, w * ww - off + 63] -- set by the rasterisation at the end of a run of sub-steps (ChainPaint). A window row
// whose bit is clear is read from ONE map. bits == nullptr: not known, both maps are read everywhere.
struct ObstChange { unsigned* bits; int words; int ww; int off; };
void collide_stream_windows(int* ww, int* off);   // the fused kernel's window geometry (lbm_fused.hip)
// test aid: counts the (row, window) pairs whose bit is clear although the two maps differ there (lbm_obst.hip)
void launch_count_differences(const real* a, const real* b, long n, int* bad, hipStream_t st);
void launch_change_verify(const int* ob_old, const int* ob_new, const LatticeView& L, const ObstChange& chg, int windows, int* bad, hipStream_t st);
void launch_obst_paint(int* obst, const LatticeView& L, int n, const real* x1, const real* x2, const real* r,
                       const real* rLB, const real* v1, const real* v2, const real* v3, real* xc,
                       real* yc, real* r2, real* rbl0, real* pk, unsigned char* touched,
                       const unsigned char* mask, unsigned* mincov, unsigned epoch, const int* list,
                       const int* list_count, int list_cap, const int* verlet_offsets, const int* verlet_nbr,
                       const ObstSnap& snap_out,
                       hipStream_t st);   // verlet_*: the symmetric pair list (null: every grain takes the atomic path)
// a reduced disc on the lattice: centre, squared radii, clamped bounding box (main.c:1009-1027); shared by k_obst_update and
// the rasterisation at the end of k_dem_chain
struct DiscGeo {
  real xc, yc, r2, R2;
  int xi, xf, yi, yf;
  bool any;
};
__device__ __forceinline__ DiscGeo disc_geo(const LatticeView& L, real xc, real yc, real rlb, real rbl0, bool valid) {
  DiscGeo g;
  g.xc = xc; g.yc = yc; g.r2 = rlb * rlb; g.R2 = rbl0 * rbl0;
  g.xi = (int)(xc - rbl0); g.xf = (int)(xc + rbl0);   // main.c:1016-1023, as k_obst_paint
  if (g.xi < 1) g.xi = 1;
  if (g.xf >= L.lx - 1) g.xf = L.lx - 2;
  g.yi = (int)(yc - rbl0); g.yf = (int)(yc + rbl0);
  if (g.yi < 1) g.yi = 1;
  if (g.yf >= L.ly - 1) g.yf = L.ly - 2;
  if (g.xi < L.gx0) g.xi = L.gx0;
  if (g.xf > L.gx0 + L.nxl - 1) g.xf = L.gx0 + L.nxl - 1;
  g.any = valid && g.xi <= g.xf && g.yi <= g.yf;
  return g;
}
__device__ __forceinline__ bool disc_has(const DiscGeo& g, int x, int y) {
  if (!g.any || x < g.xi || x > g.xf || y < g.yi || y > g.yf) return false;
  const real d2 = (x - g.xc) * (x - g.xc) + (y - g.yc) * (y - g.yc);
  return d2 <= g.R2 && d2 <= g.r2;
}

// the same outputs, the map written only where a disc's footprint differs from the one `was` describes (needs the list)
void launch_obst_update(int* obst, const LatticeView& L, int n, const real* x1, const real* x2, const real* r,
                        const real* rLB, const real* v1, const real* v2, const real* v3, real* xc, real* yc, real* r2,
                        real* rbl0, real* pk, unsigned char* touched, unsigned* mincov, unsigned epoch, const int* voff,
                        const int* vnbr, const ObstSnap& was, const ObstSnap& now, const real* xreb, const real* yreb,
                        real moved_limit, int* moved_flag, int list_generation, hipStream_t st);
void launch_collide_stream(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                           const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, hipStream_t st,
                           const ObstChange& chg = ObstChange{nullptr, 0, 0, 0});
// the two edge-row ranges [lo0, lo1) and [hi0, hi1) of a strip (either may be empty): one launch when they are equally
// wide (two segments), else one launch_collide_stream each
void launch_collide_stream_edges(const real* fin, real* fout, const int* obst_old, const int* obst_new,
                                 const LatticeView& L, const GrainFluidView& G, const ForceSlots& S, int lo0, int lo1,
                                 int hi0, int hi1, hipStream_t st);
// true when launch_collide_stream(..., S) with S.tab != nullptr fills the table (the marching kernel does)
bool collide_stream_fills_slots(const LatticeView& L);
void collide_stream_work_order(const LatticeView& L, int* info12);   // the launch's work items (lbmdem_fused_work_order)
void launch_slots_clear(const ForceSlots& S, int n, hipStream_t st);
// parity forces from the slot table (resets it to empty); grains whose table is incomplete gather from f
struct ObstFillJob;   // (below) a range of rows of an obstacle map to reset beside the kernel's own work; map == nullptr: none
void launch_forces_slots(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                         const ForceSlots& S, double scale12, double scale3, real* fhf, unsigned char* owner,
                         int fast, const ObstFillJob& fill, hipStream_t st);
// strip decomposition: this rank's part of the tables of the listed grains (owned by a neighbour rank), completed
// and written as {count; count x {id, 8 * spd slots}} to a message buffer
void launch_forces_table_pack(const real* f, const int* obst, const LatticeView& L, const GrainFluidView& G,
                              const ForceSlots& S, const int* const list[2], const int* const list_count[2], int cap,
                              real* const buf[2], hipStream_t st);

// dist_kernels.hip -- strip decomposition with the grains distributed over the ranks (strips.py)
struct DistDevice {
  unsigned char* active;     // [n] grains this rank integrates: owned + margin (refreshed from the neighbours)
  unsigned char* fluidmask;  // [n] grains with exact state that may overlap this rank's rows (paint, force table)
  int* send_list[2];         // [cap_g] owned grains within `margin` rows of the low / high cut
  int* strad_list[2];        // [cap_t] grains owned by the low / high neighbour whose link ring reaches this rank's rows
  int* recv_ids[2];          // [cap_g] ids of the last kinematics message from the low / high neighbour
  int* local_list;           // [cap_l] the grains of fluidmask, compacted (launch bound of the per-grain kernels)
  int* counters;             // [8] send x2, straddler x2, received x2, local, ticket -- this period's set
  int* counters_alt;         // the other set: zeroed during this period, used by the next (the host swaps the two)
  int cap_g, cap_t, cap_l;
};
struct DistGeom { real lo, hi, margin, dx, Mgx; int has_lo, has_hi, first, last, gx0, nxl; };
int dist_alloc(DistDevice& D, int n, int cap_g, int cap_t, int cap_l);
void dist_free(DistDevice& D);
// ownership, masks and the send / straddler lists from the current positions of the grains that were active
// (error_mirror: device address of the pinned host word that follows *error, written by the launch; may be null)
void launch_dist_classify(const DistDevice& D, const DistGeom& Gm, int n, const real* x1, const real* r,
                          const real* rLB, unsigned char* owner, int* error, int* error_mirror, hipStream_t st);
// the same + the two kinematics messages packed by the launch's last workgroup (C transport: one launch instead of two)
void launch_dist_classify_pack_kin(const DistDevice& D, const DistGeom& Gm, int n, const real* x1, const real* r,
                                   const real* rLB, unsigned char* owner, int* error, int* error_mirror, const Kin& K,
                                   real* kin_lo, real* kin_hi, hipStream_t st);
// C transport, one launch: merge the neighbours' TABLES messages, unpack their KIN messages, reset `dead_obst` (null: not)
void launch_dist_unpack_tables_kin_fill(const ForceSlots& S, const real* tab_lo, const real* tab_hi, int cap_t,
                                        const DistDevice& D, const real* kin_lo, const real* kin_hi, const Kin& K, int n,
                                        int* dead_obst, const LatticeView& L, hipStream_t st);
// messages: {count; count x {id, x1 x2 x3 v1 v2 v3 a1 a2 a3}} / {count x {fhf1 fhf2 fhf3}} in the order of that list
// (both sides in one launch: `lo` / `hi` = the low / high neighbour's buffer, null to skip)
void launch_dist_pack_kin(const DistDevice& D, const Kin& K, real* lo, real* hi, hipStream_t st);
void launch_dist_unpack_kin(const DistDevice& D, const Kin& K, const real* lo, const real* hi, int n, int* error,
                            hipStream_t st);
void launch_dist_pack_fhf(const DistDevice& D, const real* fhf, int n, real* lo, real* hi, hipStream_t st);
void launch_dist_unpack_fhf(const DistDevice& D, real* fhf, int n, const real* lo, const real* hi, hipStream_t st);
void launch_dist_merge_tables(const ForceSlots& S, const real* lo, const real* hi, int cap, hipStream_t st);
void launch_dist_poison(const DistDevice& D, const Kin& a, const Kin& b, int n, hipStream_t st);
void launch_forces_parity(const real* f, const int* obst, const LatticeView& L,
                          const GrainFluidView& G, double scale12, double scale3, real* fhf,
                          unsigned char* owner, hipStream_t st);
void launch_forces_fast(const real* f, const int* obst, const LatticeView& L,
                        const GrainFluidView& G, double scale12, double scale3, real* fhf,
                        unsigned char* owner, hipStream_t st);
void launch_aos_to_soa(const real* aos_rows, real* f, const LatticeView& L, hipStream_t st);
void launch_soa_to_aos(const real* f, real* aos_rows, const LatticeView& L, int xl0, int nrows,
                       hipStream_t st);
void launch_fill_equilibrium(real* f, const LatticeView& L, hipStream_t st);
void launch_plain_copy(const void* src, void* dst, size_t bytes, hipStream_t st, int shape = 0);
int plain_copy_shapes();
void launch_macro(const real* f, const LatticeView& L, int xl0, int nrows, real* rho, real* ux,
                  real* uy, hipStream_t st);
void launch_density_partial(const real* f, const LatticeView& L, double* partial, int nblocks,
                            hipStream_t st);
// the reference's serial total density (see k_density_rowquanta): per owned row an approximate sum, then the integer
// number of quanta 2^(kexp[row] - 52) the row adds + a flag when the shortcut does not apply to the row
void launch_density_rowsum(const real* f, const LatticeView& L, double* rowsum, hipStream_t st);
void launch_density_rowquanta(const real* f, const LatticeView& L, const int* kexp, unsigned long long* quanta,
                              int* flags, hipStream_t st);
// both sides in one launch; a null buffer skips the side
void launch_halo_pack(const real* f, const LatticeView& L, int xl0_lo, int xl0_hi, int nrows, real* buf_lo,
                      real* buf_hi, hipStream_t st);
void launch_halo_unpack(real* f, const LatticeView& L, int xl0_lo, int xl0_hi, int nrows, const real* buf_lo,
                        const real* buf_hi, hipStream_t st);
// the five float32 fields of write_vtk (main.c:284-323), [ly][lx] order, owned rows only
void launch_vtk_fields(const real* f, const int* obst, const LatticeView& L, const real* gp,
                       const real* v1, const real* v2, const real* a1, const real* a2,
                       real rho_moy, float* grain_pressure, float* grain_velocity,
                       float* grain_acceleration, float* fluid_pressure, float* fluid_velocity,
                       hipStream_t st);

// dem_kernels.hip
struct VerletDevice {
  // uniform grid
  int ncx, ncy;
  real ox, oy, cs;
  unsigned int* keys_in;          // [n] cell of grain i
  int* vals_in; int* vals_out;    // [n] arrival rank within the cell; [n] the grains ordered by cell
  int* cell_start; int* cell_end; // [ncell + 1] first grain of cell c in vals_out (.. [ncell] = n); per-cell counts (0 between rebuilds)
  void* scan_tmp; size_t scan_tmp_bytes;
  // symmetric CSR neighbour list, partners ascending
  int* counts;   // n
  int* offsets;  // n + 1
  int* nbr;      // cap
  int* own;      // cap: the grain each entry belongs to
  long cap;
  unsigned char* wallflags;  // n
  int* overflow;             // device flag
  // what the multi-sub-step kernel (k_dem_chain) needs of the list, per tile of DEM_TILE consecutive grains; refreshed
  // by every rebuild: the distinct partners outside the tile ("halo" grains, staged in LDS every sub-step) and, per list
  // entry, where its two grains sit in that staging
  int* halo_ids;             // [tiles][DEM_CHAIN_HALO]
  int* halo_cnt;             // [tiles]
  unsigned* emeta;           // [cap] own grain's index in its tile | (own < partner) << 6 | partner's staging slot << 8
  unsigned char* tile_far;   // [tiles] the tile has a partner in a tile that is expected on another XCD
  // The tiles of k_dem_chain are COMPOSED: tile t holds the grains tile_grains[t][0 .. DEM_TILE) (ascending; -1 = none),
  // where[g] = t << 6 | position. By index (tile t = grains [64 t, 64 t + 64)) on distributed handles; by a space-filling
  // curve over the initial positions otherwise -- a tile is a compact patch of the packing whatever the grains' numbering, so
  // its partners outside itself are the patch's rim (~45 grains instead of 83 on the row-numbered bench packing, of 164 on
  // the reference's own bin/50000.data), and the tiles of one XCD eighth are a compact region. Nothing a grain's sums
  // depend on: entries stay in list order per grain, contacts in the (lower, higher) index frame.
  int* tile_grains;          // [tiles][DEM_TILE]
  int* where;                // [n]
  real *xreb, *yreb;         // [n] the positions the list was built from
  const int* gate;           // the handle's stop word (LatticeView::gate): a rebuild behind a failed launch leaves the list alone
};
constexpr int DEM_CHAIN_HALO = 256;          // halo grains staged per tile; partners beyond that are read from memory per entry
constexpr unsigned DEM_CHAIN_DIRECT = 0xFFFFu;   // emeta slot value of such a partner
int verlet_alloc(VerletDevice& V, int n, real cs, real ox, real oy, real wx, real wy);
void verlet_free(VerletDevice& V);
// returns 0 or the hipError_t of the failing call (hipCUB scan, launch)
int launch_verlet_rebuild(VerletDevice& V, const Kin& K, const real* r, const DemParams& P,
                          hipStream_t st);
// Buffers of the order-dependent contact diagnostics fr, ice, slip, rw (main.c:782-789, 840-843, 851, 893,
// 916-919, 942, 1462-1466, 1490-1494): they depend on "previous contact" carries (pft, pff, pf, ic,
// main.c:130-131) that thread through the reference's serial contact loop. Only used by the rare sub-steps
// that produce write_DEM's table (see launch_diag_extra).
struct DiagExtra {
  real *fr, *ice, *slip, *rw;          // [n] results
  real* a1gc;                          // [n] a1 after the grain contacts; the wall pass keeps adding (= g[].a1 then)
  real *e_ft, *e_f3, *e_avt, *e_av3;   // [cap] per list entry with own < partner: ft, f3, |vt dt|, |v3_own dt|
  real *e_dslip, *e_drw;               // [cap] its contribution to slip / rw once the carries are known
  unsigned char* e_touched;              // [cap] 1 = entry with own < partner and dn < 0
  int* wlist;                            // [4][n] wall candidate lists (bottom, top, left, right), grains ascending
  int* wcount;                           // [4]
  real* carry;                         // pft, pff, pf: persist from sub-step to sub-step
  const int* gate;                     // the handle's stop word (LatticeView::gate)
};
int diag_extra_alloc(DiagExtra& X, int n, long cap, real* carry);

// Where the carries come from when the sub-step before was not a diagnostic one: every ordinary sub-step leaves, per
// tile of DEM_TILE consecutive grains and per kind of contact that assigns a carry (main.c:784-786/1410-1411 grain
// contacts, 842-843 bottom wall, 919 left wall, 942 right wall), the values of the LAST such contact of the tile in
// the reference's order, stamped with the sub-step's sequence number. Before a diagnostic sub-step one small kernel
// picks, per carry, the youngest record (sub-step, then kind in program order, then tile) -- the contact the
// reference evaluated last, however long ago.
constexpr int DEM_TILE = 64;
enum : int { CARRY_GRAIN = 0, CARRY_BOTTOM = 1, CARRY_LEFT = 2, CARRY_RIGHT = 3 };
struct CarryTrack {
  long long* stamp;  // [tiles][4] sequence number of the sub-step that wrote the record, -1: none
  real* val;       // [tiles][4][2] ft, f3
  real* carry;     // [3] pft, pff, pf
  int tiles;
  long long* who;       // [tiles][4] the contact of the record: (grain << 32) | partner (walls: partner 0)
  const int* where;     // VerletDevice::where (which tile a grain's record lies in does not depend on it; kept for tools)
  long long* best_key;  // [3][2] left by launch_carry_resolve: {(stamp + 1) * 4 + kind, who} of the record each carry came from
  const int* gate;      // the handle's stop word (LatticeView::gate)
};
int carry_track_alloc(CarryTrack& T, int n);
void carry_track_free(CarryTrack& T);
// carry[] <- the youngest records with stamp >= min_stamp (carry[] itself is younger than anything below that)
void launch_carry_resolve(const CarryTrack& T, long long min_stamp, hipStream_t st);
void diag_extra_free(DiagExtra& X);
// after the DIAG sub-step kernel: carries scanned over the contacts in the reference's order, per-grain sums,
// then the four wall loops replayed serially
void launch_diag_extra(const DiagExtra& X, const Kin& in, const real* r, const VerletDevice& V,
                       const DemParams& P, int film, hipStream_t st);
void launch_fill_own(const VerletDevice& V, int n, hipStream_t st);
void launch_tile_halo(const VerletDevice& V, int n, hipStream_t st);   // halo_ids / halo_cnt / emeta from offsets, nbr, own
// the composition of the tiles: tile_grains_host[tiles * DEM_TILE] (ascending within a tile, -1 padded) -> V.tile_grains, V.where
int verlet_set_tiles(VerletDevice& V, int n, const int* tile_grains_host);
// A slice of the obstacle map that the next rasterisation starts from (obst = -1 / wall codes, main.c:997-999), reset by
// extra workgroups of a DEM sub-step launch: the sub-step kernel is a latency chain that leaves the GPU idle, the reset is
// 67 MB of stores per fluid step -- in npDEM slices they disappear under the sub-steps. map == nullptr: nothing to do.
// (`clear`: words to zero beside the force kernels -- the change bits of the map that is painted next, lbmdem_forces_fluid)
struct ObstFillJob { int* map; LatticeView L; int row0, row1; unsigned* clear; int nclear; };
void launch_dem_substep(const Kin& in, const Kin& out, const real* r, const real* m,
                        const real* It, const real* fhf, const VerletDevice& V, real* pout,
                        const DemParams& P, int film, real* diag, const DiagExtra* X, const unsigned char* active,
                        const CarryTrack* track, long long stamp, const unsigned char* owner, const ObstFillJob& fill,
                        hipStream_t st);
void launch_obst_fill_rows(int* obst, const LatticeView& L, int row0, int row1, hipStream_t st);

// Several consecutive ordinary sub-steps (no film law, no diagnostics, no list rebuild in between) in ONE launch: every
// tile keeps its grains in registers and hands the drifted state of each sub-step to the tiles of its partners through
// `pub` -- per grain and parity of the sub-step one 128-byte line of five tagged 16-byte slots {lo, tag, hi, tag}, written
// and read with sc1 (write-through / L1-bypassing) accesses: the data is its own flag, there is no grid barrier.
struct DemChain {
  void* pub;            // [local copy, remote copy][2 parities][n] lines of 128 bytes
  size_t pub_bytes;
  volatile int* err_host;  // pinned host word: a tile gave up waiting for a partner (every spin is bounded): 1 + the low bits of
                           // the launch's first sequence number ...
  int* err;             // ... and its device address
  int* gate;            // device word raised together with it: the handle's stop word (LatticeView::gate)
  int* census;          // device counter of the residency check
  int capacity;         // tile slots the launch may use (all its workgroups must be resident at once); 0: not usable
  long long* dbg;       // experiment build only: [tiles][8] clocks of the last launch (lbmdem_debug_chain_times)
};
// The rasterisation of the reduced discs (k_obst_paint's work, main.c:1009-1032) done by the tiles of k_dem_chain when the
// run of sub-steps ends where a fluid step begins: the final positions of a tile's grains and of all their partners are
// in its LDS at that moment -- no launch, none of the three dependent rounds of loads the stand-alone kernel starts with.
// The canvas must have been reset before the launch. obst == nullptr: not asked for.
struct ChainPaint {
  int* obst;
  LatticeView L;
  const real* rLB;
  real *xc, *yc, *r2, *rbl0, *pk;
  unsigned char* touched;
  unsigned* mincov;
  unsigned epoch;
  // was.xc == nullptr: the canvas is clean, every disc is painted; else the canvas holds the picture `was` describes and
  // only the nodes whose owner changes are written (k_obst_update's rules). `now` records the new picture.
  ObstSnap was, now;
  const real* r;
  const real *xreb, *yreb;   // where the pair list found the grains (has one outrun the list?)
  real moved_limit;
  int* moved_flag;
  int list_generation;
  // chg.bits != nullptr (in place only): the (row, window) bits of the nodes whose owner differs between this picture and the
  // one in the OTHER map buffer -- the previous fluid step's --, which `other` describes (the fused kernel then reads the
  // previous owner only there). All clear on entry.
  ObstChange chg;
  ObstSnap other;
  int windows;
};
int dem_chain_alloc(DemChain& C, int n);
void dem_chain_free(DemChain& C);
// how many workgroups of k_dem_chain this GPU keeps resident at once (occupancy x CUs), verified by a census launch of
// `tslots` workgroups that all have to see each other; 0 when they do not fit
int dem_chain_census(DemChain& C, int tslots, hipStream_t st);
int dem_chain_tslots(int n);
void launch_dem_chain(const Kin& in, const Kin& out, const real* r, const real* m, const real* It, const real* fhf,
                      const VerletDevice& V, real* pout, const DemParams& P, const unsigned char* active,
                      const CarryTrack* track, long long stamp0, const unsigned char* owner, const ObstFillJob& fill,
                      const DemChain& C, int nsteps, const ChainPaint& paint, hipStream_t st);
