// lbmdem_handle.h -- what the translation units of the C ABI share (lbmdem_capi.hip: handle, step driver, state
// transfer; lbmdem_output.hip: the reference's file writers; lbmdem_checkpoint.hip; lbmdem_strips.hip: strip
// decomposition; lbmdem_comm.hip: the RCCL transport). Not part of the public ABI.
#pragma once

#include "../../include/lbmdem_hip.h"
#include "lbmdem_internal.h"

#include <ctype.h>
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <new>
#include <vector>

#define REF_PI 3.14159265358979 /* main.c:42 */
#define RHO_S 2650              /* main.c:44 */

// sets the text lbmdem_last_error() returns (thread-local) and returns `code`
int lbmdem_fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
#define fail lbmdem_fail

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(LBMDEM_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)
#define RC_TRY(expr) do { int rc_ = (expr); if (rc_ != LBMDEM_OK) return rc_; } while (0)

// What lbmdem_run / lbmdem_run_dem were asked for, call by call, since the stream was last known to be sound ...
struct RunLogEntry { int fluid; long n; };
// ... and the host's side of the handle as of just before a launch of k_dem_chain (everything that moves from launch to
// launch): if that launch gives up, every kernel behind it returns at once (the stop word, LatticeView::gate), the device
// holds exactly this state, and the host goes back to it and replays the log one launch per sub-step (chain_recover,
// lbmdem_capi.hip).
struct ChainSnap {
  long long seq;      // sequence number of the launch's first sub-step (what the failing kernel reports)
  long log_idx, done; // the run-log entry the launch belongs to, and the sub-steps of that call that were done before it
  int fcur, ocur, kcur, obst_reset_rows, snap_cur[2], chg_state[2], list_generation;
  bool obst_pending, snap_ok[2], diag_valid, slots_clean, last_forces_from_table, slots_valid, verlet_ok,
      verlet_tracks_positions, chain_painted;
  long long substep_seq, carry_from;
  int *gathered, *gathered_next;
  long nbsteps;
  double Mdx, Mhy;
};

struct lbmdem_handle {
  lbmdem_config cfg;
  LatticeView L;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // lattice
  real* f[2] = {nullptr, nullptr};
  int fcur = 0;
  int* obst[2] = {nullptr, nullptr};
  int ocur = 0;           // map the current f was produced with ("old" for the next collide_stream)
  bool obst_pending = false;  // obst[1 - ocur] holds a newer map not yet consumed by collide_stream
  // obst[1 - ocur] -- the map the fused kernel has finished with, the next rasterisation's canvas -- is reset to "no
  // grains" in slices by the DEM sub-step launches (single domain) or all at once by the launch that unpacks the
  // neighbours' messages (C transport): local rows [0, obst_reset_rows) are done; obst_construction does the rest
  int obst_reset_rows = 0;
  // the map updated in place (k_obst_update, lbm_obst.hip): per map buffer the centres its discs were painted at, in two
  // slots (the update reads one and writes the other: partners look at each other's old centres meanwhile)
  ObstSnap snap[2][2] = {};
  int snap_cur[2] = {0, 0};
  bool snap_ok[2] = {false, false};   // snap[b][snap_cur[b]] describes what obst[b] holds
  bool obst_update = true;            // lbmdem_set_obst_update: maps are updated in place where a picture to start from exists
  long obst_updates = 0, obst_repaints = 0;   // lbmdem_obst_stats
  int list_generation = 0;            // rebuilds so far
  volatile int* moved_host = nullptr; // pinned: the generation of the list in which k_obst_update found a grain too far from where the list found it
  int* moved_dev = nullptr;
  // where the two maps of a fluid step differ (ObstChange, lbmdem_internal.h): chg[b] belongs to obst[b] as the NEW map.
  // chg_state[b]: 0 = says nothing, 1 = all clear and nobody has painted obst[b] since, 2 = written by the rasterisation
  // that painted obst[b] in place (k_dem_chain's tail) against the picture in obst[1 - b]
  unsigned* chg[2] = {nullptr, nullptr};
  int chg_state[2] = {0, 0};
  int chg_words = 0, chg_windows = 0, chg_ww = 0, chg_off = 0;
  bool chg_on = true;                 // lbmdem_set_change_mask
  int chg_verify = 0;                 // ... 2: every use is checked against the two maps first (tests)
  long chg_used = 0;                  // fused launches that read one map where the bits allowed it
  int* chg_bad = nullptr;             // device counters of the verification: [0] clear bits over rows that differ, [1] populations
                                      // that come out differently with and without the bits
  real* chg_fcheck = nullptr;         // (mode 2) a second output lattice
  // collide_stream in two parts (lbmdem_collide_stream_part): after EDGES the interior rows of f[fcur] are
  // still missing; the operands of the launch are kept for INTERIOR
  bool cs_interior_pending = false;
  bool cs_prepared = false;   // lbmdem_collide_stream_prepare has run for the coming EDGES part
  const real* cs_fin = nullptr;
  const int *cs_ob_old = nullptr, *cs_ob_new = nullptr;
  int cs_lo_end = 0, cs_hi_begin = 0;  // interior = local rows [cs_lo_end, cs_hi_begin)
  ForceSlots cs_slots{};
  // grains
  int n = 0;
  real* gbuf = nullptr;  // one allocation, carved below
  Kin kin[2];
  int kcur = 0;
  real *r = nullptr, *m = nullptr, *It = nullptr, *rLB = nullptr;
  real *xc = nullptr, *yc = nullptr, *r2 = nullptr, *rbl0 = nullptr;
  real* pk = nullptr;   // [n][8] packed fluid-side grain records
  real* gp = nullptr;   // [n] grain pressure g.p of the last DEM sub-step (main.c:187,776)
  real* diag = nullptr; // [8][n] reals s f1 f2 ifm M11 M12 M21 M22, then [2][n] ints z zz
  bool diag_always = false;
  DiagExtra dx{};          // buffers of the order-dependent diagnostics fr, ice, slip, rw (allocated on first use)
  bool dx_ready = false;
  CarryTrack ct{};            // "previous contact" carries: records left by every ordinary sub-step (single-domain handles)
  long long substep_seq = 0;  // sequence number of the next sub-step (the records' stamps)
  long long carry_from = 0;   // ct.carry is as of the sub-step before this one; only younger records override it
  bool diag_valid = false; // the last sub-step produced diagnostics
  real* fhf = nullptr;  // [3][n]
  unsigned char* owner = nullptr;
  unsigned* mincov = nullptr;   // GrainFluidView::mincov
  unsigned paint_epoch = 0;
  // link sums handed from the fused kernel to the force kernel (ForceSlots, lbmdem_internal.h)
  ForceSlots fs{};
  int* gathered2 = nullptr;   // both counters (fs.gathered / fs.gathered_next alternate between them)
  bool slots_clean = false;  // every slot is empty
  bool last_forces_from_table = false;
  bool slots_valid = false;  // the table was filled by the collide_stream that produced f[fcur] with the current map
  double rmax = 0.0, rmin = 0.0;
  // strip decomposition with distributed grains (lbmdem_dist_*)
  bool dist = false, dist_poison = false;
  bool dist_period_open = false;   // lbmdem_dist_begin_period has classified the grains for the coming fluid step
  int dist_margin = 0;
  DistDevice dd{};
  VerletDevice V{};
  bool verlet_ok = false;
  bool verlet_tracks_positions = false;  // the positions have only moved by DEM sub-steps since the list was built (no upload)
  volatile int* ovf_host = nullptr;  // pinned mirror of V.overflow, refreshed (asynchronously) after every rebuild
  volatile int* ferr_host = nullptr; // pinned mirror of fs.error (strip decomposition), refreshed by every period's classification launch
  int* ferr_mirror = nullptr;        // ... its device address
  // runs of ordinary sub-steps in one launch (k_dem_chain, dem_kernels.hip)
  DemChain chain{};
  int chain_max = 128;         // longest run handed to one launch; < 2: one launch per sub-step (lbmdem_set_dem_chain)
  bool chain_checked = false;  // the residency census has run (chain.capacity says what it found)
  long chain_launches = 0, chain_substeps = 0;   // lbmdem_dem_chain_stats
  // the run of sub-steps that ended at a fluid step has rasterised the discs into obst[1 - ocur] itself (ChainPaint): the
  // next obst_construction has nothing to launch. Dropped (and the canvas marked dirty) by whatever moves a grain first.
  bool chain_painted = false;
  bool chain_paint = true;     // (lbmdem_set_dem_chain: max_substeps < 0 switches only this off, for A/B)
  long chain_paints = 0;
  int dem_tiles_mode = 0;      // how the tiles of k_dem_chain are composed (lbmdem_set_dem_tiles)
  // launches of k_dem_chain that have not been seen to finish, the calls they belong to, and how often one had to be undone
  std::vector<ChainSnap> chain_pending;
  std::vector<RunLogEntry> runlog;
  int in_run = 0;              // inside lbmdem_run / lbmdem_run_dem (their pieces do not settle on their own)
  bool run_logged = false;     // ... of a call that is in the log (a replay is not)
  long chain_recoveries = 0;
  int chain_giveup_at = -1;    // (experiment build: the launch, counted from 0, that is made to give up; lbmdem_debug_chain_giveup)
  long nbsteps = 0;
  int force_mode = 0;
  // derived scalars
  double fscale12 = 0, fscale3 = 0;
  double* dpartial = nullptr;
  // profiling of the dominant kernel
  bool prof = false;
  int prof_stride = 1;      // every prof_stride-th launch of the fused kernel is timed (two event records cost the step ~10 us)
  long prof_count = 0;
  bool prof_this = false;   // the launch being issued is one of them
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<hipEvent_t> ev2;      // end of the EDGES part of a split launch (it may run on another stream, next to INTERIOR)
  std::vector<char> ev2_set;
  size_t ev_used = 0;
};

// Every entry point starts here. Outside lbmdem_run / lbmdem_run_dem a handle with unconfirmed launches of k_dem_chain is
// settled first: the stream is drained and, if one of them gave up, the state is taken back to that launch and the calls since
// are replayed (so a download, a checkpoint, a file writer, an upload never meet a state that a failed launch left behind).
#define CHECK_H(h) do { if (!(h)) return fail(LBMDEM_EINVAL, "null handle"); HIP_TRY(hipSetDevice((h)->cfg.device)); \
                        if (!(h)->in_run && !(h)->chain_pending.empty()) RC_TRY(lbmdem_chain_settle(h)); } while (0)
#define CHECK_H_RUNNING(h) do { if (!(h)) return fail(LBMDEM_EINVAL, "null handle"); HIP_TRY(hipSetDevice((h)->cfg.device)); } while (0)
#define CHECK_NOT_SPLIT(h) do { if ((h)->cs_interior_pending) return fail(LBMDEM_EINVAL, "lbmdem_collide_stream_part(LBMDEM_CS_INTERIOR) has not been called after LBMDEM_CS_EDGES"); } while (0)

// (distributed handles only: a rank cannot go back on its own -- its neighbours have moved on; single-domain handles recover)
#define CHAIN_FAILED(h) ((h)->chain.err_host && *(h)->chain.err_host)
#define CHAIN_FAIL_MSG "a launch of the multi-sub-step DEM kernel could not finish on this rank (workgroups not co-resident?): every kernel behind it was held back, the rank's state is that of the launch's first sub-step; restart the decomposition from its last checkpoint with lbmdem_set_dem_chain(h, 0)"

// Named ranges for rocprofv3 --marker-trace around the phases of a step (LBMDEM_ROCTX=1; lbmdem_capi.hip)
struct PhaseRange {
  bool on;
  explicit PhaseRange(const char* name);
  ~PhaseRange();
};

// Host buffers at the ABI are double in both builds; the device holds `real`. (Every float is a double: downloads are
// exact; uploads of values that are not floats are rounded to nearest, like an assignment to `real` in the reference.)
static inline hipError_t h2d_real(real* dst, const double* src, size_t n, hipStream_t st) {
#ifdef LBMDEM_SINGLE_PRECISION
  std::vector<real> tmp(n);
  for (size_t k = 0; k < n; ++k) tmp[k] = (real)src[k];
  hipError_t e = hipMemcpyAsync(dst, tmp.data(), sizeof(real) * n, hipMemcpyHostToDevice, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);   // tmp dies here
  return e;
#else
  return hipMemcpyAsync(dst, src, sizeof(real) * n, hipMemcpyHostToDevice, st);
#endif
}
static inline hipError_t d2h_real(double* dst, const real* src, size_t n, hipStream_t st) {
#ifdef LBMDEM_SINGLE_PRECISION
  std::vector<real> tmp(n);
  hipError_t e = hipMemcpyAsync(tmp.data(), src, sizeof(real) * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  if (e == hipSuccess) for (size_t k = 0; k < n; ++k) dst[k] = tmp[k];
  return e;
#else
  hipError_t e = hipMemcpyAsync(dst, src, sizeof(real) * n, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  return e;
#endif
}
// The float build covers the step path and the state transfers (what its parity tests and its bench line use); the
// file writers, checkpoints, the strip decomposition and the RCCL transport exist in the double build only.
#ifdef LBMDEM_SINGLE_PRECISION
#define SP_UNAVAILABLE(what) return fail(LBMDEM_EINVAL, what " is not available in the single-precision build of the library")
#else
#define SP_UNAVAILABLE(what) do { } while (0)
#endif

static inline GrainFluidView gview(const lbmdem_handle* h) {
  const Kin& K = h->kin[h->kcur];
  return GrainFluidView{K.x1, K.x2, K.v1, K.v2, K.v3, h->xc, h->yc, h->r2, h->rbl0, h->pk, h->mincov, h->paint_epoch};
}

static inline DemParams dem_params(const lbmdem_handle* h) {
  const lbmdem_config& c = h->cfg;
  const lbmdem_physics& p = c.phys;
  DemParams P;
  P.gate = h->chain.gate;
  P.n = h->n; P.dt = (real)c.dt; P.dt2 = (real)c.dt2;
  P.kg = (real)p.kg; P.nug = (real)p.nug; P.kt = (real)p.kt; P.mu = (real)p.mu; P.murf = (real)p.murf;
  P.km = (real)p.km; P.num = (real)p.num; P.ktm = (real)p.ktm; P.mumb = (real)p.mumb; P.mum = (real)p.mum; P.nugt = (real)p.nugt;
  P.Mgx = (real)c.Mgx; P.Mdx = (real)c.Mdx; P.Mby = (real)c.Mby; P.Mhy = (real)c.Mhy;
  {
    const real amp = (real)p.amp, freq = (real)p.freq, t = (real)p.t;   // main.c:163-165
    P.wallT_vel = amp * freq * cos((double)(freq * t));                 // main.c:855: cos() is <math.h>'s
  }
  P.xG = (real)c.xG; P.yG = (real)c.yG;
  P.distVerlet = (real)p.distVerlet;
  return P;
}

// shared between the translation units
#define LBMDEM_INTERNAL extern "C" __attribute__((visibility("hidden")))
LBMDEM_INTERNAL int lbmdem_write_vtk_file(const char* path, int nx, int ny, const char* name, int dim, const float* data);
LBMDEM_INTERNAL int lbmdem_verlet_build_lists(lbmdem_handle* h);
LBMDEM_INTERNAL int lbmdem_chain_settle(lbmdem_handle* h);
LBMDEM_INTERNAL int lbmdem_dem_tiles_by_index(lbmdem_handle* h);
// the next obst_construction will update obst[1 - ocur] in place: nobody resets that canvas beforehand
static inline bool obst_update_planned(const lbmdem_handle* h) {
  return h->obst_update && !h->dist && h->snap_ok[1 - h->ocur] && h->verlet_ok && h->verlet_tracks_positions &&
         !*h->ovf_host && h->obst_reset_rows == 0 && *h->moved_host != h->list_generation;
}
// the coming ordinary sub-steps that nothing separates (fluid step when `fluid`, list rebuild, film law, table sub-step), at
// most `remaining`; 0 when the run is shorter than 2 or the multi-sub-step kernel cannot be used -- and that many sub-steps
LBMDEM_INTERNAL long lbmdem_dem_chain_length(lbmdem_handle* h, long remaining, int fluid);
LBMDEM_INTERNAL int lbmdem_dem_chain(lbmdem_handle* h, long k, int fluid);   // fluid: a fluid step may follow (the run may rasterise for it)
static inline void drop_chain_paint(lbmdem_handle* h) {
  if (h->chain_painted) { h->chain_painted = false; h->obst_reset_rows = 0; }   // (the canvas holds a picture nobody will use)
}
// pieces of the fluid step for the C transport (lbmdem_comm.hip), which runs the edge rows on its halo lane's stream:
LBMDEM_INTERNAL int lbmdem_collide_stream_prepare(lbmdem_handle* h);
LBMDEM_INTERNAL int lbmdem_collide_stream_part_on(lbmdem_handle* h, int part, hipStream_t st);   // part on stream `st`
LBMDEM_INTERNAL int lbmdem_dist_begin_period_packed(lbmdem_handle* h, void* kin_lo, void* kin_hi);
LBMDEM_INTERNAL int lbmdem_dist_unpack_tables_kin_fill(lbmdem_handle* h, const void* tab_lo, const void* tab_hi,
                                                       const void* kin_lo, const void* kin_hi);
LBMDEM_INTERNAL int lbmdem_vtk_place_block(float* fields11, int lx, int ly, int x0, int nx, const float* block11);
LBMDEM_INTERNAL int lbmdem_dist_enable_caps(lbmdem_handle* h, int M, long cap_g, long cap_t, long cap_l);
