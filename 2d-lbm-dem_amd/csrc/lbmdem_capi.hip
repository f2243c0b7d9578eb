// lbmdem_capi.hip -- the C ABI of liblbmdem_hip.so (include/lbmdem_hip.h): handle, memory, the
// step driver with the reference's cadences (renderScene, main.c:1697-1765) and state transfer in
// the reference's host layout. All arithmetic of the hot path lives in lbm_kernels.hip and
// dem_kernels.hip; the host-side arithmetic here is the one-off time-step derivation
// (main.c:1836-1860) and per-grain constants (main.c:624-626,1859), kept bit-identical.

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

static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t e_ = (expr);                                                                   \
    if (e_ != hipSuccess)                                                                     \
      return fail(LBMDEM_EHIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                  __LINE__);                                                                  \
  } while (0)

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
  // collide_stream in two parts (lbmdem_collide_stream_part): after EDGES the interior rows of f[fcur] are
  // still missing; the operands of the launch are kept for INTERIOR
  bool cs_interior_pending = false;
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
  volatile int* ferr_host = nullptr; // pinned mirror of fs.error (strip decomposition), refreshed after every period's forces
  long nbsteps = 0;
  int force_mode = 0;
  // derived scalars
  double fscale12 = 0, fscale3 = 0;
  double* dpartial = nullptr;
  // profiling of the dominant kernel
  bool prof = false;
  std::vector<hipEvent_t> ev0, ev1;
  size_t ev_used = 0;
};

// Named ranges for rocprofv3 --marker-trace around the phases of a step (obstacle map, fused fluid kernel, hydrodynamic
// forces, Verlet rebuild, DEM sub-step). libroctx64.so is looked up once, lazily, and only when LBMDEM_ROCTX is set:
// without a profiler attached the ranges cost nothing.
namespace {
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
  bool tried = false;
};
Roctx g_roctx;
void roctx_init() {
  g_roctx.tried = true;
  const char* e = getenv("LBMDEM_ROCTX");
  if (!e || !*e || *e == '0') return;
  void* lib = dlopen("libroctx64.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("libroctx64.so.4", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return;
  g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
  g_roctx.pop = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
  if (!g_roctx.push || !g_roctx.pop) { g_roctx.push = nullptr; g_roctx.pop = nullptr; }
}
struct PhaseRange {
  bool on;
  explicit PhaseRange(const char* name) {
    if (!g_roctx.tried) roctx_init();
    on = g_roctx.push != nullptr;
    if (on) g_roctx.push(name);
  }
  ~PhaseRange() { if (on) g_roctx.pop(); }
};
}  // namespace

// Host buffers at the ABI are double in both builds; the device holds `real`. (Every float is a double: downloads are
// exact; uploads of values that are not floats are rounded to nearest, like an assignment to `real` in the reference.)
static hipError_t h2d_real(real* dst, const double* src, size_t n, hipStream_t st) {
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
static hipError_t d2h_real(double* dst, const real* src, size_t n, hipStream_t st) {
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

static GrainFluidView gview(const lbmdem_handle* h) {
  const Kin& K = h->kin[h->kcur];
  return GrainFluidView{K.x1, K.x2, K.v1, K.v2, K.v3, h->xc, h->yc, h->r2, h->rbl0, h->pk, h->mincov, h->paint_epoch};
}

static DemParams dem_params(const lbmdem_handle* h) {
  const lbmdem_config& c = h->cfg;
  const lbmdem_physics& p = c.phys;
  DemParams P;
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

#pragma GCC visibility push(default)
extern "C" {

const char* lbmdem_last_error(void) { return g_err; }
const char* lbmdem_version(void) { return "lbmdem-hip 0.1 (gfx950)"; }

int lbmdem_physics_defaults(lbmdem_physics* p) {
  if (!p) return fail(LBMDEM_EINVAL, "null physics");
  p->rho_moy = 1000; p->tau = 0.504;
  p->s2 = 1.5; p->s3 = 1.4; p->s5 = 1.5; p->s7 = 1.5; p->s8 = 1.9841; p->s9 = 1.9841;
  p->nu = 1e-6; p->reductionR = 0.85;
  p->G = 9.81; p->angleG = 0.0;
  p->km = 3e+6; p->kg = 1.6e+6; p->kt = 1.0e+6; p->ktm = 2e+6;
  p->nug = 6.4e+1; p->num = 8.7e+1; p->nugt = 5e-1;
  p->mu = .5317; p->mum = .466; p->mumb = .466; p->murf = 0.01;
  p->distVerlet = 5e-4; p->dtt = 0.; p->iterDEM = 100.;
  p->freq = 5; p->amp = 4.e-4; p->t = 0;
  p->updateVerlet = 100; p->stepFilm = 8000;
  return LBMDEM_OK;
}

int lbmdem_derive(lbmdem_config* cfg, int lx, int ly, double scale, int nbgrains, const double* r) {
  if (!cfg || !r || lx < 3 || ly < 3 || nbgrains < 1 || !(scale > 0))
    return fail(LBMDEM_EINVAL, "lbmdem_derive: bad arguments");
  const lbmdem_physics& p = cfg->phys;
  cfg->lx = lx; cfg->ly = ly; cfg->scale = scale; cfg->nbgrains = nbgrains;
  // The reference's globals are `real` (main.c:52, 97-118, 201-204): locals of that type + the reference's own
  // expressions give its promotions in either build (sin, cos, sqrt are <math.h>'s double functions there).
  const real G = (real)p.G, angleG = (real)p.angleG, iterDEM = (real)p.iterDEM, kg = (real)p.kg, tau = (real)p.tau,
             nu = (real)p.nu;
  // main.c:1836-1842
  const real Mgx = 0.;
  const real Mdx = 1.e-3 * lx / 10;
  const real Mhy = 1.e-3 * ly / 10;
  const real Mby = 0.;
  const real xG = -G * sin((double)angleG);
  const real yG = -G * cos((double)angleG);
  // main.c:1844-1854
  const real dx = (1. / scale) * (Mdx - Mgx) / (lx - 1);
  real rMin = (real)r[0];
  for (int i = 1; i <= nbgrains - 1; i++) rMin = (real)fmin((double)rMin, (double)(real)r[i]);
  const real dtmax = (1 / iterDEM) * REF_PI * rMin * sqrt((double)(REF_PI * RHO_S / kg));
  const real dtLB = dx * dx * (tau - 0.5) / (3 * nu);
  const int npDEM = (dtLB / dtmax + 1);
  const real c = dx / dtLB;
  const real dt = dtLB / npDEM;
  const real dt2 = dt * dt;
  cfg->Mgx = Mgx; cfg->Mdx = Mdx; cfg->Mhy = Mhy; cfg->Mby = Mby; cfg->xG = xG; cfg->yG = yG;
  cfg->dx = dx; cfg->dtLB = dtLB; cfg->npDEM = npDEM; cfg->c = c; cfg->dt = dt; cfg->dt2 = dt2;
  return LBMDEM_OK;
}

int lbmdem_read_sample(const char* path, int* nbgrains, double** r_out, double** x1_out, double** x2_out) try {
  if (!path || !nbgrains || !r_out || !x1_out || !x2_out) return fail(LBMDEM_EINVAL, "null argument");
  FILE* fp = fopen(path, "r");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open sample '%s'", path);
  char line[256];
  int n = 0;
  if (!fgets(line, sizeof line, fp) || fscanf(fp, "%d", &n) != 1 || n <= 0) {
    fclose(fp);
    return fail(LBMDEM_EINVAL, "bad sample header in '%s'", path);
  }
  double* r = (double*)malloc(sizeof(double) * n);
  double* x1 = (double*)malloc(sizeof(double) * n);
  double* x2 = (double*)malloc(sizeof(double) * n);
  if (!r || !x1 || !x2) { fclose(fp); free(r); free(x1); free(x2); return fail(LBMDEM_ENOMEM, "host alloc"); }
  const real unit = 1e-3;  // `real r = 1e-3`, main.c:114
  for (int i = 0; i < n; ++i) {
    real v[3];   // fscanf(FLOAT_FORMAT, &g[i].r, ...): the text is converted straight to `real` (main.c:36,39,619)
    for (int k = 0; k < 3; ++k) {
      int ch;
      while ((ch = fgetc(fp)) != EOF && (isspace(ch) || ch == ';')) {}
      if (ch != EOF) ungetc(ch, fp);
      if (ch == EOF || fscanf(fp, sizeof(real) == 4 ? "%e" : "%le", &v[k]) != 1) {
        fclose(fp); free(r); free(x1); free(x2);
        return fail(LBMDEM_EINVAL, "sample '%s' truncated at grain %d", path, i);
      }
    }
    r[i] = v[0] * unit; x1[i] = v[1] * unit; x2[i] = v[2] * unit;
  }
  fclose(fp);
  *nbgrains = n; *r_out = r; *x1_out = x1; *x2_out = x2;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

void lbmdem_free_host(void* p) { free(p); }

static int paint_into(lbmdem_handle* h, int* obst) {
  const Kin& K = h->kin[h->kcur];
  if (h->mincov) {   // records of an older rasterisation lose against this one's; the 12-bit epoch is wound back rarely
    if (++h->paint_epoch > 0xFFFu) {
      HIP_TRY(hipMemsetAsync(h->mincov, 0, sizeof(unsigned) * (size_t)h->L.plane, h->stream));
      h->paint_epoch = 1;
    }
  }
  launch_obst_fill(obst, h->L, h->stream);
  launch_obst_paint(obst, h->L, h->n, K.x1, K.x2, h->r, h->rLB, K.v1, K.v2, K.v3, h->xc, h->yc, h->r2, h->rbl0, h->pk,
                    h->fs.touched, h->dist ? h->dd.fluidmask : nullptr, h->mincov, h->paint_epoch,
                    h->dist ? h->dd.local_list : nullptr, h->dist ? h->dd.counters + 6 : nullptr, h->dist ? h->dd.cap_l : 0,
                    // the pair list tells which discs cannot share a node with another one (plain stores instead of
                    // atomics). Not with distributed grains: a rank's list is only right for the grains it integrates
                    (h->verlet_ok && h->verlet_tracks_positions && !h->dist && !*h->ovf_host) ? h->V.offsets : nullptr,
                    (h->verlet_ok && h->verlet_tracks_positions && !h->dist && !*h->ovf_host) ? h->V.nbr : nullptr, h->stream);
  h->slots_valid = false;  // the grain geometry the table is indexed with has changed
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_create(const lbmdem_config* cfg, const double* r, const double* x1, const double* x2,
                  lbmdem_handle** out) try {
  if (!cfg || !r || !x1 || !x2 || !out) return fail(LBMDEM_EINVAL, "lbmdem_create: null argument");
  *out = nullptr;
  if (cfg->lx < 3 || cfg->ly < 3 || cfg->nbgrains < 1) return fail(LBMDEM_EINVAL, "bad lattice/grain count");
  if (cfg->x_begin < 0 || cfg->x_end > cfg->lx || cfg->x_begin >= cfg->x_end || cfg->halo < 0)
    return fail(LBMDEM_EINVAL, "bad strip [%d,%d) halo %d", cfg->x_begin, cfg->x_end, cfg->halo);
  if (!(cfg->dx > 0) || !(cfg->dt > 0) || cfg->npDEM < 1) return fail(LBMDEM_EINVAL, "derived block not filled (lbmdem_derive)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
    return fail(LBMDEM_ENODEVICE, "no HIP device available (this library has no CPU fallback)");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(LBMDEM_ENODEVICE, "device %d not present (%d devices)", cfg->device, ndev);
  HIP_TRY(hipSetDevice(cfg->device));

  const int n = cfg->nbgrains;
  double rmax = r[0];
  for (int i = 1; i < n; ++i) if (r[i] > rmax) rmax = r[i];
  double rmin = r[0];
  for (int i = 1; i < n; ++i) if (r[i] < rmin) rmin = r[i];
  const bool cut_lo = cfg->x_begin > 0, cut_hi = cfg->x_end < cfg->lx;
  // 2 rows: the f row and the two obstacle rows the fused kernel reads beyond a cut. (Replicated grains need
  // 2 + the largest grain radius so that an owner sees its grains' whole footprints: checked in forces_fluid.)
  if ((cut_lo || cut_hi) && cfg->halo < 2)
    return fail(LBMDEM_EINVAL, "strip decomposition needs halo >= 2 rows");

  lbmdem_handle* h = new lbmdem_handle();
  h->cfg = *cfg;
  h->n = n;
  h->rmax = rmax;
  h->rmin = rmin;
  LatticeView& L = h->L;
  L.lx = cfg->lx; L.ly = cfg->ly;
  L.gx0 = cut_lo ? cfg->x_begin - cfg->halo : 0;
  if (L.gx0 < 0) L.gx0 = 0;
  int gx1 = cut_hi ? cfg->x_end + cfg->halo : cfg->lx;
  if (gx1 > cfg->lx) gx1 = cfg->lx;
  L.nxl = gx1 - L.gx0;
  L.xo0 = cfg->x_begin - L.gx0; L.xo1 = cfg->x_end - L.gx0;
  L.sy = ((cfg->ly + LBMDEM_TILE_Y - 1) / LBMDEM_TILE_Y) * LBMDEM_TILE_Y;
  L.plane = (long)L.nxl * L.sy;
  L.n = n;
  L.dx = (real)cfg->dx; L.c = (real)cfg->c; L.Mgx = (real)cfg->Mgx; L.Mby = (real)cfg->Mby;
  const lbmdem_physics& p = cfg->phys;
  L.s2 = (real)p.s2; L.s3 = (real)p.s3; L.s5 = (real)p.s5; L.s7 = (real)p.s7; L.s8 = (real)p.s8; L.s9 = (real)p.s9;
  L.reduced_lt1 = p.reductionR < 1.0 ? 1 : 0;
  {
    const real cr = (real)cfg->c;
    const real cc = cr * cr;                 // c * c of main.c:976 (real arithmetic)
    L.rc = (real)(1.0 / cr);                 // RN(1 / b) in `real` for exact_div
    L.rcc = (real)(1.0 / cc);
    L.cc = cc;
    L.lid6 = 0.0;
    const real w_diag = 1. / 36, w_axis = 1. / 9;   // real _w[Q] (main.c:53-54)
    L.wc_diag = w_diag / cr;                 // w[iLB] / c of main.c:1174 (real arithmetic)
    L.wc_axis = w_axis / cr;
    auto all_ones = [](real v) {
#ifdef LBMDEM_SINGLE_PRECISION
      uint32_t b;
      memcpy(&b, &v, sizeof b);
      return (b & 0x7FFFFFu) == 0x7FFFFFu;
#else
      uint64_t b;
      memcpy(&b, &v, sizeof b);
      return (b & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
#endif
    };
    L.recip_ok = (!all_ones(cr) && !all_ones(cc) && cfg->c > 1e-4 && cfg->c < 1e4) ? 1 : 0;
#ifdef LBMDEM_AB
    if (const char* e = getenv("LBMDEM_TRUE_DIVISIONS")) if (atoi(e)) L.recip_ok = 0;  // A/B switch
#endif
  }
  // force scaling, main.c:1329-1331
  {  // `fhf1[i] *= rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5))` with real globals: the numerator is a real
     // product, `tau - 0.5` makes the denominator -- and the quotient, and the multiplication -- double
    const real rho_moy = (real)p.rho_moy, nu = (real)p.nu, tau = (real)p.tau, dxr = (real)cfg->dx;
    h->fscale12 = rho_moy * 9 * nu * nu / (dxr * (tau - 0.5) * (tau - 0.5));
    h->fscale3 = dxr * rho_moy * 9 * nu * nu / (dxr * (tau - 0.5) * (tau - 0.5));
  }

#define CREATE_TRY(expr)                                                                             \
  do {                                                                                               \
    hipError_t e_ = (expr);                                                                          \
    if (e_ != hipSuccess) {                                                                          \
      int rc_ = fail(e_ == hipErrorOutOfMemory ? LBMDEM_ENOMEM : LBMDEM_EHIP, "%s failed: %s", #expr, \
                     hipGetErrorString(e_));                                                         \
      lbmdem_destroy(h);                                                                             \
      return rc_;                                                                                    \
    }                                                                                                \
  } while (0)

  CREATE_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  h->stream = h->own_stream;
  const size_t fbytes = sizeof(real) * 9 * (size_t)L.plane;
  CREATE_TRY(hipMalloc((void**)&h->f[0], fbytes));
  CREATE_TRY(hipMalloc((void**)&h->f[1], fbytes));
  CREATE_TRY(hipMalloc((void**)&h->obst[0], sizeof(int) * (size_t)L.plane));
  CREATE_TRY(hipMalloc((void**)&h->obst[1], sizeof(int) * (size_t)L.plane));
  // grains: 18 kinematic + r m It rLB xc yc r2 rbl0 + 3 fhf + 8 packed + p = 38 columns
  const size_t cols = 18 + 8 + 3 + 8 + 1 + 9;
  CREATE_TRY(hipMalloc((void**)&h->gbuf, sizeof(real) * cols * n));
  CREATE_TRY(hipMemset(h->gbuf, 0, sizeof(real) * cols * n));
  {
    real* p0 = h->gbuf;
    for (int b = 0; b < 2; ++b) {
      Kin& K = h->kin[b];
      K.x1 = p0; K.x2 = p0 + n; K.x3 = p0 + 2 * n; K.v1 = p0 + 3 * n; K.v2 = p0 + 4 * n; K.v3 = p0 + 5 * n;
      K.a1 = p0 + 6 * n; K.a2 = p0 + 7 * n; K.a3 = p0 + 8 * n;
      p0 += 9 * (size_t)n;
    }
    h->r = p0; h->m = p0 + n; h->It = p0 + 2 * n; h->rLB = p0 + 3 * n;
    h->xc = p0 + 4 * n; h->yc = p0 + 5 * n; h->r2 = p0 + 6 * n; h->rbl0 = p0 + 7 * n;
    h->fhf = p0 + 8 * n;
    h->pk = p0 + 11 * (size_t)n;
    h->gp = p0 + 19 * (size_t)n;
    h->diag = p0 + 20 * (size_t)n;
  }
  CREATE_TRY(hipMalloc((void**)&h->owner, n));
  CREATE_TRY(hipMemsetAsync(h->owner, 1, n, h->stream));
  if (carry_track_alloc(h->ct, n) != 0) {
    lbmdem_destroy(h);
    return fail(LBMDEM_ENOMEM, "carry records: hipMalloc failed");
  }
  if (n < LBMDEM_MINCOV_IDS) {
    CREATE_TRY(hipMalloc((void**)&h->mincov, sizeof(unsigned) * (size_t)L.plane));
    CREATE_TRY(hipMemsetAsync(h->mincov, 0, sizeof(unsigned) * (size_t)L.plane, h->stream));
  }
  CREATE_TRY(hipMalloc((void**)&h->fs.queue, sizeof(int) * n));
  CREATE_TRY(hipMalloc((void**)&h->fs.error, sizeof(int)));
  CREATE_TRY(hipMemsetAsync(h->fs.error, 0, sizeof(int), h->stream));
  CREATE_TRY(hipMalloc((void**)&h->gathered2, 2 * sizeof(int)));
  CREATE_TRY(hipMemsetAsync(h->gathered2, 0, 2 * sizeof(int), h->stream));
  h->fs.gathered = h->gathered2; h->fs.gathered_next = h->gathered2 + 1;
  CREATE_TRY(hipMalloc((void**)&h->fs.touched, n));
  CREATE_TRY(hipMemsetAsync(h->fs.touched, 0, n, h->stream));
  {
    // lattice lines through a reduced disc, any direction: |ey dx - ex dy| <= sqrt(2) rLB, +2 for the truncated centre
    const int half = (int)ceil(1.4143 * p.reductionR * rmax / cfg->dx) + 2;
    const int spd = (2 * half + 1 + 3) & ~3;
    bool want = spd <= LBMDEM_SPD_MAX && collide_stream_fills_slots(L) && n < (1 << 18);  // grain id: 18 bits of a link descriptor
#ifdef LBMDEM_SINGLE_PRECISION
    want = false;   // the table kernel's chord geometry is calibrated for double rounding errors: float build gathers
#endif
#ifdef LBMDEM_AB
    if (const char* e = getenv("LBMDEM_NO_SLOTS")) if (atoi(e)) want = false;  // A/B: forces gathered from the lattice
#endif
    if (want) {
      h->fs.half = half;
      h->fs.spd = spd;
      h->fs.hb = (int)ceil(p.reductionR * rmax / cfg->dx) + 1;
      CREATE_TRY(hipMalloc((void**)&h->fs.tab, sizeof(real) * 8 * (size_t)spd * n));
      launch_slots_clear(h->fs, n, h->stream);
      h->slots_clean = true;
    }
  }
  CREATE_TRY(hipMalloc((void**)&h->dpartial, sizeof(double) * 1024));
  {
    // per-grain constants on the host, reference arithmetic: main.c:624-626, 1859
    std::vector<real> hr(n), hm(n), hIt(n), hrLB(n), hx1(n), hx2(n);
    const real reductionR = (real)p.reductionR, dxr = (real)cfg->dx;
    for (int i = 0; i < n; ++i) {
      hr[i] = (real)r[i]; hx1[i] = (real)x1[i]; hx2[i] = (real)x2[i];
      hm[i] = RHO_S * REF_PI * hr[i] * hr[i];      // rhoS * pi * r * r: a double product (pi), stored as real
      hIt[i] = hm[i] * hr[i] * hr[i] / 2;          // real arithmetic
      hrLB[i] = reductionR * hr[i] / dxr;
    }
    CREATE_TRY(hipMemcpy(h->r, hr.data(), sizeof(real) * n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->m, hm.data(), sizeof(real) * n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->It, hIt.data(), sizeof(real) * n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->rLB, hrLB.data(), sizeof(real) * n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->kin[0].x1, hx1.data(), sizeof(real) * n, hipMemcpyHostToDevice));
    CREATE_TRY(hipMemcpy(h->kin[0].x2, hx2.data(), sizeof(real) * n, hipMemcpyHostToDevice));
  }
  // Verlet grid over the fluid domain; grains outside are clamped into the edge cells
  {
    // (cell edge: the float build's cell coordinates carry a relative error of ~1e-7; a hair more keeps every partner
    // within the 3 x 3 cells that are scanned)
    const double cs = (2 * rmax + p.distVerlet) * (sizeof(real) == 4 ? 1.001 : 1.0);
    const double wx = cfg->dx * (cfg->lx - 1), wy = cfg->dx * (cfg->ly - 1);
    if (verlet_alloc(h->V, n, cs, cfg->Mgx, cfg->Mby, wx, wy) != 0) {
      int rc = fail(LBMDEM_ENOMEM, "verlet_alloc failed");
      lbmdem_destroy(h);
      return rc;
    }
  }
  CREATE_TRY(hipHostMalloc((void**)&h->ovf_host, sizeof(int), hipHostMallocDefault));
  *h->ovf_host = 0;
  CREATE_TRY(hipHostMalloc((void**)&h->ferr_host, sizeof(int), hipHostMallocDefault));
  *h->ferr_host = 0;
  // init_density (main.c:716-724) and init_obst (main.c:663-711)
  launch_fill_equilibrium(h->f[0], L, h->stream);
  launch_fill_equilibrium(h->f[1], L, h->stream);
  {
    int rc = paint_into(h, h->obst[0]);
    if (rc != LBMDEM_OK) { lbmdem_destroy(h); return rc; }
  }
  CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
  *out = h;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_destroy(lbmdem_handle* h) {
  if (!h) return LBMDEM_OK;
  (void)hipSetDevice(h->cfg.device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  for (int b = 0; b < 2; ++b) {
    if (h->f[b]) (void)hipFree(h->f[b]);
    if (h->obst[b]) (void)hipFree(h->obst[b]);
  }
  if (h->gbuf) (void)hipFree(h->gbuf);
  if (h->owner) (void)hipFree(h->owner);
  if (h->mincov) (void)hipFree(h->mincov);
  if (h->fs.touched) (void)hipFree(h->fs.touched);
  if (h->fs.tab) (void)hipFree(h->fs.tab);
  if (h->gathered2) (void)hipFree(h->gathered2);
  if (h->fs.queue) (void)hipFree(h->fs.queue);
  if (h->fs.error) (void)hipFree(h->fs.error);
  if (h->dpartial) (void)hipFree(h->dpartial);
  verlet_free(h->V);
  dist_free(h->dd);
  if (h->ovf_host) (void)hipHostFree((void*)h->ovf_host);
  if (h->ferr_host) (void)hipHostFree((void*)h->ferr_host);
  diag_extra_free(h->dx);
  carry_track_free(h->ct);
  for (hipEvent_t e : h->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev1) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return LBMDEM_OK;
}

#define CHECK_H(h) do { if (!(h)) return fail(LBMDEM_EINVAL, "null handle"); HIP_TRY(hipSetDevice((h)->cfg.device)); } while (0)

int lbmdem_obst_construction(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:obst_construction");
  if (h->dist && !h->dist_period_open)
    return fail(LBMDEM_EINVAL, "distributed grains: lbmdem_dist_begin_period comes before the fluid step");
  if (h->cs_interior_pending) return fail(LBMDEM_EINVAL, "lbmdem_collide_stream_part(LBMDEM_CS_INTERIOR) has not been called after LBMDEM_CS_EDGES");
  int rc = paint_into(h, h->obst[1 - h->ocur]);
  if (rc == LBMDEM_OK) h->obst_pending = true;
  return rc;
}

// The slot table the next fused-kernel launch fills (tab == nullptr: none). A table that still holds sums nobody
// consumed is emptied first.
static ForceSlots slots_for_launch(lbmdem_handle* h) {
  ForceSlots S = h->fs;
  if (S.tab) {
    if (!h->slots_clean) launch_slots_clear(h->fs, h->n, h->stream);
    h->slots_clean = false;
  }
  return S;
}

static int prof_begin(lbmdem_handle* h, hipEvent_t* e1) {
  *e1 = nullptr;
  if (!h->prof) return LBMDEM_OK;
  if (h->ev_used == h->ev0.size()) {
    hipEvent_t a, b;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    h->ev0.push_back(a); h->ev1.push_back(b);
  }
  hipEvent_t e0 = h->ev0[h->ev_used];
  *e1 = h->ev1[h->ev_used];
  ++h->ev_used;
  HIP_TRY(hipEventRecord(e0, h->stream));
  return LBMDEM_OK;
}

#define CHECK_NOT_SPLIT(h) do { if ((h)->cs_interior_pending) return fail(LBMDEM_EINVAL, "lbmdem_collide_stream_part(LBMDEM_CS_INTERIOR) has not been called after LBMDEM_CS_EDGES"); } while (0)

int lbmdem_collide_stream(lbmdem_handle* h) try {
  CHECK_H(h);
  PhaseRange range_("lbmdem:collide_stream");
  CHECK_NOT_SPLIT(h);
  const int* ob_old = h->obst[h->ocur];
  const int* ob_new = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
  hipEvent_t e1 = nullptr;
  int rc = prof_begin(h, &e1);
  if (rc != LBMDEM_OK) return rc;
  launch_collide_stream(h->f[h->fcur], h->f[1 - h->fcur], ob_old, ob_new, h->L, gview(h), slots_for_launch(h), h->stream);
  if (e1) HIP_TRY(hipEventRecord(e1, h->stream));
  h->slots_valid = h->fs.tab != nullptr;
  HIP_TRY(hipGetLastError());
  h->fcur = 1 - h->fcur;
  if (h->obst_pending) { h->ocur = 1 - h->ocur; h->obst_pending = false; }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_collide_stream_part(lbmdem_handle* h, int part) try {
  CHECK_H(h);
  PhaseRange range_("lbmdem:collide_stream_part");
  const LatticeView& L = h->L;
  if (part == LBMDEM_CS_EDGES) {
    CHECK_NOT_SPLIT(h);
    const int H = h->cfg.halo;
    const bool has_lo = L.gx0 + L.xo0 > 0, has_hi = L.gx0 + L.xo1 < L.lx;  // interior cuts
    int lo_end = has_lo ? L.xo0 + H : L.xo0;
    if (lo_end > L.xo1) lo_end = L.xo1;
    int hi_begin = has_hi ? L.xo1 - H : L.xo1;
    if (hi_begin < lo_end) hi_begin = lo_end;
    h->cs_fin = h->f[h->fcur];
    h->cs_ob_old = h->obst[h->ocur];
    h->cs_ob_new = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
    h->cs_lo_end = lo_end; h->cs_hi_begin = hi_begin;
    hipEvent_t e1 = nullptr;  // the interval of this step's fused kernels ends in INTERIOR
    int rc = prof_begin(h, &e1);
    if (rc != LBMDEM_OK) return rc;
    h->cs_slots = slots_for_launch(h);
    launch_collide_stream_edges(h->cs_fin, h->f[1 - h->fcur], h->cs_ob_old, h->cs_ob_new, L, gview(h), h->cs_slots, L.xo0,
                                lo_end, hi_begin, L.xo1, h->stream);
    HIP_TRY(hipGetLastError());
    h->fcur = 1 - h->fcur;
    if (h->obst_pending) { h->ocur = 1 - h->ocur; h->obst_pending = false; }
    h->cs_interior_pending = true;
    return LBMDEM_OK;
  }
  if (part == LBMDEM_CS_INTERIOR) {
    if (!h->cs_interior_pending) return fail(LBMDEM_EINVAL, "LBMDEM_CS_INTERIOR without a preceding LBMDEM_CS_EDGES");
    if (h->cs_hi_begin > h->cs_lo_end) {
      LatticeView Ls = L;
      Ls.xo0 = h->cs_lo_end; Ls.xo1 = h->cs_hi_begin;
      // the grain records are those of EDGES: nothing moves the grains between the two parts
      launch_collide_stream(h->cs_fin, h->f[h->fcur], h->cs_ob_old, h->cs_ob_new, Ls, gview(h), h->cs_slots, h->stream);
    }
    h->slots_valid = h->cs_slots.tab != nullptr;
    if (h->prof && h->ev_used > 0) HIP_TRY(hipEventRecord(h->ev1[h->ev_used - 1], h->stream));
    HIP_TRY(hipGetLastError());
    h->cs_interior_pending = false;
    return LBMDEM_OK;
  }
  return fail(LBMDEM_EINVAL, "part must be LBMDEM_CS_EDGES or LBMDEM_CS_INTERIOR");
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_forces_fluid(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:forces_fluid");
  CHECK_NOT_SPLIT(h);
  const int* ob = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
  {
    const bool cut = h->cfg.x_begin > 0 || h->cfg.x_end < h->cfg.lx;
    const int need = 2 + (int)ceil(h->rmax / h->cfg.dx);
    if (cut && !h->dist && h->cfg.halo < need)
      return fail(LBMDEM_EINVAL, "strips with replicated grains need halo >= %d rows (2 + largest grain radius in nodes) "
                                 "for the hydrodynamic forces; or distribute the grains (lbmdem_dist_enable)", need);
    if (h->dist && (h->force_mode != 0 || !h->slots_valid || h->obst_pending))
      return fail(LBMDEM_EINVAL, "distributed grains: forces_fluid must follow collide_stream directly (parity force kernel)");
  }
  if (h->slots_valid && !h->obst_pending) {
    // the link sums were left in the slot table by the fused kernel; the kernel empties the table again
    { int* t = h->fs.gathered; h->fs.gathered = h->fs.gathered_next; h->fs.gathered_next = t; }  // zeroed by the last queue kernel
    launch_forces_slots(h->f[h->fcur], ob, h->L, gview(h), h->fs, h->fscale12, h->fscale3, h->fhf, h->owner,
                        h->force_mode != 0, h->stream);
    h->last_forces_from_table = true;
    h->dist_period_open = false;
    h->slots_valid = false;
    h->slots_clean = true;
  } else if (h->force_mode == 0) {
    h->last_forces_from_table = false;
    launch_forces_parity(h->f[h->fcur], ob, h->L, gview(h), h->fscale12, h->fscale3, h->fhf, h->owner, h->stream);
  } else
    launch_forces_fast(h->f[h->fcur], ob, h->L, gview(h), h->fscale12, h->fscale3, h->fhf, h->owner, h->stream);
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_force_stats(lbmdem_handle* h, int* from_table, int* gathered) {
  CHECK_H(h);
  int g = h->n;
  if (h->last_forces_from_table) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    HIP_TRY(hipMemcpy(&g, h->fs.gathered, sizeof(int), hipMemcpyDeviceToHost));
  }
  if (from_table) *from_table = h->last_forces_from_table ? h->n - g : 0;
  if (gathered) *gathered = g;
  return LBMDEM_OK;
}

// Which of the size-dependent fast paths this handle runs with (they switch off by size, silently otherwise):
// info[0] = link-sum table in use (needs < 2^18 grains, reduced radius < ~20 nodes, reductionR < 1), info[1] = its slots
// per direction, info[2] = the lowest-cover record that makes `act` exact where three or more discs overlap (needs
// < 2^20 grains; without it the two-disc rule applies), info[3] = the marching fused kernel (reductionR < 1).
int lbmdem_path_info(lbmdem_handle* h, int* info4) {
  if (!h || !info4) return fail(LBMDEM_EINVAL, "null argument");
  info4[0] = h->fs.tab != nullptr ? 1 : 0;
  info4[1] = h->fs.tab != nullptr ? h->fs.spd : 0;
  info4[2] = h->mincov != nullptr ? 1 : 0;
  info4[3] = h->L.reduced_lt1 ? 1 : 0;
  return LBMDEM_OK;
}

#ifdef LBMDEM_AB
int lbmdem_debug_gather_queue(lbmdem_handle* h, int* out, int cap) {   // experiment builds only
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int g = 0;
  HIP_TRY(hipMemcpy(&g, h->fs.gathered, sizeof(int), hipMemcpyDeviceToHost));
  if (g > cap) g = cap;
  if (g > 0) HIP_TRY(hipMemcpy(out, h->fs.queue, sizeof(int) * g, hipMemcpyDeviceToHost));
  return g;
}
#endif

int lbmdem_lbm_step(lbmdem_handle* h) {
  int rc = lbmdem_obst_construction(h);
  if (rc == LBMDEM_OK) rc = lbmdem_collide_stream(h);
  if (rc == LBMDEM_OK) rc = lbmdem_forces_fluid(h);
  return rc;
}

int lbmdem_verlet_rebuild(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:verlet_rebuild");
  // VerletWall moves the right/top DEM walls: main.c:1555-1561
  lbmdem_config& c = h->cfg;
  if (h->nbsteps * c.dt < c.phys.dtt) {
    c.Mdx = 1.e-3 * c.lx / 10;
    c.Mhy = (1.e-3 * c.ly / 10);
  } else {
    c.Mdx = 1.e-3 * c.lx;
    c.Mhy = 1.e-3 * c.ly;
  }
  if (*h->ovf_host) return fail(LBMDEM_ENOMEM, "Verlet list overflow at the previous rebuild (more than %ld symmetric entries)", h->V.cap);
  const int e = launch_verlet_rebuild(h->V, h->kin[h->kcur], h->r, dem_params(h), h->stream);
  if (e != 0) return fail(LBMDEM_EHIP, "Verlet rebuild failed: %s", hipGetErrorString((hipError_t)e));
  // a truncated list is flagged on the device; the flag travels to the host behind the rebuild and is looked at
  // by the next sub-step that finds it set, by the next rebuild and by lbmdem_sync (no stall here)
  HIP_TRY(hipMemcpyAsync((void*)h->ovf_host, h->V.overflow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  h->verlet_ok = true;
  h->verlet_tracks_positions = true;
  return LBMDEM_OK;
}

int lbmdem_dem_substep(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:dem_substep");
  CHECK_NOT_SPLIT(h);
  if (!h->verlet_ok) return fail(LBMDEM_EINVAL, "lbmdem_dem_substep before the first lbmdem_verlet_rebuild");
  if (*h->ovf_host) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  const int film = (h->nbsteps % h->cfg.phys.stepFilm == 0) ? 1 : 0;  // main.c:1342
  // contact diagnostics are only needed by write_DEM, which renderScene calls when the step counter
  // reaches a multiple of stepStrob = 4000 (main.c:142,1773): produce them in exactly that sub-step
  // (with distributed grains the order-dependent diagnostics are not produced: they thread through ALL grains in
  // index order, and write_DEM is a single-GPU output)
  const bool want_table = !h->dist && (h->diag_always || ((h->nbsteps + 1) % 4000 == 0));
  // fr, ice, slip, rw read "previous contact" carries that thread from sub-step to sub-step (main.c:130-131): every
  // ordinary sub-step leaves per-tile records of its last contacts (CarryTrack), resolved just before a table sub-step
  const bool want_diag = want_table;
  if (want_diag && !h->dx_ready) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (diag_extra_alloc(h->dx, h->n, h->V.cap, h->ct.carry) != 0) return fail(LBMDEM_ENOMEM, "diagnostic buffers: hipMalloc failed");
    h->dx_ready = true;
  }
  if (want_diag && h->carry_from < h->substep_seq) launch_carry_resolve(h->ct, h->carry_from, h->stream);
  // strips with distributed grains: every rank records the last contacts of the grains it OWNS; the table sub-step itself
  // is run by one rank on a full replica (lbmdem_dist_table_substep)
  const CarryTrack* track = want_diag ? nullptr : &h->ct;
  const bool table_cadence = (h->nbsteps + 1) % 4000 == 0;
#ifdef LBMDEM_NO_CARRY_TRACK   /* experiment: shows what the records are for (tests/test_gpu_dem_output.py fails) */
  track = nullptr;
#endif
  const DemParams P = dem_params(h);
  launch_dem_substep(h->kin[h->kcur], h->kin[1 - h->kcur], h->r, h->m, h->It, h->fhf, h->V, h->gp,
                     P, film, want_diag ? h->diag : nullptr, want_diag ? &h->dx : nullptr,
                     h->dist ? h->dd.active : nullptr, track, h->substep_seq, h->dist ? h->owner : nullptr,
                     h->stream);
  if (h->dist && h->dist_poison) launch_dist_poison(h->dd, h->kin[0], h->kin[1], h->n, h->stream);
  if (want_diag) {
    launch_diag_extra(h->dx, h->kin[h->kcur], h->r, h->V, P, film, h->stream);   // leaves the carries as of this sub-step
    h->carry_from = h->substep_seq + 1;
  }
  // (distributed grains: the rank holding the full replica ran this sub-step as a table sub-step and now holds the
  // carries; the records the other ranks have left up to here are superseded on every rank alike)
  if (h->dist && table_cadence) h->carry_from = h->substep_seq + 1;
  h->substep_seq++;
  h->diag_valid = want_table;
  HIP_TRY(hipGetLastError());
  h->kcur = 1 - h->kcur;
  h->nbsteps++;
  return LBMDEM_OK;
}

int lbmdem_run(lbmdem_handle* h, long n_dem_steps) {
  CHECK_H(h);
  for (long k = 0; k < n_dem_steps; ++k) {
    int rc = LBMDEM_OK;
    if (h->nbsteps % h->cfg.npDEM == 0) rc = lbmdem_lbm_step(h);                                  // main.c:1710-1718
    if (rc == LBMDEM_OK && h->nbsteps % h->cfg.phys.updateVerlet == 0) rc = lbmdem_verlet_rebuild(h);  // main.c:1721-1724
    if (rc == LBMDEM_OK) rc = lbmdem_dem_substep(h);                                              // main.c:1733-1764
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
}

int lbmdem_run_dem(lbmdem_handle* h, long n_dem_steps) {
  CHECK_H(h);
  for (long k = 0; k < n_dem_steps; ++k) {
    int rc = LBMDEM_OK;
    if (h->nbsteps % h->cfg.phys.updateVerlet == 0) rc = lbmdem_verlet_rebuild(h);  // main.c:1721-1724
    if (rc == LBMDEM_OK) rc = lbmdem_dem_substep(h);                                // main.c:1733-1764
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
}

int lbmdem_set_lid(lbmdem_handle* h, double uw_h) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  h->L.lid6 = uw_h / 6;   // the reference's commented-out expression: -uw_h/6, +uw_h/6
  return LBMDEM_OK;
}

int lbmdem_set_force_mode(lbmdem_handle* h, int mode) {
  if (!h || (mode != 0 && mode != 1)) return fail(LBMDEM_EINVAL, "bad force mode");
  h->force_mode = mode;
  return LBMDEM_OK;
}

// No C++ exception may cross the C ABI: every entry point that allocates host memory (std::vector, new) is a
// function-try-block that turns std::bad_alloc into LBMDEM_ENOMEM.

// ---- state transfer ---------------------------------------------------------------------------

int lbmdem_upload_f(lbmdem_handle* h, const double* f_aos) {
  CHECK_H(h);
  if (!f_aos) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const size_t cnt = (size_t)L.nxl * L.ly * 9;
  real* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, sizeof(real) * cnt));
  hipError_t e = h2d_real(tmp, f_aos + (size_t)L.gx0 * L.ly * 9, cnt, h->stream);
  h->slots_valid = false;  // the populations the link sums were formed from are being replaced
  if (e == hipSuccess) { launch_aos_to_soa(tmp, h->f[h->fcur], L, h->stream); e = hipStreamSynchronize(h->stream); }
  (void)hipFree(tmp);
  HIP_TRY(e);
  return LBMDEM_OK;
}

int lbmdem_download_f(lbmdem_handle* h, double* f_aos) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!f_aos) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const int rows = L.xo1 - L.xo0;
  const size_t cnt = (size_t)rows * L.ly * 9;
  real* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, sizeof(real) * cnt));
  launch_soa_to_aos(h->f[h->fcur], tmp, L, L.xo0, rows, h->stream);
  hipError_t e = d2h_real(f_aos + (size_t)(L.gx0 + L.xo0) * L.ly * 9, tmp, cnt, h->stream);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return LBMDEM_OK;
}

int lbmdem_download_obst(lbmdem_handle* h, int* obst) {
  CHECK_H(h);
  if (!obst) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const int* ob = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
  const int rows = L.xo1 - L.xo0;
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy2D(obst + (size_t)(L.gx0 + L.xo0) * L.ly, sizeof(int) * L.ly, ob + (size_t)L.xo0 * L.sy,
                      sizeof(int) * L.sy, sizeof(int) * L.ly, rows, hipMemcpyDeviceToHost));
  return LBMDEM_OK;
}

int lbmdem_download_macro(lbmdem_handle* h, double* rho, double* ux, double* uy) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!rho || !ux || !uy) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const int rows = L.xo1 - L.xo0;
  const size_t cnt = (size_t)rows * L.ly;
  real* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, sizeof(real) * cnt * 3));
  launch_macro(h->f[h->fcur], L, L.xo0, rows, tmp, tmp + cnt, tmp + 2 * cnt, h->stream);
  const size_t off = (size_t)(L.gx0 + L.xo0) * L.ly;
  hipError_t e = d2h_real(rho + off, tmp, cnt, h->stream);
  if (e == hipSuccess) e = d2h_real(ux + off, tmp + cnt, cnt, h->stream);
  if (e == hipSuccess) e = d2h_real(uy + off, tmp + 2 * cnt, cnt, h->stream);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return LBMDEM_OK;
}

int lbmdem_total_density(lbmdem_handle* h, double* sum) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!sum) return fail(LBMDEM_EINVAL, "null buffer");
  const int nb = 1024;
  launch_density_partial(h->f[h->fcur], h->L, h->dpartial, nb, h->stream);
  double part[1024];
  HIP_TRY(hipMemcpyAsync(part, h->dpartial, sizeof(double) * nb, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  double s = 0.0;
  for (int k = 0; k < nb; ++k) s += part[k];
  *sum = s;
  return LBMDEM_OK;
}

// check_density / final_density exactly as the reference adds them (main.c:1249-1273): one serial chain over
// f[x][y][q], continued from `sum_in` over this handle's owned rows (a strip passes its result on to the next strip).
// The printed line `final_density: %f` is what the reference's own benchmark parses (benchmark.xml:99-102).
int lbmdem_total_density_serial(lbmdem_handle* h, double sum_in, double* sum_out, int* rows_replayed) try {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!sum_out) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const int rows = L.xo1 - L.xo0;
  const real* f = h->f[h->fcur];
  double* d_rowsum = nullptr; int* d_k = nullptr; unsigned long long* d_q = nullptr; int* d_flag = nullptr; real* d_row = nullptr;
  std::vector<double> rowsum(rows);
  std::vector<real> rowbuf((size_t)L.ly * 9);
  std::vector<int> kexp(rows), flag(rows);
  std::vector<unsigned long long> quanta(rows);
  hipError_t e = hipMalloc((void**)&d_rowsum, sizeof(double) * rows);
  if (e == hipSuccess) e = hipMalloc((void**)&d_k, sizeof(int) * rows);
  if (e == hipSuccess) e = hipMalloc((void**)&d_q, sizeof(unsigned long long) * rows);
  if (e == hipSuccess) e = hipMalloc((void**)&d_flag, sizeof(int) * rows);
  if (e == hipSuccess) e = hipMalloc((void**)&d_row, sizeof(real) * L.ly * 9);
  auto cleanup = [&] { (void)hipFree(d_rowsum); (void)hipFree(d_k); (void)hipFree(d_q); (void)hipFree(d_flag); (void)hipFree(d_row); };
  if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
  // pass 1: approximate row sums -> the binade the running sum is (most probably) in when it reaches each row
  launch_density_rowsum(f, L, d_rowsum, h->stream);
  e = hipMemcpyAsync(rowsum.data(), d_rowsum, sizeof(double) * rows, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
  {
    double approx = sum_in;
    for (int r = 0; r < rows; ++r) {
      int ex = 0;
      if (approx > 0.0 && isfinite(approx)) (void)frexp(approx, &ex);   // approx = m * 2^ex, m in [0.5, 1)
      kexp[r] = ex - 1;
      approx += rowsum[r];
    }
  }
  // pass 2: integer quanta per row for that binade
  e = hipMemcpyAsync(d_k, kexp.data(), sizeof(int) * rows, hipMemcpyHostToDevice, h->stream);
  if (e == hipSuccess) {
    launch_density_rowquanta(f, L, d_k, d_q, d_flag, h->stream);
    e = hipMemcpyAsync(quanta.data(), d_q, sizeof(unsigned long long) * rows, hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipMemcpyAsync(flag.data(), d_flag, sizeof(int) * rows, hipMemcpyDeviceToHost, h->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
  // the chain, with the exact running sum
  volatile real s = (real)sum_in;   // the reference's accumulator is a `real` (main.c:1251); volatile: every addition rounded to it
  int replayed = 0;
  for (int r = 0; r < rows; ++r) {
    bool fast = false;
    const double sv = s;
    if (!flag[r] && sv > 0.0 && isfinite(sv) && quanta[r] < (1ull << LBMDEM_REAL_MANT)) {
      int ex = 0;
      (void)frexp(sv, &ex);
      if (ex - 1 == kexp[r]) {
        const double u = ldexp(1.0, kexp[r] - (LBMDEM_REAL_MANT - 1));
        const double add = (double)quanta[r] * u;          // exact: quanta < 2^p, u a power of two
        const double top = ldexp(1.0, kexp[r] + 1);
        const double t = sv + add;                          // exact while the result stays below 2^(k+1) (multiples of u)
        if (t < top) { s = (real)t; fast = true; }
      }
    }
    if (fast) continue;
    // replay this row element by element in the reference's order (y, then q)
    launch_soa_to_aos(f, d_row, L, L.xo0 + r, 1, h->stream);
    e = hipMemcpyAsync(rowbuf.data(), d_row, sizeof(real) * L.ly * 9, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) { cleanup(); HIP_TRY(e); }
    const size_t cnt = (size_t)L.ly * 9;
    for (size_t i = 0; i < cnt; ++i) s = s + rowbuf[i];
    ++replayed;
  }
  cleanup();
  *sum_out = s;
  if (rows_replayed) *rows_replayed = replayed;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_upload_kinematics(lbmdem_handle* h, const double* k9) try {
  CHECK_H(h);
  if (!k9) return fail(LBMDEM_EINVAL, "null buffer");
  const int n = h->n;
  std::vector<real> soa(9 * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 9; ++c) soa[(size_t)c * n + i] = (real)k9[(size_t)i * 9 + c];
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->kin[h->kcur].x1, soa.data(), sizeof(real) * 9 * n, hipMemcpyHostToDevice));
  h->verlet_tracks_positions = false;   // the pair list no longer bounds which discs can meet (obst_construction: atomics)
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_download_kinematics(lbmdem_handle* h, double* k9) try {
  CHECK_H(h);
  if (!k9) return fail(LBMDEM_EINVAL, "null buffer");
  const int n = h->n;
  std::vector<real> soa(9 * (size_t)n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(soa.data(), h->kin[h->kcur].x1, sizeof(real) * 9 * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 9; ++c) k9[(size_t)i * 9 + c] = soa[(size_t)c * n + i];
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_download_fhf(lbmdem_handle* h, double* fhf3) try {
  CHECK_H(h);
  if (!fhf3) return fail(LBMDEM_EINVAL, "null buffer");
  const int n = h->n;
  std::vector<real> soa(3 * (size_t)n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(soa.data(), h->fhf, sizeof(real) * 3 * n, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i)
    for (int c = 0; c < 3; ++c) fhf3[(size_t)i * 3 + c] = soa[(size_t)c * n + i];
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_download_verlet(lbmdem_handle* h, int* cumul, int* neighbours, int cap, int* npairs,
                           int* wallflags) try {
  SP_UNAVAILABLE("the Verlet list download");
  CHECK_H(h);
  if (!h->verlet_ok) return fail(LBMDEM_EINVAL, "no Verlet list built yet");
  const int n = h->n;
  std::vector<int> off(n + 1);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int ovf = 0;
  HIP_TRY(hipMemcpy(&ovf, h->V.overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (ovf) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  HIP_TRY(hipMemcpy(off.data(), h->V.offsets, sizeof(int) * (n + 1), hipMemcpyDeviceToHost));
  std::vector<int> nb(off[n] > 0 ? off[n] : 1);
  if (off[n] > 0) HIP_TRY(hipMemcpy(nb.data(), h->V.nbr, sizeof(int) * off[n], hipMemcpyDeviceToHost));
  // reference form: for each i the partners j > i, ascending; cumul[i] = running end offset,
  // never written for the last grain (main.c:1526-1540 with the memset of main.c:1816)
  int cnt = 0;
  for (int i = 0; i < n; ++i) {
    for (int k = off[i]; k < off[i + 1]; ++k) {
      if (nb[k] > i) {
        if (neighbours && cnt < cap) neighbours[cnt] = nb[k];
        ++cnt;
      }
    }
    if (cumul) cumul[i] = (i < n - 1) ? cnt : 0;
  }
  if (npairs) *npairs = cnt;
  if (wallflags) {
    std::vector<unsigned char> wf(n);
    HIP_TRY(hipMemcpy(wf.data(), h->V.wallflags, n, hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) wallflags[i] = wf[i];
  }
  if (neighbours && cnt > cap) return fail(LBMDEM_EINVAL, "neighbours[] too small: need %d", cnt);
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_download_grain_pressure(lbmdem_handle* h, double* p) {
  CHECK_H(h);
  if (!p) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(d2h_real(p, h->gp, (size_t)h->n, h->stream));
  return LBMDEM_OK;
}

int lbmdem_download_vtk_fields(lbmdem_handle* h, float* grain_pressure, float* grain_velocity,
                               float* grain_acceleration, float* fluid_pressure, float* fluid_velocity) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!grain_pressure || !grain_velocity || !grain_acceleration || !fluid_pressure || !fluid_velocity)
    return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const size_t cnt = (size_t)(L.xo1 - L.xo0) * L.ly;
  float* tmp = nullptr;
  HIP_TRY(hipMalloc((void**)&tmp, sizeof(float) * cnt * 11));
  float *d_gp = tmp, *d_gv = tmp + cnt, *d_ga = tmp + 4 * cnt, *d_fp = tmp + 7 * cnt, *d_fv = tmp + 8 * cnt;
  const int* ob = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
  const Kin& K = h->kin[h->kcur];
  launch_vtk_fields(h->f[h->fcur], ob, L, h->gp, K.v1, K.v2, K.a1, K.a2, h->cfg.phys.rho_moy, d_gp, d_gv, d_ga,
                    d_fp, d_fv, h->stream);
  hipError_t e = hipStreamSynchronize(h->stream);
  if (e == hipSuccess) e = hipMemcpy(grain_pressure, d_gp, sizeof(float) * cnt, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(grain_velocity, d_gv, sizeof(float) * cnt * 3, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(grain_acceleration, d_ga, sizeof(float) * cnt * 3, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(fluid_pressure, d_fp, sizeof(float) * cnt, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(fluid_velocity, d_fv, sizeof(float) * cnt * 3, hipMemcpyDeviceToHost);
  (void)hipFree(tmp);
  HIP_TRY(e);
  return LBMDEM_OK;
}

// One legacy-VTK file: binary, big-endian float32, RECTILINEAR_GRID with one point-data variable --
// the byte layout the reference obtains from write_rectilinear_mesh(..., useBinary = 1, ...)
// (main.c:326-328): header, DIMENSIONS, X/Y/Z_COORDINATES, CELL_DATA, POINT_DATA, one SCALARS
// (+ LOOKUP_TABLE default) or VECTORS block, no separators after binary blocks.
static void put_be(FILE* fp, const float* v, size_t n) {
  std::vector<unsigned char> buf(n * 4);
  for (size_t k = 0; k < n; ++k) {
    unsigned char b[4];
    memcpy(b, &v[k], 4);
    buf[4 * k] = b[3]; buf[4 * k + 1] = b[2]; buf[4 * k + 2] = b[1]; buf[4 * k + 3] = b[0];
  }
  fwrite(buf.data(), 1, buf.size(), fp);
}

static int write_vtk_file(const char* path, int nx, int ny, const char* name, int dim, const float* data) {
  FILE* fp = fopen(path, "w+");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open '%s' for writing", path);
  fprintf(fp, "# vtk DataFile Version 2.0\nWritten using VisIt writer\nBINARY\n");
  fprintf(fp, "DATASET RECTILINEAR_GRID\nDIMENSIONS %d %d 1\n", nx, ny);
  // coordinates: i * (float)(1/nx) on BOTH axes, z = 0 (main.c:255-258)
  const float pas = 1. / nx;
  std::vector<float> xs(nx), ys(ny);
  for (int i = 0; i < nx; ++i) xs[i] = i * pas;
  for (int i = 0; i < ny; ++i) ys[i] = i * pas;
  const float z = 0.f;
  fprintf(fp, "X_COORDINATES %d float\n", nx); put_be(fp, xs.data(), nx);
  fprintf(fp, "Y_COORDINATES %d float\n", ny); put_be(fp, ys.data(), ny);
  fprintf(fp, "Z_COORDINATES 1 float\n"); put_be(fp, &z, 1);
  fprintf(fp, "CELL_DATA %d\nPOINT_DATA %d\n", (nx - 1) * (ny - 1), nx * ny);
  if (dim == 1) fprintf(fp, "SCALARS %s float\nLOOKUP_TABLE default\n", name);
  else fprintf(fp, "VECTORS %s float\n", name);
  put_be(fp, data, (size_t)nx * ny * dim);
  fclose(fp);
  return LBMDEM_OK;
}

int lbmdem_write_vtk(lbmdem_handle* h, const char* dir, int nfile) try {
  CHECK_H(h);
  const LatticeView& L = h->L;
  if (L.xo0 != 0 || L.xo1 != L.lx || L.gx0 != 0)
    return fail(LBMDEM_EINVAL, "lbmdem_write_vtk needs the whole lattice on this handle; gather strips with "
                               "lbmdem_download_vtk_fields");
  const size_t cnt = (size_t)L.lx * L.ly;
  std::vector<float> gp(cnt), gv(3 * cnt), ga(3 * cnt), fp(cnt), fv(3 * cnt);
  int rc = lbmdem_download_vtk_fields(h, gp.data(), gv.data(), ga.data(), fp.data(), fv.data());
  if (rc != LBMDEM_OK) return rc;
  const char* names[5] = {"grain_pressure", "grain_velocity", "grain_acceleration", "fluid_pressure", "fluid_velocity"};
  const int dims[5] = {1, 3, 3, 1, 3};
  const float* data[5] = {gp.data(), gv.data(), ga.data(), fp.data(), fv.data()};
  for (int k = 0; k < 5; ++k) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s_%.6i.vtk", (dir && *dir) ? dir : ".", names[k], nfile);  // main.c:241-249
    rc = write_vtk_file(path, L.lx, L.ly, names[k], dims[k], data[k]);
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_set_diagnostics(lbmdem_handle* h, int always) {
  SP_UNAVAILABLE("the write_DEM diagnostics table");
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  h->diag_always = always != 0;
  return LBMDEM_OK;
}

// 30 columns per grain in the reference's struct order (main.c:182-197):
// x1 x2 x3 v1 v2 v3 a1 a2 a3 r m mw It p s f1 f2 ifm fm fr ifr M11 M12 M21 M22 ice slip rw z zz
int lbmdem_download_grain_table(lbmdem_handle* h, double* t) try {
  SP_UNAVAILABLE("the write_DEM diagnostics table");
  CHECK_H(h);
  if (!t) return fail(LBMDEM_EINVAL, "null buffer");
  if (!h->diag_valid) return fail(LBMDEM_EINVAL, "no contact diagnostics for the last sub-step (lbmdem_set_diagnostics, or "
                                                 "the sub-step that reaches a multiple of 4000)");
  const int n = h->n;
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> kin(9 * (size_t)n), rr(n), mm(n), it(n), gp(n), dg(9 * (size_t)n), ex(4 * (size_t)n);
  HIP_TRY(hipMemcpy(ex.data(), h->dx.fr, sizeof(double) * 4 * n, hipMemcpyDeviceToHost));  // fr, ice, slip, rw
  HIP_TRY(hipMemcpy(kin.data(), h->kin[h->kcur].x1, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(rr.data(), h->r, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(mm.data(), h->m, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(it.data(), h->It, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(gp.data(), h->gp, sizeof(double) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(dg.data(), h->diag, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
  const int* zi = reinterpret_cast<const int*>(dg.data() + 8 * (size_t)n);
  const lbmdem_config& c = h->cfg;
  for (int i = 0; i < n; ++i) {
    double* o = t + (size_t)i * 30;
    for (int k = 0; k < 9; ++k) o[k] = kin[(size_t)k * n + i];
    o[9] = rr[i]; o[10] = mm[i]; o[11] = 0.0; o[12] = it[i];
    o[13] = gp[i]; o[14] = dg[i]; o[15] = dg[(size_t)n + i]; o[16] = dg[2 * (size_t)n + i];
    o[17] = dg[3 * (size_t)n + i];
    const int z = zi[i], zz = zi[n + i];
    o[18] = (z == 0) ? 0. : o[17] / z;  // fm, main.c:409-412
    o[19] = ex[i];                      // fr
    // ifr, main.c:388-390
    o[20] = fabs(((o[10] * c.phys.G + o[16]) * (c.dt * o[4] + c.dt2 * o[7] / 2.)) + (o[15] * (c.dt * o[3] + c.dt2 * o[6] / 2.)));
    o[21] = dg[4 * (size_t)n + i]; o[22] = dg[5 * (size_t)n + i]; o[23] = dg[6 * (size_t)n + i]; o[24] = dg[7 * (size_t)n + i];
    o[25] = ex[(size_t)n + i]; o[26] = ex[2 * (size_t)n + i]; o[27] = ex[3 * (size_t)n + i];  // ice, slip, rw
    o[28] = z; o[29] = zz;
  }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// write_DEM, main.c:340-438: DEM%06d.dat (28 tab-separated columns per grain) and one line appended to
// stats.data. energies8 (may be NULL): KE, PE, SE, IFR, WF, INCE, TSLIP, TRW.
int lbmdem_write_dem(lbmdem_handle* h, const char* dir, int nfile, double* energies8) try {
  SP_UNAVAILABLE("write_DEM");
  CHECK_H(h);
  const int n = h->n;
  std::vector<double> t(30 * (size_t)n), hf(3 * (size_t)n);
  int rc = lbmdem_download_grain_table(h, t.data());
  if (rc != LBMDEM_OK) return rc;
  rc = lbmdem_download_fhf(h, hf.data());
  if (rc != LBMDEM_OK) return rc;
  const lbmdem_config& c = h->cfg;
  const lbmdem_physics& p = c.phys;
  char path[4096];
  snprintf(path, sizeof path, "%s/DEM%.6i.dat", (dir && *dir) ? dir : ".", nfile);
  FILE* fp = fopen(path, "w");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open '%s' for writing", path);
  auto G = [&](int i, int col) { return t[(size_t)i * 30 + col]; };
  double xfront = G(0, 0) + G(0, 9), height = G(0, 1) + G(0, 9), xgrainmax = G(0, 0);
  double energie_x = 0., energie_y = 0., energie_teta = 0., energy_p = 0., SE = 0., IFR = 0., zmean = 0;
  double WF = 0., INCE = 0., TSLIP = 0., TRW = 0.;
  double N[6] = {0, 0, 0, 0, 0, 0};
  for (int i = 0; i < n; i++) {
    const double x1 = G(i, 0), x2 = G(i, 1), v1 = G(i, 3), v2 = G(i, 4), v3 = G(i, 5), r = G(i, 9), m = G(i, 10),
                 It = G(i, 12), pp = G(i, 13), ss = G(i, 14);
    const int z = (int)G(i, 28), zz = (int)G(i, 29);
    zmean += z;
    if (z >= 0 && z <= 5) N[z] += 1;
    energie_x += 0.5 * m * v1 * v1;
    energie_y += 0.5 * m * v2 * v2;
    energie_teta += 0.5 * It * v3 * v3;
    energy_p += m * p.G * x2;
    SE += 0.5 * (((pp * pp) / p.kg) + ((ss * ss) / p.kt));
    WF += G(i, 19);
    IFR += G(i, 20);
    TSLIP += G(i, 26);
    TRW += G(i, 27);
    INCE += G(i, 25);
    const double ESE = 0.5 * (((pp * pp) / p.kg) + ((ss * ss) / p.kt));
    if (x1 + r > xgrainmax) xgrainmax = x1 + r;
    if (x2 + r > height) height = x2 + r;
    if (zz > 0 && x1 + r >= xfront) xfront = x1 + r;
    fprintf(fp,
            "%i\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%le\t%i\n",
            i, r, x1, x2, G(i, 2), v1, v2, v3, G(i, 6), G(i, 7), G(i, 8), hf[3 * (size_t)i], hf[3 * (size_t)i + 1],
            hf[3 * (size_t)i + 2], pp, ss, ESE, G(i, 19), G(i, 20), G(i, 25), G(i, 26), G(i, 27), G(i, 18), G(i, 21),
            G(i, 22), G(i, 23), G(i, 24), z);
  }
  fclose(fp);
  const double energie_cin = energie_x + energie_y + energie_teta;
  zmean = zmean / n;
  snprintf(path, sizeof path, "%s/stats.data", (dir && *dir) ? dir : ".");
  fp = fopen(path, "a");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open '%s' for appending", path);
  fprintf(fp, "%le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le %le\n",
          h->nbsteps * c.dt - p.dtt, xfront, xgrainmax, height, zmean, energie_x, energie_y, energie_teta, energie_cin,
          N[0] / n, N[1] / n, N[2] / n, N[3] / n, N[4] / n, N[5] / n, energy_p, SE, WF, IFR, INCE, TSLIP, TRW);
  fclose(fp);
  if (energies8) {
    energies8[0] = energie_cin; energies8[1] = energy_p; energies8[2] = SE; energies8[3] = IFR;
    energies8[4] = WF; energies8[5] = INCE; energies8[6] = TSLIP; energies8[7] = TRW;
  }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_write_forces(lbmdem_handle* h, const char* dir, int nfile) try {
  SP_UNAVAILABLE("write_forces");
  CHECK_H(h);
  const int n = h->n;
  std::vector<double> t(30 * (size_t)n);
  int rc = lbmdem_download_grain_table(h, t.data());
  if (rc != LBMDEM_OK) return rc;
  auto X1 = [&](int i) { return t[(size_t)i * 30 + 0]; };
  auto X2 = [&](int i) { return t[(size_t)i * 30 + 1]; };
  auto R = [&](int i) { return t[(size_t)i * 30 + 9]; };
  auto FM = [&](int i) { return t[(size_t)i * 30 + 18]; };
  char path[4096];
  snprintf(path, sizeof path, "%s/DEM%.6i.ps", (dir && *dir) ? dir : ".", nfile);
  FILE* fp = fopen(path, "w");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open '%s' for writing", path);
  const double margin = 10 * R(0), hrx1 = h->cfg.lx, hry2 = h->cfg.ly;  // main.c:449
  fprintf(fp, "%%!PS-Adobe-3.0 EPSF-3.0 \n");
  fprintf(fp, "%%%%BoundingBox: %f %f %f %f \n", -margin, -margin, hrx1 + margin, hry2 + margin);
  fprintf(fp, "%%%%Creator: lbmdem-hip \n");
  fprintf(fp, "%%%%Title: DEM Grains & Forces \n");
  fprintf(fp, "0.1 setlinewidth 0.0 setgray \n");
  for (int i = 0; i < n; i++)
    fprintf(fp, "newpath %le %le %le 0.0 setlinewidth %.2f setgray 0 360 arc gsave fill grestore\n", X1(i) * 10000,
            X2(i) * 10000, R(i) * 10000, (0.8 - FM(i) / 2));
  // overlapping pairs, dn < -1e-10 (main.c:462-466), found on a uniform grid of cell size 2 r_max: any pair
  // with dn < 0 has its centres closer than that, i.e. in adjacent cells
  double xmin = X1(0), xmax = X1(0), ymin = X2(0), ymax = X2(0), rmax = R(0);
  for (int i = 1; i < n; i++) {
    if (X1(i) < xmin) xmin = X1(i);
    if (X1(i) > xmax) xmax = X1(i);
    if (X2(i) < ymin) ymin = X2(i);
    if (X2(i) > ymax) ymax = X2(i);
    if (R(i) > rmax) rmax = R(i);
  }
  const double cs = 2 * rmax > 0 ? 2 * rmax : 1.0;
  long ncx = (long)((xmax - xmin) / cs) + 1, ncy = (long)((ymax - ymin) / cs) + 1;
  while (ncx * ncy > 4L * n + 64) {  // far-flung grains: coarsen (still correct, cells only get larger)
    if (ncx >= ncy) ncx = (ncx + 1) / 2; else ncy = (ncy + 1) / 2;
  }
  const double csx = (xmax - xmin) / ncx > cs ? (xmax - xmin) / ncx * (1 + 1e-12) : cs;
  const double csy = (ymax - ymin) / ncy > cs ? (ymax - ymin) / ncy * (1 + 1e-12) : cs;
  auto cell = [&](double v, double lo, double c, long nc) {
    long k = (long)((v - lo) / c);
    return k < 0 ? 0 : (k >= nc ? nc - 1 : k);
  };
  std::vector<int> start((size_t)(ncx * ncy) + 1, 0), order(n);
  for (int i = 0; i < n; i++) start[(size_t)(cell(X2(i), ymin, csy, ncy) * ncx + cell(X1(i), xmin, csx, ncx)) + 1]++;
  for (size_t k = 1; k < start.size(); k++) start[k] += start[k - 1];
  {
    std::vector<int> fill(start.begin(), start.end() - 1);
    for (int i = 0; i < n; i++) order[(size_t)fill[(size_t)(cell(X2(i), ymin, csy, ncy) * ncx + cell(X1(i), xmin, csx, ncx))]++] = i;
  }
  std::vector<int> js;
  for (int i = 0; i < n; i++) {
    js.clear();
    const long cx = cell(X1(i), xmin, csx, ncx), cy = cell(X2(i), ymin, csy, ncy);
    for (long yy = cy - 1; yy <= cy + 1; ++yy) {
      if (yy < 0 || yy >= ncy) continue;
      for (long xx = cx - 1; xx <= cx + 1; ++xx) {
        if (xx < 0 || xx >= ncx) continue;
        for (int k = start[(size_t)(yy * ncx + xx)]; k < start[(size_t)(yy * ncx + xx) + 1]; ++k) {
          const int j = order[(size_t)k];
          if (j == i) continue;
          const double dn = (sqrt((X1(i) - X1(j)) * (X1(i) - X1(j)) + (X2(i) - X2(j)) * (X2(i) - X2(j)))) - R(i) - R(j);
          if (dn < -1e-10) js.push_back(j);
        }
      }
    }
    for (size_t a = 1; a < js.size(); ++a) {  // ascending j: the reference's inner loop order
      const int v = js[a];
      size_t b = a;
      while (b > 0 && js[b - 1] > v) { js[b] = js[b - 1]; --b; }
      js[b] = v;
    }
    for (int j : js) {
      fprintf(fp, "%le setlinewidth \n 0.0 setgray \n", 1.);
      fprintf(fp, "1 setlinecap \n newpath \n");
      fprintf(fp, "%le %le moveto \n %le %le lineto\n", X1(i) * 10000, X2(i) * 10000, X1(j) * 10000, X2(j) * 10000);
      fprintf(fp, "stroke \n");
    }
  }
  fclose(fp);
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// ---- checkpoint / restart ------------------------------------------------------------------------

namespace {
struct CkptHeader {
  char magic[8];       // "LBMDEMC4"
  double lid6;         // lbmdem_set_lid
  int layout;          // device layout of the populations in the file: 1 = 16-node tiles f[x][y/16][q][y%16]
  int force_mode, diag_always, has_carry;
  double carry[3];     // pft, pff, pf of the order-dependent contact diagnostics (main.c:130-131), when has_carry
  lbmdem_config cfg;   // incl. the wall positions VerletWall may have moved
  long nbsteps;
  int verlet_ok, nnbr; // symmetric list length
  long plane;          // sanity: nxl * sy of the writer
};
}  // namespace
static int dist_enable_caps(lbmdem_handle* h, int M, long cap_g, long cap_t, long cap_l);
namespace {
struct CkptDist {       // follows the lattice when the writer had its grains distributed over strips
  char magic[8];        // "LBMDIST1"
  int margin, cap_g, cap_t, cap_l, poison, pad;
};
constexpr int CKPT_LAYOUT = 1;
static bool wr(FILE* fp, const void* p, size_t n) { return fwrite(p, 1, n, fp) == n; }
static bool rd(FILE* fp, void* p, size_t n) { return fread(p, 1, n, fp) == n; }
}  // namespace

int lbmdem_checkpoint_save(lbmdem_handle* h, const char* path) try {
  SP_UNAVAILABLE("checkpointing");
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!path) return fail(LBMDEM_EINVAL, "null path");
  if (h->obst_pending) return fail(LBMDEM_EINVAL, "checkpoint between obst_construction and collide_stream");
  // (a handle with distributed grains writes ITS strip, the grains as it holds them, its ownership masks and message
  // capacities: one file per rank; the carries must have been agreed over the ranks first, lbmdem_comm_sync_carries)
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int n = h->n;
  std::vector<int> off(n + 1, 0);
  if (h->verlet_ok) HIP_TRY(hipMemcpy(off.data(), h->V.offsets, sizeof(int) * (n + 1), hipMemcpyDeviceToHost));
  CkptHeader H;
  memset(&H, 0, sizeof H);
  memcpy(H.magic, "LBMDEMC4", 8);
  H.lid6 = h->L.lid6;
  H.layout = CKPT_LAYOUT; H.force_mode = h->force_mode; H.diag_always = h->diag_always ? 1 : 0;
  H.has_carry = 1;
  if (!h->dist && h->carry_from < h->substep_seq) {
    launch_carry_resolve(h->ct, h->carry_from, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  HIP_TRY(hipMemcpy(H.carry, h->ct.carry, sizeof H.carry, hipMemcpyDeviceToHost));
  H.cfg = h->cfg; H.nbsteps = h->nbsteps; H.verlet_ok = h->verlet_ok ? 1 : 0; H.nnbr = off[n]; H.plane = h->L.plane;
  FILE* fp = fopen(path, "wb");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open '%s' for writing", path);
  bool ok = wr(fp, &H, sizeof H);
  auto dump = [&](const void* dev, size_t bytes) {
    if (!ok || bytes == 0) return;
    std::vector<char> buf(bytes);
    if (hipMemcpy(buf.data(), dev, bytes, hipMemcpyDeviceToHost) != hipSuccess) { ok = false; return; }
    ok = wr(fp, buf.data(), bytes);
  };
  dump(h->r, sizeof(double) * n);
  dump(h->kin[h->kcur].x1, sizeof(double) * 9 * n);
  dump(h->fhf, sizeof(double) * 3 * n);
  dump(h->gp, sizeof(double) * n);
  dump(h->V.offsets, sizeof(int) * (n + 1));
  dump(h->V.nbr, sizeof(int) * (size_t)H.nnbr);
  dump(h->V.wallflags, n);
  dump(h->obst[h->ocur], sizeof(int) * (size_t)h->L.plane);
  for (int q = 0; q < 9 && ok; ++q)  // the lattice (device layout) in nine chunks: bounded host staging
    dump(h->f[h->fcur] + (size_t)q * h->L.plane, sizeof(double) * (size_t)h->L.plane);
  if (h->dist && ok) {   // optional trailing section
    CkptDist D;
    memset(&D, 0, sizeof D);
    memcpy(D.magic, "LBMDIST1", 8);
    D.margin = h->dist_margin; D.cap_g = h->dd.cap_g; D.cap_t = h->dd.cap_t; D.cap_l = h->dd.cap_l;
    D.poison = h->dist_poison ? 1 : 0;
    ok = wr(fp, &D, sizeof D);
    dump(h->dd.active, n); dump(h->dd.fluidmask, n); dump(h->owner, n);
  }
  ok = (fclose(fp) == 0) && ok;
  if (!ok) return fail(LBMDEM_EHIP, "writing checkpoint '%s' failed", path);
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

int lbmdem_checkpoint_load(const char* path, int device, lbmdem_handle** out) try {
  SP_UNAVAILABLE("checkpointing");
  if (!path || !out) return fail(LBMDEM_EINVAL, "null argument");
  *out = nullptr;
  FILE* fp = fopen(path, "rb");
  if (!fp) return fail(LBMDEM_EINVAL, "cannot open checkpoint '%s'", path);
  CkptHeader H;
  if (!rd(fp, &H, sizeof H) || memcmp(H.magic, "LBMDEMC4", 8) != 0) { fclose(fp); return fail(LBMDEM_EINVAL, "'%s' is not a checkpoint of this library version", path); }
  if (H.layout != CKPT_LAYOUT) { fclose(fp); return fail(LBMDEM_EINVAL, "checkpoint '%s' holds another device layout (%d)", path, H.layout); }
  const int n = H.cfg.nbgrains;
  std::vector<double> r(n), kin(9 * (size_t)n);
  if (!rd(fp, r.data(), sizeof(double) * n) || !rd(fp, kin.data(), sizeof(double) * 9 * n)) { fclose(fp); return fail(LBMDEM_EINVAL, "checkpoint truncated"); }
  lbmdem_config cfg = H.cfg;
  cfg.device = device;
  lbmdem_handle* h = nullptr;
  int rc = lbmdem_create(&cfg, r.data(), kin.data(), kin.data() + n, &h);  // x1, x2 are the first two columns
  if (rc != LBMDEM_OK) { fclose(fp); return rc; }
  bool ok = h->L.plane == H.plane && H.nnbr >= 0 && H.nnbr <= h->V.cap;
  auto fill = [&](void* dev, size_t bytes) {
    if (!ok || bytes == 0) return;
    std::vector<char> buf(bytes);
    ok = rd(fp, buf.data(), bytes) && hipMemcpy(dev, buf.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (ok) ok = hipMemcpy(h->kin[0].x1, kin.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice) == hipSuccess;
  h->kcur = 0;
  fill(h->fhf, sizeof(double) * 3 * n);
  fill(h->gp, sizeof(double) * n);
  fill(h->V.offsets, sizeof(int) * (n + 1));
  fill(h->V.nbr, sizeof(int) * (size_t)H.nnbr);
  fill(h->V.wallflags, n);
  fill(h->obst[0], sizeof(int) * (size_t)h->L.plane);
  h->ocur = 0; h->obst_pending = false;
  for (int q = 0; q < 9 && ok; ++q) fill(h->f[0] + (size_t)q * h->L.plane, sizeof(double) * (size_t)h->L.plane);
  h->fcur = 0;
  if (ok) {   // a strip with distributed grains: masks and message capacities as the writer had them
    CkptDist D;
    if (rd(fp, &D, sizeof D)) {
      ok = memcmp(D.magic, "LBMDIST1", 8) == 0 && dist_enable_caps(h, D.margin, D.cap_g, D.cap_t, D.cap_l) == LBMDEM_OK;
      if (ok) { fill(h->dd.active, n); fill(h->dd.fluidmask, n); fill(h->owner, n); h->dist_poison = D.poison != 0; }
    }
  }
  fclose(fp);
  if (!ok) { lbmdem_destroy(h); return fail(LBMDEM_EINVAL, "checkpoint '%s' is truncated or from a different decomposition", path); }
  h->cfg = cfg;  // wall positions as saved
  h->force_mode = H.force_mode;
  h->L.lid6 = H.lid6;
  h->diag_always = H.diag_always != 0;
  if (H.has_carry) {  // the "previous contact" carries continue across the restart (no records yet: ct.carry stands)
    if (hipMemcpy(h->ct.carry, H.carry, sizeof H.carry, hipMemcpyHostToDevice) != hipSuccess) {
      lbmdem_destroy(h);
      return fail(LBMDEM_EHIP, "checkpoint: carries not restored");
    }
  }
  h->nbsteps = H.nbsteps;
  h->verlet_ok = H.verlet_ok != 0;
  h->verlet_tracks_positions = h->verlet_ok;
  if (h->verlet_ok) {  // the entry -> grain map is derived from the offsets
    launch_fill_own(h->V, n, h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { lbmdem_destroy(h); return fail(LBMDEM_EHIP, "k_fill_own failed"); }
  }
  *out = h;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

long lbmdem_nbsteps(lbmdem_handle* h) { return h ? h->nbsteps : -1; }

int lbmdem_set_nbsteps(lbmdem_handle* h, long n) {
  if (!h || n < 0) return fail(LBMDEM_EINVAL, "bad argument");
  h->nbsteps = n;
  return LBMDEM_OK;
}

int lbmdem_get_config(lbmdem_handle* h, lbmdem_config* out) {
  if (!h || !out) return fail(LBMDEM_EINVAL, "null argument");
  *out = h->cfg;
  return LBMDEM_OK;
}

// ---- streams, timing, multi-GPU plumbing -------------------------------------------------------

int lbmdem_set_stream(lbmdem_handle* h, void* hip_stream) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stream = (hipStream_t)hip_stream;
  return LBMDEM_OK;
}

int lbmdem_use_own_stream(lbmdem_handle* h) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->stream = h->own_stream;
  return LBMDEM_OK;
}

int lbmdem_sync(lbmdem_handle* h) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int ovf = 0;
  HIP_TRY(hipMemcpy(&ovf, h->V.overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (ovf) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  int ferr = 0;
  HIP_TRY(hipMemcpy(&ferr, h->fs.error, sizeof(int), hipMemcpyDeviceToHost));
  if (ferr) return fail(LBMDEM_EINVAL, "hydrodynamic force of a grain cut by a strip boundary could not be formed (code %d: "
                                       "overlapping reduced discs across the cut, or a message capacity exceeded)", ferr);
  return LBMDEM_OK;
}

int lbmdem_profile_enable(lbmdem_handle* h, int on) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->prof = on != 0;
  h->ev_used = 0;
  return LBMDEM_OK;
}

int lbmdem_profile_read(lbmdem_handle* h, double* mean_ms, long* launches) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  double tot = 0.0;
  for (size_t k = 0; k < h->ev_used; ++k) {
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, h->ev0[k], h->ev1[k]));
    tot += ms;
  }
  if (mean_ms) *mean_ms = h->ev_used ? tot / (double)h->ev_used : 0.0;
  if (launches) *launches = (long)h->ev_used;
  return LBMDEM_OK;
}

long lbmdem_halo_doubles(lbmdem_handle* h) { return h ? 9L * h->cfg.halo * h->L.ly : -1; }

int lbmdem_halo_pack2(lbmdem_handle* h, void* buf_lo, void* buf_hi) {
  SP_UNAVAILABLE("the strip decomposition");
  CHECK_H(h);
  const LatticeView& L = h->L;
  const int H = h->cfg.halo;
  if (!buf_lo && !buf_hi) return LBMDEM_OK;
  if (H < 1) return fail(LBMDEM_EINVAL, "no halo on this handle");
  if (L.xo1 - L.xo0 < H) return fail(LBMDEM_EINVAL, "strip narrower than the halo");
  launch_halo_pack(h->f[h->fcur], L, L.xo0, L.xo1 - H, H, (real*)buf_lo, (real*)buf_hi, h->stream);
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_halo_unpack2(lbmdem_handle* h, const void* buf_lo, const void* buf_hi) {
  SP_UNAVAILABLE("the strip decomposition");
  CHECK_H(h);
  const LatticeView& L = h->L;
  const int H = h->cfg.halo;
  if (!buf_lo && !buf_hi) return LBMDEM_OK;
  if (H < 1) return fail(LBMDEM_EINVAL, "no halo on this handle");
  if ((buf_lo && L.xo0 - H < 0) || (buf_hi && L.xo1 + H > L.nxl)) return fail(LBMDEM_EINVAL, "no halo rows on that side");
  if (!buf_lo && !buf_hi) return LBMDEM_OK;
  launch_halo_unpack(h->f[h->fcur], L, L.xo0 - H, L.xo1, H, (const real*)buf_lo, (const real*)buf_hi, h->stream);
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_halo_pack(lbmdem_handle* h, int side, void* dev_buf) {
  if (!dev_buf || (side != 0 && side != 1)) return fail(LBMDEM_EINVAL, "bad halo_pack arguments");
  return lbmdem_halo_pack2(h, side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr);
}

int lbmdem_halo_unpack(lbmdem_handle* h, int side, const void* dev_buf) {
  if (!dev_buf || (side != 0 && side != 1)) return fail(LBMDEM_EINVAL, "bad halo_unpack arguments");
  return lbmdem_halo_unpack2(h, side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr);
}

// ---- strip decomposition with distributed grains ------------------------------------------------------------

// host arithmetic only (a driver checks its decomposition before it forks one process per GPU)
int lbmdem_dist_margin_for(const lbmdem_config* cfg, double rmax) {
  if (!cfg || !(rmax > 0) || !(cfg->dx > 0)) return -1;
  // an error travels one Verlet-list edge per sub-step: centre distance <= 2 r_max + distVerlet (+ drift)
  const double hop = (2 * rmax + cfg->phys.distVerlet) / cfg->dx + 1.0;
  return (int)ceil(cfg->npDEM * hop + rmax / cfg->dx) + 6;
}

int lbmdem_dist_default_margin(lbmdem_handle* h) {
  if (!h) return -1;
  return lbmdem_dist_margin_for(&h->cfg, h->rmax);
}

// allocation + switches of the distributed-grain mode with given message capacities (lbmdem_dist_enable derives them
// from the packing; a restart takes them from the checkpoint: neighbours must agree on the message sizes)
static int dist_enable_caps(lbmdem_handle* h, int M, long cap_g, long cap_t, long cap_l) {
  if (dist_alloc(h->dd, h->n, (int)cap_g, (int)cap_t, (int)cap_l) != 0) { dist_free(h->dd); return fail(LBMDEM_ENOMEM, "dist_alloc failed"); }
  h->dist = true;
  h->dist_margin = M;
  h->fs.mask = h->dd.fluidmask;
  h->fs.local_list = h->dd.local_list;
  h->fs.local_count = h->dd.counters + 6;
  h->fs.local_cap = h->dd.cap_l;
  return LBMDEM_OK;
}

int lbmdem_dist_enable(lbmdem_handle* h, int margin_rows) try {
  SP_UNAVAILABLE("the strip decomposition with distributed grains");
  CHECK_H(h);
  const lbmdem_config& c = h->cfg;
  if (h->dist) return fail(LBMDEM_EINVAL, "already enabled");
  if (!h->fs.tab) return fail(LBMDEM_EINVAL, "distributed grains need the link-sum table (reductionR < 1, < 2^18 grains)");
  const int M = margin_rows > 0 ? margin_rows : lbmdem_dist_default_margin(h);
  const bool cut_lo = c.x_begin > 0, cut_hi = c.x_end < c.lx;
  if ((cut_lo || cut_hi) && c.x_end - c.x_begin < M)
    return fail(LBMDEM_EINVAL, "strip of %d rows is narrower than the margin of %d rows: a margin grain could belong to a "
                               "rank that is not a neighbour (use fewer strips, or replicated grains)", c.x_end - c.x_begin, M);
  if (h->nbsteps % c.npDEM != 0) return fail(LBMDEM_EINVAL, "enable at a fluid-step boundary");
  // message capacities. Grains per side: 1.5 x the fullest band of (M + a grain) rows in the present packing (the
  // same number on every rank: all ranks see the same positions now); tables: every disc a cut can go through.
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> hx(h->n);
  HIP_TRY(hipMemcpy(hx.data(), h->kin[h->kcur].x1, sizeof(double) * h->n, hipMemcpyDeviceToHost));
  const int bandw = M + 2 * (int)ceil(h->rmax / c.dx) + 4;
  std::vector<int> hist(c.lx + 1, 0);
  for (int i = 0; i < h->n; ++i) {
    long row = (long)floor((hx[i] - c.Mgx) / c.dx);
    if (row < 0) row = 0;
    if (row > c.lx - 1) row = c.lx - 1;
    hist[row]++;
  }
  long win = 0, best = 0;
  for (int x = 0; x < c.lx; ++x) {
    win += hist[x];
    if (x >= bandw) win -= hist[x - bandw];
    if (win > best) best = win;
  }
  long cap_g = best + best / 2 + 256;
  if (cap_g > h->n) cap_g = h->n;
  // grains a cut can go through (link ring included): the fullest band of one largest diameter + 4 rows anywhere in the
  // present packing, x 1.5 -- from the same histogram as cap_g, hence also right when several columns of small grains
  // fit into the band (the former ly / (2 rmin) counted one column)
  long cap_t;
  {
    const int tw = 2 * (int)ceil(h->rmax / c.dx) + 4;
    long w2 = 0, b2 = 0;
    for (int x = 0; x < c.lx; ++x) {
      w2 += hist[x];
      if (x >= tw) w2 -= hist[x - tw];
      if (w2 > b2) b2 = w2;
    }
    cap_t = b2 + b2 / 2 + 32;
    const long one_column = (long)(c.ly / (2 * h->rmin / c.dx)) + 32;
    if (cap_t < one_column) cap_t = one_column;
  }
  if (cap_t > h->n) cap_t = h->n;
  // grains that can reach this rank's rows (+ halo): launch bound of the rasteriser and the force-table kernel
  long cap_l = 0;
  {
    const int reach = (int)ceil(h->rmax / c.dx) + 8;
    for (int x = (c.x_begin - reach > 0 ? c.x_begin - reach : 0); x < c.lx && x < c.x_end + reach; ++x) cap_l += hist[x];
    cap_l = cap_l + cap_l / 2 + 256;
    if (cap_l > h->n) cap_l = h->n;
  }
  return dist_enable_caps(h, M, cap_g, cap_t, cap_l);
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// The "previous contact" carries as agreed over all ranks (strip decomposition: before a checkpoint, see
// lbmdem_comm_sync_carries): they stand until a younger contact is recorded.
int lbmdem_dist_set_carries(lbmdem_handle* h, const double* carry3) {
  CHECK_H(h);
  if (!carry3) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->ct.carry, carry3, sizeof(double) * 3, hipMemcpyHostToDevice));
  h->carry_from = h->substep_seq;
  return LBMDEM_OK;
}

// this rank's youngest record per carry (keys {0,0} = none) and its carry[] as it stands (only meaningful on the rank
// that ran the last table sub-step)
int lbmdem_dist_export_carries(lbmdem_handle* h, long long* carry_keys, double* carry_vals, double* carry_standing) {
  CHECK_H(h);
  if (!carry_keys || !carry_vals || !carry_standing) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(carry_standing, h->ct.carry, sizeof(double) * 3, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemsetAsync(h->ct.best_key, 0, sizeof(long long) * 6, h->stream));
  if (h->carry_from < h->substep_seq) launch_carry_resolve(h->ct, h->carry_from, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(carry_keys, h->ct.best_key, sizeof(long long) * 6, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(carry_vals, h->ct.carry, sizeof(double) * 3, hipMemcpyDeviceToHost));
  // the resolve may have overwritten carry[] with a local record: put the standing values back (the caller decides)
  HIP_TRY(hipMemcpy(h->ct.carry, carry_standing, sizeof(double) * 3, hipMemcpyHostToDevice));
  return LBMDEM_OK;
}

int lbmdem_dist_set_poison(lbmdem_handle* h, int on) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  h->dist_poison = on != 0;
  return LBMDEM_OK;
}

long lbmdem_dist_message_doubles(lbmdem_handle* h, int kind) {
  if (!h || !h->dist) return -1;
  switch (kind) {
    case LBMDEM_MSG_KIN: return 1 + 10L * h->dd.cap_g;
    case LBMDEM_MSG_FHF: return 3L * h->dd.cap_g;
    case LBMDEM_MSG_TABLES: return 1 + (1 + 8L * h->fs.spd) * h->dd.cap_t;
  }
  return -1;
}

int lbmdem_dist_begin_period(lbmdem_handle* h) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  // errors of earlier periods (a truncated message list, overlapping discs across a cut, a clash while merging tables)
  // are flagged on the device; the flag follows every period to pinned host memory and stops the run HERE, at the next
  // period, instead of letting it continue on truncated messages until somebody calls lbmdem_sync
  if (*h->ferr_host)
    return fail(LBMDEM_EINVAL, "strip decomposition: device error flag %d in an earlier fluid step (4: more grains near a cut "
                               "than the message capacity, 8: two ranks produced the same link sum, others: the force of a "
                               "grain on a cut could not be formed)", (int)*h->ferr_host);
  HIP_TRY(hipMemcpyAsync((void*)h->ferr_host, h->fs.error, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  const lbmdem_config& c = h->cfg;
  DistGeom Gm;
  Gm.lo = (double)c.x_begin; Gm.hi = (double)c.x_end; Gm.margin = (double)h->dist_margin; Gm.dx = c.dx; Gm.Mgx = c.Mgx;
  Gm.has_lo = c.x_begin > 0; Gm.has_hi = c.x_end < c.lx; Gm.first = c.x_begin == 0; Gm.last = c.x_end == c.lx;
  Gm.gx0 = h->L.gx0; Gm.nxl = h->L.nxl;
  { int* t = h->dd.counters; h->dd.counters = h->dd.counters_alt; h->dd.counters_alt = t; }   // the set cleared last period
  h->fs.local_count = h->dd.counters + 6;
  launch_dist_classify(h->dd, Gm, h->n, h->kin[h->kcur].x1, h->r, h->rLB, h->owner, h->fs.error, h->stream);
  HIP_TRY(hipGetLastError());
  h->dist_period_open = true;
  return LBMDEM_OK;
}

int lbmdem_dist_pack2(lbmdem_handle* h, int kind, void* buf_lo, void* buf_hi) {
  CHECK_H(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (!buf_lo && !buf_hi) return LBMDEM_OK;
  if (kind == LBMDEM_MSG_KIN) launch_dist_pack_kin(h->dd, h->kin[h->kcur], (real*)buf_lo, (real*)buf_hi, h->stream);
  else if (kind == LBMDEM_MSG_FHF) launch_dist_pack_fhf(h->dd, h->fhf, h->n, (real*)buf_lo, (real*)buf_hi, h->stream);
  else if (kind == LBMDEM_MSG_TABLES) {
    CHECK_NOT_SPLIT(h);
    if (!h->slots_valid) return fail(LBMDEM_EINVAL, "table messages are packed between collide_stream and forces_fluid");
    // both neighbours in one launch (a null buffer skips the side)
    real* const bufs[2] = {(real*)buf_lo, (real*)buf_hi};
    const int* const lists[2] = {h->dd.strad_list[0], h->dd.strad_list[1]};
    const int* const counts[2] = {h->dd.counters + 2, h->dd.counters + 3};
    launch_forces_table_pack(h->f[h->fcur], h->obst[h->ocur], h->L, gview(h), h->fs, lists, counts, h->dd.cap_t, bufs,
                             h->stream);
  } else return fail(LBMDEM_EINVAL, "unknown message kind");
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_dist_unpack2(lbmdem_handle* h, int kind, const void* buf_lo, const void* buf_hi) {
  CHECK_H(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (!buf_lo && !buf_hi) return LBMDEM_OK;
  if (kind == LBMDEM_MSG_KIN)
    launch_dist_unpack_kin(h->dd, h->kin[h->kcur], (const real*)buf_lo, (const real*)buf_hi, h->n, h->fs.error, h->stream);
  else if (kind == LBMDEM_MSG_FHF)
    launch_dist_unpack_fhf(h->dd, h->fhf, h->n, (const real*)buf_lo, (const real*)buf_hi, h->stream);
  else if (kind == LBMDEM_MSG_TABLES)
    launch_dist_merge_tables(h->fs, (const real*)buf_lo, (const real*)buf_hi, h->dd.cap_t, h->stream);
  else return fail(LBMDEM_EINVAL, "unknown message kind");
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

int lbmdem_dist_pack(lbmdem_handle* h, int kind, int side, void* dev_buf) {
  if (!dev_buf || (side != 0 && side != 1)) return fail(LBMDEM_EINVAL, "bad lbmdem_dist_pack arguments");
  return lbmdem_dist_pack2(h, kind, side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr);
}

int lbmdem_dist_unpack(lbmdem_handle* h, int kind, int side, const void* dev_buf) {
  if (!dev_buf || (side != 0 && side != 1)) return fail(LBMDEM_EINVAL, "bad lbmdem_dist_unpack arguments");
  return lbmdem_dist_unpack2(h, kind, side == 0 ? dev_buf : nullptr, side == 1 ? dev_buf : nullptr);
}

// ---- drop-in outputs of a strip decomposition -----------------------------------------------------------------------
// write_DEM's table (main.c:340-438) holds, for every grain, diagnostics of the last sub-step; four of them (fr, ice,
// slip, rw) thread "previous contact" carries through ALL contacts in grain-index order (main.c:130-131), which no
// strip can do alone. So the sub-step that feeds write_DEM (every 4000th) is run by ONE rank -- the root -- on a full
// replica: every rank exports the exact state of the grains it owns (+ the youngest carry records of their contacts),
// the caller merges the exports (disjoint: every grain has exactly one owner), the root imports the merged state,
// rebuilds its Verlet list from it and runs the sub-step for all n grains with the single-domain diagnostic pipeline.
// Its own grains come out as the distributed sub-step would have left them (same arithmetic), so it simply carries on.

// state12: [n][12] = 9 kinematic columns + fhf1..3 of the grains this rank owns, zeros elsewhere; owned: [n] 0/1;
// carry_keys: [3][2], carry_vals: [3] -- the youngest record of each carry among the owned grains' contacts
// ({0, 0} = none since the last table sub-step). Pure host outputs; nothing on the device changes.
int lbmdem_dist_export_owned(lbmdem_handle* h, double* state12, unsigned char* owned, long long* carry_keys,
                             double* carry_vals) try {
  CHECK_H(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (!state12 || !owned || !carry_keys || !carry_vals) return fail(LBMDEM_EINVAL, "null buffer");
  const int n = h->n;
  HIP_TRY(hipMemsetAsync(h->ct.best_key, 0, sizeof(long long) * 6, h->stream));
  if (h->carry_from < h->substep_seq) launch_carry_resolve(h->ct, h->carry_from, h->stream);
  HIP_TRY(hipStreamSynchronize(h->stream));
  std::vector<double> kin(9 * (size_t)n), hf(3 * (size_t)n);
  HIP_TRY(hipMemcpy(kin.data(), h->kin[h->kcur].x1, sizeof(double) * 9 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(hf.data(), h->fhf, sizeof(double) * 3 * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(owned, h->owner, n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(carry_keys, h->ct.best_key, sizeof(long long) * 6, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(carry_vals, h->ct.carry, sizeof(double) * 3, hipMemcpyDeviceToHost));
  for (int i = 0; i < n; ++i) {
    double* o = state12 + (size_t)i * 12;
    if (owned[i]) {
      for (int k = 0; k < 9; ++k) o[k] = kin[(size_t)k * n + i];
      for (int k = 0; k < 3; ++k) o[9 + k] = hf[(size_t)k * n + i];
    } else {
      for (int k = 0; k < 12; ++k) o[k] = 0.0;
    }
  }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// The root's table sub-step: state12_full = the merged exports of all ranks ([n][12]); carry_vals[c] replaces carry c
// where carry_has[c] != 0 (the youngest record over all ranks; otherwise the root's own carry, as of the last table
// sub-step, stands). Replaces lbmdem_dem_substep for this one sub-step on this rank.
int lbmdem_dist_table_substep(lbmdem_handle* h, const double* state12_full, const double* carry_vals,
                              const int* carry_has) try {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (!state12_full || !carry_vals || !carry_has) return fail(LBMDEM_EINVAL, "null buffer");
  const int n = h->n;
  std::vector<double> soa(12 * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 12; ++k) soa[(size_t)k * n + i] = state12_full[(size_t)i * 12 + k];
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->kin[h->kcur].x1, soa.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice));
  HIP_TRY(hipMemcpy(h->fhf, soa.data() + 9 * (size_t)n, sizeof(double) * 3 * n, hipMemcpyHostToDevice));
  for (int c = 0; c < 3; ++c)
    if (carry_has[c]) HIP_TRY(hipMemcpy(h->ct.carry + c, carry_vals + c, sizeof(double), hipMemcpyHostToDevice));
  // a list over ALL grains from their exact positions (this rank's own list was built with whatever the grains it does
  // not integrate held). Every pair in contact is in any valid list, pairs that do not touch contribute nothing, and
  // partners are sorted by index: the sub-step's sums are those of the reference's list.
  int rc = lbmdem_verlet_rebuild(h);
  if (rc != LBMDEM_OK) return rc;
  if (!h->dx_ready) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (diag_extra_alloc(h->dx, h->n, h->V.cap, h->ct.carry) != 0) return fail(LBMDEM_ENOMEM, "diagnostic buffers: hipMalloc failed");
    h->dx_ready = true;
  }
  const int film = (h->nbsteps % h->cfg.phys.stepFilm == 0) ? 1 : 0;
  const DemParams P = dem_params(h);
  launch_dem_substep(h->kin[h->kcur], h->kin[1 - h->kcur], h->r, h->m, h->It, h->fhf, h->V, h->gp, P, film, h->diag,
                     &h->dx, nullptr, nullptr, h->substep_seq, nullptr, h->stream);
  launch_diag_extra(h->dx, h->kin[h->kcur], h->r, h->V, P, film, h->stream);
  h->carry_from = h->substep_seq + 1;
  h->substep_seq++;
  h->diag_valid = true;
  HIP_TRY(hipGetLastError());
  h->kcur = 1 - h->kcur;
  h->nbsteps++;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// write_vtk of a strip decomposition (main.c:237-338): every rank drops its owned columns into zero-initialised
// lattice-sized arrays (fields11 = grain_pressure[cnt], grain_velocity[3 cnt], grain_acceleration[3 cnt],
// fluid_pressure[cnt], fluid_velocity[3 cnt], cnt = lx * ly, each [ly][lx]); the caller merges the ranks' arrays
// (disjoint columns) and one rank writes the five files with lbmdem_write_vtk_fields.
int lbmdem_vtk_place_owned(lbmdem_handle* h, float* fields11) try {
  CHECK_H(h);
  if (!fields11) return fail(LBMDEM_EINVAL, "null buffer");
  const LatticeView& L = h->L;
  const int nx = L.xo1 - L.xo0, x0 = L.gx0 + L.xo0;
  const size_t part = (size_t)nx * L.ly, cnt = (size_t)L.lx * L.ly;
  std::vector<float> loc(11 * part);
  float* lp[5] = {loc.data(), loc.data() + part, loc.data() + 4 * part, loc.data() + 7 * part, loc.data() + 8 * part};
  int rc = lbmdem_download_vtk_fields(h, lp[0], lp[1], lp[2], lp[3], lp[4]);
  if (rc != LBMDEM_OK) return rc;
  float* fp[5] = {fields11, fields11 + cnt, fields11 + 4 * cnt, fields11 + 7 * cnt, fields11 + 8 * cnt};
  const int dims[5] = {1, 3, 3, 1, 3};
  for (int k = 0; k < 5; ++k)
    for (int y = 0; y < L.ly; ++y)
      memcpy(fp[k] + ((size_t)y * L.lx + x0) * dims[k], lp[k] + (size_t)y * nx * dims[k], sizeof(float) * nx * dims[k]);
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

static int write_vtk_file(const char* path, int nx, int ny, const char* name, int dim, const float* data);

int lbmdem_write_vtk_fields(const char* dir, int nfile, int lx, int ly, const float* fields11) {
  if (!fields11 || lx < 2 || ly < 2) return fail(LBMDEM_EINVAL, "bad lbmdem_write_vtk_fields arguments");
  const size_t cnt = (size_t)lx * ly;
  const char* names[5] = {"grain_pressure", "grain_velocity", "grain_acceleration", "fluid_pressure", "fluid_velocity"};
  const int dims[5] = {1, 3, 3, 1, 3};
  const float* data[5] = {fields11, fields11 + cnt, fields11 + 4 * cnt, fields11 + 7 * cnt, fields11 + 8 * cnt};
  for (int k = 0; k < 5; ++k) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s_%.6i.vtk", (dir && *dir) ? dir : ".", names[k], nfile);  // main.c:241-249
    const int rc = write_vtk_file(path, lx, ly, names[k], dims[k], data[k]);
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
}

int lbmdem_fhf_device(lbmdem_handle* h, void** fhf, void** owner_mask) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  if (fhf) *fhf = h->fhf;
  if (owner_mask) *owner_mask = h->owner;
  return LBMDEM_OK;
}

int lbmdem_fhf_export(lbmdem_handle* h, void* dev_buf) {
  SP_UNAVAILABLE("the strip decomposition");
  CHECK_H(h);
  if (!dev_buf) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(hipMemcpyAsync(dev_buf, h->fhf, sizeof(double) * 3 * h->n, hipMemcpyDeviceToDevice, h->stream));
  return LBMDEM_OK;
}

int lbmdem_fhf_import(lbmdem_handle* h, const void* dev_buf) {
  SP_UNAVAILABLE("the strip decomposition");
  CHECK_H(h);
  if (!dev_buf) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(hipMemcpyAsync(h->fhf, dev_buf, sizeof(double) * 3 * h->n, hipMemcpyDeviceToDevice, h->stream));
  return LBMDEM_OK;
}

// ---- RCCL transport for the distributed-grain strips (the C host driver; strips.py does the same over
// torch.distributed) ---------------------------------------------------------------------------------------------
// RCCL is loaded with dlopen when the first communicator is made: processes that never call lbmdem_comm_* (the
// single-GPU driver, Python with torch's own RCCL) do not load a second copy of the library.

}  // extern "C"  (reopened below)
#pragma GCC visibility pop

#include <dlfcn.h>
#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.lib) return LBMDEM_OK;
  // a copy that is already in the process (PyTorch-ROCm ships its own as "librccl.so") is reused: one RCCL per process
  void* lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(LBMDEM_EHIP, "cannot load RCCL: %s", dlerror());
#define RCCL_SYM(field, name)                                                                  \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));                  \
  if (!g_rccl.field) { dlclose(lib); return fail(LBMDEM_EHIP, "RCCL lacks %s", name); }
  RCCL_SYM(GetUniqueId, "ncclGetUniqueId") RCCL_SYM(CommInitRank, "ncclCommInitRank") RCCL_SYM(CommDestroy, "ncclCommDestroy")
  RCCL_SYM(Send, "ncclSend") RCCL_SYM(Recv, "ncclRecv") RCCL_SYM(GroupStart, "ncclGroupStart") RCCL_SYM(GroupEnd, "ncclGroupEnd")
  RCCL_SYM(AllReduce, "ncclAllReduce") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
  g_rccl.lib = lib;
  return LBMDEM_OK;
}
}  // namespace

#define NCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return fail(LBMDEM_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));    \
  } while (0)

// message classes that can be in flight at the same time each have their own side stream
enum { LANE_KIN = 0, LANE_HALO, LANE_TAB, LANE_FHF, LANE_COUNT };

struct lbmdem_comm {
  // one communicator per lane: messages of different lanes are in flight at the same time, and RCCL orders the
  // operations of ONE communicator
  ncclComm_t nccl[LANE_COUNT] = {};
  int rank = 0, world = 1, device = 0;
  hipStream_t side[LANE_COUNT] = {};
  hipEvent_t ready[LANE_COUNT] = {}, done[LANE_COUNT] = {};
  // device buffers for one handle: [kind or halo][side][send/recv]
  lbmdem_handle* bound = nullptr;
  double* buf[4][2][2] = {};
  size_t count[4] = {};   // doubles per message: KIN, FHF, TABLES, halo
  double* scratch = nullptr;
};

#pragma GCC visibility push(default)
extern "C" {

int lbmdem_comm_unique_id(void* id128) {
  if (!id128) return fail(LBMDEM_EINVAL, "null buffer");
  int rc = rccl_load();
  if (rc != LBMDEM_OK) return rc;
  static_assert(sizeof(ncclUniqueId) * LANE_COUNT == LBMDEM_COMM_ID_BYTES, "one ncclUniqueId per lane");
  for (int l = 0; l < LANE_COUNT; ++l) NCCL_TRY(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128) + l));
  return LBMDEM_OK;
}

int lbmdem_comm_destroy(lbmdem_comm* c) {
  if (!c) return LBMDEM_OK;
  (void)hipSetDevice(c->device);
  for (int l = 0; l < LANE_COUNT; ++l) {
    if (c->side[l]) { (void)hipStreamSynchronize(c->side[l]); (void)hipStreamDestroy(c->side[l]); }
    if (c->ready[l]) (void)hipEventDestroy(c->ready[l]);
    if (c->done[l]) (void)hipEventDestroy(c->done[l]);
  }
  for (auto& k : c->buf) for (auto& s : k) for (double*& p : s) if (p) (void)hipFree(p);
  if (c->scratch) (void)hipFree(c->scratch);
  for (int l = 0; l < LANE_COUNT; ++l) if (c->nccl[l]) (void)g_rccl.CommDestroy(c->nccl[l]);
  delete c;
  return LBMDEM_OK;
}

int lbmdem_comm_create(const void* id128, int rank, int world, int device, lbmdem_comm** out) try {
  SP_UNAVAILABLE("the RCCL transport");
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_create arguments");
  *out = nullptr;
  int rc = rccl_load();
  if (rc != LBMDEM_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  lbmdem_comm* c = new lbmdem_comm();
  c->rank = rank; c->world = world; c->device = device;
  for (int l = 0; l < LANE_COUNT; ++l) {   // every rank creates them in the same order
    ncclUniqueId id;
    memcpy(&id, static_cast<const char*>(id128) + l * sizeof id, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl[l], world, id, rank);
    if (r != ncclSuccess) { lbmdem_comm_destroy(c); return fail(LBMDEM_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
  }
  for (int l = 0; l < LANE_COUNT; ++l) {
    if (hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready[l], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[l], hipEventDisableTiming) != hipSuccess) {
      lbmdem_comm_destroy(c);
      return fail(LBMDEM_EHIP, "stream / event creation failed");
    }
  }
  if (hipMalloc((void**)&c->scratch, sizeof(double) * 1024) != hipSuccess) { lbmdem_comm_destroy(c); return fail(LBMDEM_ENOMEM, "hipMalloc"); }
  *out = c;
  return LBMDEM_OK;
} catch (...) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

// transfers of one lane with both neighbours: they depend on what the main stream has enqueued so far, not on
// what it enqueues next; `done` is what the main stream waits for later
static int comm_begin(lbmdem_comm* c, hipStream_t main, int lane, int kind, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  HIP_TRY(hipEventRecord(c->ready[lane], main));
  HIP_TRY(hipStreamWaitEvent(c->side[lane], c->ready[lane], 0));
  NCCL_TRY(g_rccl.GroupStart());
  for (int s = 0; s < 2; ++s) {
    if (!has[s]) continue;
    const int peer = s == 0 ? c->rank - 1 : c->rank + 1;
    NCCL_TRY(g_rccl.Send(c->buf[kind][s][0], c->count[kind], ncclDouble, peer, c->nccl[lane], c->side[lane]));
    NCCL_TRY(g_rccl.Recv(c->buf[kind][s][1], c->count[kind], ncclDouble, peer, c->nccl[lane], c->side[lane]));
  }
  NCCL_TRY(g_rccl.GroupEnd());
  HIP_TRY(hipEventRecord(c->done[lane], c->side[lane]));
  return LBMDEM_OK;
}
// A transfer nothing can overlap with (link-sum tables, forces: the next kernel needs them) simply takes its place in the
// main stream: measured with lbmdem_comm_exchange_probe, the two event hand-overs of the side-stream form cost ~25 us
// more than the transfer itself (~10 us).
static int comm_inline(lbmdem_comm* c, hipStream_t main, int lane, int kind, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  NCCL_TRY(g_rccl.GroupStart());
  for (int s = 0; s < 2; ++s) {
    if (!has[s]) continue;
    const int peer = s == 0 ? c->rank - 1 : c->rank + 1;
    NCCL_TRY(g_rccl.Send(c->buf[kind][s][0], c->count[kind], ncclDouble, peer, c->nccl[lane], main));
    NCCL_TRY(g_rccl.Recv(c->buf[kind][s][1], c->count[kind], ncclDouble, peer, c->nccl[lane], main));
  }
  NCCL_TRY(g_rccl.GroupEnd());
  return LBMDEM_OK;
}
static int comm_end(lbmdem_comm* c, hipStream_t main, int lane, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  HIP_TRY(hipStreamWaitEvent(main, c->done[lane], 0));
  return LBMDEM_OK;
}

static int comm_bind(lbmdem_comm* c, lbmdem_handle* h) {
  if (c->bound == h) return LBMDEM_OK;
  if (c->bound) return fail(LBMDEM_EINVAL, "a communicator serves one handle");
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  c->count[LBMDEM_MSG_KIN] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_KIN);
  c->count[LBMDEM_MSG_FHF] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_FHF);
  c->count[LBMDEM_MSG_TABLES] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_TABLES);
  c->count[3] = (size_t)lbmdem_halo_doubles(h);
  for (int k = 0; k < 4; ++k)
    for (int s = 0; s < 2; ++s)
      for (int d = 0; d < 2; ++d) {
        HIP_TRY(hipMalloc((void**)&c->buf[k][s][d], sizeof(double) * (c->count[k] ? c->count[k] : 1)));
        HIP_TRY(hipMemset(c->buf[k][s][d], 0, sizeof(double) * (c->count[k] ? c->count[k] : 1)));
      }
  HIP_TRY(hipDeviceSynchronize());
  c->bound = h;
  return LBMDEM_OK;
}

#define RC_TRY(expr) do { int rc_ = (expr); if (rc_ != LBMDEM_OK) return rc_; } while (0)

int lbmdem_comm_lbm_step(lbmdem_handle* h, lbmdem_comm* c) {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  RC_TRY(comm_bind(c, h));
  const bool has[2] = {h->cfg.x_begin > 0, h->cfg.x_end < h->cfg.lx};
  hipStream_t main = h->stream;
  RC_TRY(lbmdem_dist_begin_period(h));                       // ownership + message lists from the current positions
  RC_TRY(lbmdem_dist_pack2(h, LBMDEM_MSG_KIN, has[0] ? c->buf[LBMDEM_MSG_KIN][0][0] : nullptr, has[1] ? c->buf[LBMDEM_MSG_KIN][1][0] : nullptr));
  RC_TRY(comm_begin(c, main, LANE_KIN, LBMDEM_MSG_KIN, has));      // margin refresh / migration, under the fluid step
  RC_TRY(lbmdem_obst_construction(h));
  RC_TRY(lbmdem_collide_stream_part(h, LBMDEM_CS_EDGES));
  RC_TRY(lbmdem_halo_pack2(h, has[0] ? c->buf[3][0][0] : nullptr, has[1] ? c->buf[3][1][0] : nullptr));
  RC_TRY(comm_begin(c, main, LANE_HALO, 3, has));
  RC_TRY(lbmdem_collide_stream_part(h, LBMDEM_CS_INTERIOR));  // ... while the bulk of the rows is computed
  RC_TRY(comm_end(c, main, LANE_HALO, has));
  RC_TRY(lbmdem_halo_unpack2(h, has[0] ? c->buf[3][0][1] : nullptr, has[1] ? c->buf[3][1][1] : nullptr));
  RC_TRY(lbmdem_dist_pack2(h, LBMDEM_MSG_TABLES, has[0] ? c->buf[LBMDEM_MSG_TABLES][0][0] : nullptr, has[1] ? c->buf[LBMDEM_MSG_TABLES][1][0] : nullptr));
  RC_TRY(comm_inline(c, main, LANE_TAB, LBMDEM_MSG_TABLES, has));  // link sums of the grains the neighbours own
  RC_TRY(lbmdem_dist_unpack2(h, LBMDEM_MSG_TABLES, has[0] ? c->buf[LBMDEM_MSG_TABLES][0][1] : nullptr, has[1] ? c->buf[LBMDEM_MSG_TABLES][1][1] : nullptr));
  RC_TRY(lbmdem_forces_fluid(h));
  RC_TRY(comm_end(c, main, LANE_KIN, has));
  RC_TRY(lbmdem_dist_unpack2(h, LBMDEM_MSG_KIN, has[0] ? c->buf[LBMDEM_MSG_KIN][0][1] : nullptr, has[1] ? c->buf[LBMDEM_MSG_KIN][1][1] : nullptr));
  RC_TRY(lbmdem_dist_pack2(h, LBMDEM_MSG_FHF, has[0] ? c->buf[LBMDEM_MSG_FHF][0][0] : nullptr, has[1] ? c->buf[LBMDEM_MSG_FHF][1][0] : nullptr));
  RC_TRY(comm_inline(c, main, LANE_FHF, LBMDEM_MSG_FHF, has));    // forces of the margin grains, from their owners
  RC_TRY(lbmdem_dist_unpack2(h, LBMDEM_MSG_FHF, has[0] ? c->buf[LBMDEM_MSG_FHF][0][1] : nullptr, has[1] ? c->buf[LBMDEM_MSG_FHF][1][1] : nullptr));
  return LBMDEM_OK;
}

// Bitwise merge of host buffers whose non-zero bits are DISJOINT across the ranks (every grain has one owner, every
// lattice column one rank): an integer SUM all-reduce then is a bitwise OR (no bit position receives two ones, so no
// carries). Not on the step path (output cadence). In place; every rank gets the merged buffer.
int lbmdem_comm_allreduce_bits(lbmdem_comm* c, void* host_buf, size_t nbytes) {
  if (!c || !host_buf || nbytes == 0) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_allreduce_bits arguments");
  HIP_TRY(hipSetDevice(c->device));
  const size_t words = (nbytes + 7) / 8;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, words * 8));
  hipError_t e = hipMemset(d, 0, words * 8);
  if (e == hipSuccess) e = hipMemcpy(d, host_buf, nbytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(e); }
  const ncclResult_t r = g_rccl.AllReduce(d, d, words, ncclUint64, ncclSum, c->nccl[0], c->side[0]);
  if (r != ncclSuccess) { (void)hipFree(d); return fail(LBMDEM_EHIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r)); }
  e = hipStreamSynchronize(c->side[0]);
  if (e == hipSuccess) e = hipMemcpy(host_buf, d, nbytes, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return LBMDEM_OK;
}

// The sub-step that feeds write_DEM (the one that brings the step counter to a multiple of 4000, main.c:1773) over
// the ranks: exports merged with lbmdem_comm_allreduce_bits, the youngest carry record picked over all ranks, rank 0
// runs lbmdem_dist_table_substep on the full replica, the others their ordinary sub-step.
static int comm_table_substep(lbmdem_handle* h, lbmdem_comm* c) try {
  const int n = h->n, W = c->world;
  std::vector<double> st(12 * (size_t)n), vals(3 * (size_t)W, 0.0);
  std::vector<unsigned char> owned(n);
  std::vector<long long> keys(6 * (size_t)W, 0);
  RC_TRY(lbmdem_dist_export_owned(h, st.data(), owned.data(), keys.data() + 6 * (size_t)c->rank, vals.data() + 3 * (size_t)c->rank));
  RC_TRY(lbmdem_comm_allreduce_bits(c, st.data(), sizeof(double) * st.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, owned.data(), owned.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, keys.data(), sizeof(long long) * keys.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, vals.data(), sizeof(double) * vals.size()));
  for (int i = 0; i < n; ++i)
    if (owned[i] != 1) return fail(LBMDEM_EINVAL, "grain %d has %d owners at sub-step %ld", i, (int)owned[i], h->nbsteps);
  if (c->rank != 0) return lbmdem_dem_substep(h);
  double best_val[3] = {0, 0, 0};
  int has[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) {
    long long b0 = 0, b1 = 0;
    for (int r = 0; r < W; ++r) {
      const long long k0 = keys[6 * (size_t)r + 2 * k], k1 = keys[6 * (size_t)r + 2 * k + 1];
      if (k0 > b0 || (k0 == b0 && k0 != 0 && k1 > b1)) { b0 = k0; b1 = k1; best_val[k] = vals[3 * (size_t)r + k]; has[k] = 1; }
    }
  }
  return lbmdem_dist_table_substep(h, st.data(), best_val, has);
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

// Before a checkpoint: every rank learns the carries as the reference holds them now -- per carry the youngest record
// over all ranks, else what rank 0 has kept since the last table sub-step -- and they stand from here on.
int lbmdem_comm_sync_carries(lbmdem_handle* h, lbmdem_comm* c) try {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  const int W = c->world;
  std::vector<long long> keys(6 * (size_t)W, 0);
  std::vector<double> vals(3 * (size_t)W, 0.0), standing(3 * (size_t)W, 0.0);
  RC_TRY(lbmdem_dist_export_carries(h, keys.data() + 6 * (size_t)c->rank, vals.data() + 3 * (size_t)c->rank,
                                    standing.data() + 3 * (size_t)c->rank));
  if (W > 1) {
    RC_TRY(lbmdem_comm_allreduce_bits(c, keys.data(), sizeof(long long) * keys.size()));
    RC_TRY(lbmdem_comm_allreduce_bits(c, vals.data(), sizeof(double) * vals.size()));
    RC_TRY(lbmdem_comm_allreduce_bits(c, standing.data(), sizeof(double) * standing.size()));
  }
  double out[3];
  for (int k = 0; k < 3; ++k) {
    out[k] = standing[k];   // rank 0's
    long long b0 = 0, b1 = 0;
    for (int r = 0; r < W; ++r) {
      const long long k0 = keys[6 * (size_t)r + 2 * k], k1 = keys[6 * (size_t)r + 2 * k + 1];
      if (k0 > b0 || (k0 == b0 && k0 != 0 && k1 > b1)) { b0 = k0; b1 = k1; out[k] = vals[3 * (size_t)r + k]; }
    }
  }
  return lbmdem_dist_set_carries(h, out);
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

int lbmdem_comm_run(lbmdem_handle* h, lbmdem_comm* c, long n_dem_steps) {
  CHECK_H(h);
  for (long k = 0; k < n_dem_steps; ++k) {
    if (h->nbsteps % h->cfg.npDEM == 0) RC_TRY(lbmdem_comm_lbm_step(h, c));                     // main.c:1710-1718
    if (h->nbsteps % h->cfg.phys.updateVerlet == 0) RC_TRY(lbmdem_verlet_rebuild(h));            // main.c:1721-1724
    if ((h->nbsteps + 1) % 4000 == 0) RC_TRY(comm_table_substep(h, c));                          // feeds write_DEM, main.c:1773
    else RC_TRY(lbmdem_dem_substep(h));                                                          // main.c:1733-1764
  }
  return LBMDEM_OK;
}

// write_vtk (main.c:237-338) of the whole lattice: the strips' columns merged, rank 0 writes the five files.
int lbmdem_comm_write_vtk(lbmdem_handle* h, lbmdem_comm* c, const char* dir, int nfile) try {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  const size_t cnt = (size_t)h->cfg.lx * h->cfg.ly;
  std::vector<float> fields(11 * cnt, 0.f);
  RC_TRY(lbmdem_vtk_place_owned(h, fields.data()));
  if (c->world > 1) RC_TRY(lbmdem_comm_allreduce_bits(c, fields.data(), sizeof(float) * fields.size()));
  if (c->rank != 0) return LBMDEM_OK;
  return lbmdem_write_vtk_fields(dir, nfile, h->cfg.lx, h->cfg.ly, fields.data());
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

int lbmdem_comm_allreduce_sum(lbmdem_comm* c, double* values, int n) {
  if (!c || !values || n < 1 || n > 1024) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_allreduce_sum arguments");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(c->scratch, values, sizeof(double) * n, hipMemcpyHostToDevice));
  NCCL_TRY(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, ncclSum, c->nccl[0], c->side[0]));
  HIP_TRY(hipStreamSynchronize(c->side[0]));
  HIP_TRY(hipMemcpy(values, c->scratch, sizeof(double) * n, hipMemcpyDeviceToHost));
  return LBMDEM_OK;
}

/* A send to and a receive from THIS rank, grouped on a side stream while the caller's stream is busy: the
 * transport of lbmdem_comm_lbm_step exercised with a single rank. */
int lbmdem_comm_selftest(lbmdem_comm* c, int doubles) {
  if (!c || doubles < 1) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_selftest arguments");
  HIP_TRY(hipSetDevice(c->device));
  double *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc((void**)&a, sizeof(double) * doubles));
  HIP_TRY(hipMalloc((void**)&b, sizeof(double) * doubles));
  std::vector<double> ha(doubles), hb(doubles, -1.0);
  for (int k = 0; k < doubles; ++k) ha[k] = 0.5 * k + 1.0;
  hipStream_t main = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&main, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMemcpyAsync(a, ha.data(), sizeof(double) * doubles, hipMemcpyHostToDevice, main);
  if (e == hipSuccess) e = hipEventRecord(c->ready[LANE_HALO], main);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->side[LANE_HALO], c->ready[LANE_HALO], 0);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) {
    r = g_rccl.GroupStart();
    if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[LANE_HALO], c->side[LANE_HALO]);
    if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[LANE_HALO], c->side[LANE_HALO]);
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  if (e == hipSuccess && r == ncclSuccess) e = hipEventRecord(c->done[LANE_HALO], c->side[LANE_HALO]);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamWaitEvent(main, c->done[LANE_HALO], 0);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(hb.data(), b, sizeof(double) * doubles, hipMemcpyDeviceToHost, main);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(main);
  if (main) (void)hipStreamDestroy(main);
  (void)hipFree(a); (void)hipFree(b);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL self send/recv failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  for (int k = 0; k < doubles; ++k) if (hb[k] != ha[k]) return fail(LBMDEM_EHIP, "self send/recv returned wrong data at %d", k);
  if (c->world == 1) return LBMDEM_OK;
  // several ranks: the step's own pattern -- on every lane one grouped exchange with both neighbours, all lanes in
  // flight at once -- with a payload that names sender and lane
  const int left = c->rank - 1, right = c->rank + 1 < c->world ? c->rank + 1 : -1;
  double* d = nullptr;   // [lane][send L, send R, recv L, recv R][doubles]
  HIP_TRY(hipMalloc((void**)&d, sizeof(double) * doubles * 4 * LANE_COUNT));
  std::vector<double> host((size_t)doubles * 4 * LANE_COUNT, -1.0);
  auto value = [&](int rank, int lane, int to_right, int k) { return 1000.0 * rank + 100.0 * lane + 10.0 * to_right + 1e-3 * k; };
  for (int l = 0; l < LANE_COUNT; ++l)
    for (int sd = 0; sd < 2; ++sd)
      for (int k = 0; k < doubles; ++k) host[((size_t)l * 4 + sd) * doubles + k] = value(c->rank, l, sd, k);
  e = hipMemcpy(d, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice);
  r = ncclSuccess;
  for (int l = 0; l < LANE_COUNT && e == hipSuccess && r == ncclSuccess; ++l) {
    double* base = d + (size_t)l * 4 * doubles;
    r = g_rccl.GroupStart();
    if (left >= 0 && r == ncclSuccess) r = g_rccl.Send(base, (size_t)doubles, ncclDouble, left, c->nccl[l], c->side[l]);
    if (left >= 0 && r == ncclSuccess) r = g_rccl.Recv(base + 2 * (size_t)doubles, (size_t)doubles, ncclDouble, left, c->nccl[l], c->side[l]);
    if (right >= 0 && r == ncclSuccess) r = g_rccl.Send(base + (size_t)doubles, (size_t)doubles, ncclDouble, right, c->nccl[l], c->side[l]);
    if (right >= 0 && r == ncclSuccess) r = g_rccl.Recv(base + 3 * (size_t)doubles, (size_t)doubles, ncclDouble, right, c->nccl[l], c->side[l]);
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  for (int l = 0; l < LANE_COUNT; ++l) {
    const hipError_t e2 = hipStreamSynchronize(c->side[l]);
    if (e == hipSuccess) e = e2;
  }
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(host.data(), d, sizeof(double) * host.size(), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL neighbour exchange failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  for (int l = 0; l < LANE_COUNT; ++l)
    for (int k = 0; k < doubles; ++k) {
      // the left neighbour's message "to the right" lands in recv L, the right neighbour's "to the left" in recv R
      if (left >= 0 && host[((size_t)l * 4 + 2) * doubles + k] != value(left, l, 1, k))
        return fail(LBMDEM_EHIP, "lane %d: wrong data from rank %d at %d", l, left, k);
      if (right >= 0 && host[((size_t)l * 4 + 3) * doubles + k] != value(right, l, 0, k))
        return fail(LBMDEM_EHIP, "lane %d: wrong data from rank %d at %d", l, right, k);
    }
  return LBMDEM_OK;
}

/* Measurement helper: what one exchange on the step's critical path costs on this stack. `iters` times
 * { small kernel on a main stream; ready event -> side stream; grouped send + receive of `doubles` values to this rank
 * itself; done event -> main stream; small kernel on the main stream }, timed with events on the main stream, and the
 * same loop without the exchange, and with the send + receive enqueued on the main stream itself (no events).
 * us[0] = mean with the exchange on the side stream, us[1] = without, us[2] = with it in line. */
int lbmdem_comm_exchange_probe(lbmdem_comm* c, int doubles, int iters, double* us) {
  if (!c || doubles < 1 || iters < 1 || !us) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_exchange_probe arguments");
  HIP_TRY(hipSetDevice(c->device));
  double *a = nullptr, *b = nullptr;
  hipStream_t main = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&a, sizeof(double) * doubles);
  if (e == hipSuccess) e = hipMalloc((void**)&b, sizeof(double) * doubles);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&main, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  ncclResult_t r = ncclSuccess;
  const int lane = LANE_TAB;
  for (int with = 2; with >= 0 && e == hipSuccess && r == ncclSuccess; --with) {
    for (int pass = 0; pass < 2 && e == hipSuccess && r == ncclSuccess; ++pass) {   // pass 0 warms up
      const int n = pass == 0 ? 10 : iters;
      if (pass == 1) e = hipEventRecord(e0, main);
      for (int k = 0; k < n && e == hipSuccess && r == ncclSuccess; ++k) {
        e = hipMemsetAsync(a, 0, 8, main);                       // the producer of the message
        if (with == 2) {   // in line: the transfer simply takes its place in the main stream
          if (e == hipSuccess) {
            r = g_rccl.GroupStart();
            if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], main);
            if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], main);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
          }
        } else if (with == 1) {
          if (e == hipSuccess) e = hipEventRecord(c->ready[lane], main);
          if (e == hipSuccess) e = hipStreamWaitEvent(c->side[lane], c->ready[lane], 0);
          if (e == hipSuccess) {
            r = g_rccl.GroupStart();
            if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], c->side[lane]);
            if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], c->side[lane]);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
          }
          if (e == hipSuccess && r == ncclSuccess) e = hipEventRecord(c->done[lane], c->side[lane]);
          if (e == hipSuccess && r == ncclSuccess) e = hipStreamWaitEvent(main, c->done[lane], 0);
        }
        if (e == hipSuccess && r == ncclSuccess) e = hipMemsetAsync(b, 0, 8, main);   // its consumer
      }
      if (pass == 1 && e == hipSuccess && r == ncclSuccess) {
        e = hipEventRecord(e1, main);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        us[with == 1 ? 0 : (with == 0 ? 1 : 2)] = 1e3 * ms / iters;
      } else if (e == hipSuccess) e = hipStreamSynchronize(main);
    }
  }
  if (main) { (void)hipStreamSynchronize(main); (void)hipStreamDestroy(main); }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL self send/recv failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  return LBMDEM_OK;
}

}  // extern "C"
#pragma GCC visibility pop
