// lbmdem_capi.hip -- the C ABI of liblbmdem_hip.so (include/lbmdem_hip.h): handle, memory, the
// step driver with the reference's cadences (renderScene, main.c:1697-1765) and state transfer in
// the reference's host layout. All arithmetic of the hot path lives in lbm_fused.hip, lbm_forces.hip, lbm_obst.hip, lbm_lattice.hip and
// dem_kernels.hip; the host-side arithmetic here is the one-off time-step derivation
// (main.c:1836-1860) and per-grain constants (main.c:624-626,1859), kept bit-identical.


#include "lbmdem_handle.h"
#include <algorithm>

#include <mutex>

static thread_local char g_err[512] = "";

int lbmdem_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

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
}  // namespace
PhaseRange::PhaseRange(const char* name) {
  if (!g_roctx.tried) roctx_init();
  on = g_roctx.push != nullptr;
  if (on) g_roctx.push(name);
}
PhaseRange::~PhaseRange() { if (on) g_roctx.pop(); }

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

static void chain_forget_stream(int device, hipStream_t st);   // (the launches of k_dem_chain are chained across streams: below)

static int next_paint_epoch(lbmdem_handle* h) {
  if (h->mincov) {   // records of an older rasterisation lose against this one's; the 12-bit epoch is wound back rarely
    if (++h->paint_epoch > 0xFFFu) {
      HIP_TRY(hipMemsetAsync(h->mincov, 0, sizeof(unsigned) * (size_t)h->L.plane, h->stream));
      h->paint_epoch = 1;
    }
  }
  return LBMDEM_OK;
}

static int paint_into(lbmdem_handle* h, int* obst) {
  const Kin& K = h->kin[h->kcur];
  if (h->chain_painted && obst == h->obst[1 - h->ocur]) {
    // the run of sub-steps that ended here has painted the discs at these very positions (k_dem_chain, ChainPaint)
    h->chain_painted = false;
    h->obst_reset_rows = 0;
    h->slots_valid = false;
    return LBMDEM_OK;
  }
  drop_chain_paint(h);
  RC_TRY(next_paint_epoch(h));
  const int b = obst == h->obst[1] ? 1 : 0;
  h->chg_state[b] = 0;   // (this rasterisation keeps no account of what it changes)
  // the pair list tells which discs cannot share a node with another one (plain stores instead of atomics). Not with
  // distributed grains: a rank's list is only right for the grains it integrates
  const bool list_ok = h->verlet_ok && h->verlet_tracks_positions && !h->dist && !*h->ovf_host &&
                       *h->moved_host != h->list_generation;   // (a grain has outrun the list: atomics for everybody)
  const int reset_rows = obst == h->obst[1 - h->ocur] ? h->obst_reset_rows : 0;
  const ObstSnap was = h->snap[b][h->snap_cur[b]], now = h->snap[b][1 - h->snap_cur[b]];
  if (h->obst_update && list_ok && h->snap_ok[b] && reset_rows == 0 && was.xc) {
    // the canvas holds this buffer's last picture: only the nodes whose owner changes are written
    launch_obst_update(obst, h->L, h->n, K.x1, K.x2, h->r, h->rLB, K.v1, K.v2, K.v3, h->xc, h->yc, h->r2, h->rbl0, h->pk,
                       h->fs.touched, h->mincov, h->paint_epoch, h->V.offsets, h->V.nbr, was, now, h->V.xreb, h->V.yreb,
                       (real)(0.5 * h->cfg.phys.distVerlet), h->moved_dev, h->list_generation, h->stream);
    h->snap_cur[b] = 1 - h->snap_cur[b];
    h->obst_updates++;
  } else {
    launch_obst_fill_rows(obst, h->L, reset_rows, h->L.nxl, h->stream);
    const bool record = !h->dist && now.xc != nullptr;   // (every grain leaves its disc in the cleared canvas)
    launch_obst_paint(obst, h->L, h->n, K.x1, K.x2, h->r, h->rLB, K.v1, K.v2, K.v3, h->xc, h->yc, h->r2, h->rbl0, h->pk,
                      h->fs.touched, h->dist ? h->dd.fluidmask : nullptr, h->mincov, h->paint_epoch,
                      h->dist ? h->dd.local_list : nullptr, h->dist ? h->dd.counters + 6 : nullptr, h->dist ? h->dd.cap_l : 0,
                      list_ok ? h->V.offsets : nullptr, list_ok ? h->V.nbr : nullptr, record ? now : ObstSnap{nullptr, nullptr, nullptr, nullptr},
                      h->stream);
    if (record) h->snap_cur[b] = 1 - h->snap_cur[b];
    h->snap_ok[b] = record;
    h->obst_repaints++;
  }
  h->obst_reset_rows = 0;
  h->slots_valid = false;  // the grain geometry the table is indexed with has changed
  HIP_TRY(hipGetLastError());
  return LBMDEM_OK;
}

// The grains of the tiles of k_dem_chain (VerletDevice::tile_grains). mode 1: consecutive stretches of 64 grains along a
// Hilbert curve over the positions -- compact patches of the packing, whatever the numbering of the grains; mode 0: by index.
// Only the speed of that kernel depends on it. (A position that is not a number -- a grain some rank does not hold -- sorts
// to the origin.)
static int dem_tiles_compose(lbmdem_handle* h, const double* x1, const double* x2, int mode) {
  const int n = h->n;
  const size_t tiles = ((size_t)n + DEM_TILE - 1) / DEM_TILE;
  std::vector<int> tg(tiles * DEM_TILE, -1);
  std::vector<int> order((size_t)n);
  for (int i = 0; i < n; ++i) order[i] = i;
  if (mode == 1 && n > DEM_TILE) {
    double lo1 = 1e300, hi1 = -1e300, lo2 = 1e300, hi2 = -1e300;
    for (int i = 0; i < n; ++i) {
      if (x1[i] == x1[i]) { lo1 = x1[i] < lo1 ? x1[i] : lo1; hi1 = x1[i] > hi1 ? x1[i] : hi1; }
      if (x2[i] == x2[i]) { lo2 = x2[i] < lo2 ? x2[i] : lo2; hi2 = x2[i] > hi2 ? x2[i] : hi2; }
    }
    const double span = (hi1 - lo1 > hi2 - lo2 ? hi1 - lo1 : hi2 - lo2);
    const double scale = span > 0 ? 65535.0 / span : 0.0;
    std::vector<unsigned long long> key((size_t)n);
    for (int i = 0; i < n; ++i) {
      unsigned x = x1[i] == x1[i] ? (unsigned)((x1[i] - lo1) * scale) : 0u, y = x2[i] == x2[i] ? (unsigned)((x2[i] - lo2) * scale) : 0u;
      if (x > 65535u) x = 65535u;
      if (y > 65535u) y = 65535u;
      unsigned long long d = 0;   // xy -> distance along the Hilbert curve of order 16
      for (unsigned sft = 32768u; sft > 0; sft >>= 1) {
        const unsigned rx = (x & sft) ? 1u : 0u, ry = (y & sft) ? 1u : 0u;
        d += (unsigned long long)sft * sft * ((3u * rx) ^ ry);
        if (ry == 0) {
          if (rx == 1) { x = 65535u - x; y = 65535u - y; }
          const unsigned t = x; x = y; y = t;
        }
      }
      key[i] = d;
    }
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return key[a] < key[b]; });
  }
  for (size_t t = 0; t < tiles; ++t) {
    const size_t b = t * DEM_TILE, e = b + DEM_TILE < (size_t)n ? b + DEM_TILE : (size_t)n;
    std::sort(order.begin() + b, order.begin() + e);   // ascending within a tile: lane order = index order = list order
    for (size_t k = b; k < e; ++k) tg[k] = order[k];
  }
  if (verlet_set_tiles(h->V, n, tg.data()) != 0) return fail(LBMDEM_EHIP, "the tiles of the DEM run could not be set");
  h->dem_tiles_mode = mode;
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
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 2; ++k) {
      CREATE_TRY(hipMalloc((void**)&h->snap[b][k].xc, sizeof(real) * 3 * (size_t)n + n));
      h->snap[b][k].yc = h->snap[b][k].xc + n;
      h->snap[b][k].still2 = h->snap[b][k].xc + 2 * (size_t)n;
      h->snap[b][k].mode = reinterpret_cast<unsigned char*>(h->snap[b][k].xc + 3 * (size_t)n);
      CREATE_TRY(hipMemsetAsync(h->snap[b][k].xc, 0, sizeof(real) * 3 * (size_t)n + n, h->stream));
    }
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
  if (dem_tiles_compose(h, x1, x2, 1) != LBMDEM_OK) { lbmdem_destroy(h); return LBMDEM_EHIP; }
#ifdef LBMDEM_AB
  if (const char* e = getenv("LBMDEM_DEM_CHAIN")) h->chain_max = atoi(e);   // A/B: 0 = one launch per sub-step
#endif
  if (dem_chain_alloc(h->chain, n) != 0) {
    int rc = fail(LBMDEM_ENOMEM, "sub-step hand-over lines: hipMalloc failed");
    lbmdem_destroy(h);
    return rc;
  }
  h->L.gate = h->chain.gate;   // the stop word every kernel of the step path looks at first (LatticeView::gate)
  h->V.gate = h->chain.gate;
  h->ct.gate = h->chain.gate;
  CREATE_TRY(hipHostMalloc((void**)&h->ovf_host, sizeof(int), hipHostMallocDefault));
  *h->ovf_host = 0;
  CREATE_TRY(hipHostMalloc((void**)&h->moved_host, sizeof(int), hipHostMallocDefault));
  *h->moved_host = -1;
  CREATE_TRY(hipHostGetDevicePointer((void**)&h->moved_dev, (void*)h->moved_host, 0));
  if (cfg->x_begin == 0 && cfg->x_end == cfg->lx) {   // (a strip's maps are painted by the stand-alone rasteriser)
    collide_stream_windows(&h->chg_ww, &h->chg_off);
    h->chg_windows = (cfg->ly + h->chg_ww - 1) / h->chg_ww;
    h->chg_words = (L.nxl + 31) / 32 + 4;   // four words of padding: a wave reads 128 rows' worth from its first row on
    for (int b = 0; b < 2; ++b) {
      CREATE_TRY(hipMalloc((void**)&h->chg[b], sizeof(unsigned) * (size_t)h->chg_windows * h->chg_words));
      CREATE_TRY(hipMemset(h->chg[b], 0, sizeof(unsigned) * (size_t)h->chg_windows * h->chg_words));
    }
    CREATE_TRY(hipMalloc((void**)&h->chg_bad, 2 * sizeof(int)));
    CREATE_TRY(hipMemset(h->chg_bad, 0, 2 * sizeof(int)));
  }
  CREATE_TRY(hipHostMalloc((void**)&h->ferr_host, sizeof(int), hipHostMallocDefault));
  *h->ferr_host = 0;
  CREATE_TRY(hipHostGetDevicePointer((void**)&h->ferr_mirror, (void*)h->ferr_host, 0));
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
  chain_forget_stream(h->cfg.device, h->stream);
  for (int b = 0; b < 2; ++b) {
    if (h->f[b]) (void)hipFree(h->f[b]);
    if (h->obst[b]) (void)hipFree(h->obst[b]);
  }
  if (h->gbuf) (void)hipFree(h->gbuf);
  for (int b = 0; b < 2; ++b)
    for (int k = 0; k < 2; ++k)
      if (h->snap[b][k].xc) (void)hipFree(h->snap[b][k].xc);
  if (h->owner) (void)hipFree(h->owner);
  if (h->mincov) (void)hipFree(h->mincov);
  if (h->fs.touched) (void)hipFree(h->fs.touched);
  if (h->fs.tab) (void)hipFree(h->fs.tab);
  if (h->gathered2) (void)hipFree(h->gathered2);
  if (h->fs.queue) (void)hipFree(h->fs.queue);
  if (h->fs.error) (void)hipFree(h->fs.error);
  if (h->dpartial) (void)hipFree(h->dpartial);
  verlet_free(h->V);
  dem_chain_free(h->chain);
  dist_free(h->dd);
  if (h->ovf_host) (void)hipHostFree((void*)h->ovf_host);
  if (h->ferr_host) (void)hipHostFree((void*)h->ferr_host);
  if (h->moved_host) (void)hipHostFree((void*)h->moved_host);
  for (int b = 0; b < 2; ++b) if (h->chg[b]) (void)hipFree(h->chg[b]);
  if (h->chg_bad) (void)hipFree(h->chg_bad);
  if (h->chg_fcheck) (void)hipFree(h->chg_fcheck);
  diag_extra_free(h->dx);
  carry_track_free(h->ct);
  for (hipEvent_t e : h->ev0) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev1) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev2) (void)hipEventDestroy(e);
  if (h->own_stream) (void)hipStreamDestroy(h->own_stream);
  delete h;
  return LBMDEM_OK;
}


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
  h->prof_this = false;
  if (!h->prof) return LBMDEM_OK;
  if ((h->prof_count++ % h->prof_stride) != 0) return LBMDEM_OK;
  h->prof_this = true;
  if (h->ev_used == h->ev0.size()) {
    hipEvent_t a, b, c2;
    HIP_TRY(hipEventCreate(&a));
    HIP_TRY(hipEventCreate(&b));
    HIP_TRY(hipEventCreate(&c2));
    h->ev0.push_back(a); h->ev1.push_back(b); h->ev2.push_back(c2); h->ev2_set.push_back(0);
  }
  h->ev2_set[h->ev_used] = 0;
  hipEvent_t e0 = h->ev0[h->ev_used];
  *e1 = h->ev1[h->ev_used];
  ++h->ev_used;
  HIP_TRY(hipEventRecord(e0, h->stream));
  return LBMDEM_OK;
}


int lbmdem_collide_stream(lbmdem_handle* h) try {
  CHECK_H(h);
  PhaseRange range_("lbmdem:collide_stream");
  CHECK_NOT_SPLIT(h);
  h->cs_prepared = false;   // (a lbmdem_collide_stream_prepare that no EDGES part followed)
  const int* ob_old = h->obst[h->ocur];
  const int* ob_new = h->obst_pending ? h->obst[1 - h->ocur] : h->obst[h->ocur];
  hipEvent_t e1 = nullptr;
  int rc = prof_begin(h, &e1);
  if (rc != LBMDEM_OK) return rc;
  ObstChange chg{nullptr, 0, 0, 0};
  {
    const int bn = 1 - h->ocur;   // the new map's buffer
    if (h->obst_pending && h->chg_on && h->chg[bn] && h->chg_state[bn] == 2) {
      chg = ObstChange{h->chg[bn], h->chg_words, h->chg_ww, h->chg_off};
      h->chg_used++;
      if (h->chg_verify) launch_change_verify(ob_old, ob_new, h->L, chg, h->chg_windows, h->chg_bad, h->stream);
    }
  }
  const ForceSlots S_launch = slots_for_launch(h);
  launch_collide_stream(h->f[h->fcur], h->f[1 - h->fcur], ob_old, ob_new, h->L, gview(h), S_launch, h->stream, chg);
  if (chg.bits && h->chg_verify) {   // the same launch with both maps read everywhere: the same populations (and the same link sums again)
    const size_t fbytes = sizeof(real) * 9 * (size_t)h->L.plane;
    if (!h->chg_fcheck) HIP_TRY(hipMalloc((void**)&h->chg_fcheck, fbytes));
    HIP_TRY(hipMemcpyAsync(h->chg_fcheck, h->f[1 - h->fcur], fbytes, hipMemcpyDeviceToDevice, h->stream));   // (rows the launch does not write)
    launch_collide_stream(h->f[h->fcur], h->chg_fcheck, ob_old, ob_new, h->L, gview(h), S_launch, h->stream);
    launch_count_differences(h->f[1 - h->fcur], h->chg_fcheck, 9 * (long)h->L.plane, h->chg_bad + 1, h->stream);
  }
  if (e1) HIP_TRY(hipEventRecord(e1, h->stream));
  h->slots_valid = h->fs.tab != nullptr;
  HIP_TRY(hipGetLastError());
  h->fcur = 1 - h->fcur;
  if (h->obst_pending) { h->ocur = 1 - h->ocur; h->obst_pending = false; h->obst_reset_rows = 0; }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// what a split collide_stream does on the handle's stream BEFORE its first kernel: start of the profiling interval, the
// slot table the kernels fill (emptied here if it still holds unconsumed sums)
int lbmdem_collide_stream_prepare(lbmdem_handle* h) {
  hipEvent_t e1 = nullptr;  // the interval of this step's fused kernels ends in INTERIOR
  int rc = prof_begin(h, &e1);
  if (rc != LBMDEM_OK) return rc;
  h->cs_slots = slots_for_launch(h);
  h->cs_prepared = true;
  return LBMDEM_OK;
}

int lbmdem_collide_stream_part(lbmdem_handle* h, int part) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  return lbmdem_collide_stream_part_on(h, part, h->stream);
}

// `st`: the stream the part's kernel goes to. EDGES prepares the launch (slot table emptied if need be, profiling start)
// on the handle's stream and, when st is another stream, expects the caller to have made st wait for that point; the two
// parts write disjoint rows from the same old lattice, so they may run concurrently.
static int collide_stream_part_on_impl(lbmdem_handle* h, int part, hipStream_t st);
int lbmdem_collide_stream_part_on(lbmdem_handle* h, int part, hipStream_t st) try {
  const int rc = collide_stream_part_on_impl(h, part, st);
  if (rc != LBMDEM_OK && h) h->cs_prepared = false;   // a failed part leaves no half-prepared launch behind
  return rc;
} catch (const std::bad_alloc&) {
  if (h) h->cs_prepared = false;
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  if (h) h->cs_prepared = false;
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

static int collide_stream_part_on_impl(lbmdem_handle* h, int part, hipStream_t st) {
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
    if (!h->cs_prepared) {   // (a caller that hands the part to another stream has prepared before its hand-over)
      if (st != h->stream) return fail(LBMDEM_EINVAL, "lbmdem_collide_stream_prepare comes before a part on another stream");
      int rc = lbmdem_collide_stream_prepare(h);
      if (rc != LBMDEM_OK) return rc;
    }
    h->cs_prepared = false;
    launch_collide_stream_edges(h->cs_fin, h->f[1 - h->fcur], h->cs_ob_old, h->cs_ob_new, L, gview(h), h->cs_slots, L.xo0,
                                lo_end, hi_begin, L.xo1, st);
    // the profiled interval of a split launch ends with whichever part ends last
    if (h->prof_this && h->ev_used > 0) { HIP_TRY(hipEventRecord(h->ev2[h->ev_used - 1], st)); h->ev2_set[h->ev_used - 1] = 1; }
    HIP_TRY(hipGetLastError());
    h->fcur = 1 - h->fcur;
    if (h->obst_pending) { h->ocur = 1 - h->ocur; h->obst_pending = false; h->obst_reset_rows = 0; }
    h->cs_interior_pending = true;
    return LBMDEM_OK;
  }
  if (part == LBMDEM_CS_INTERIOR) {
    if (!h->cs_interior_pending) return fail(LBMDEM_EINVAL, "LBMDEM_CS_INTERIOR without a preceding LBMDEM_CS_EDGES");
    if (h->cs_hi_begin > h->cs_lo_end) {
      LatticeView Ls = L;
      Ls.xo0 = h->cs_lo_end; Ls.xo1 = h->cs_hi_begin;
      // the grain records are those of EDGES: nothing moves the grains between the two parts
      launch_collide_stream(h->cs_fin, h->f[h->fcur], h->cs_ob_old, h->cs_ob_new, Ls, gview(h), h->cs_slots, st);
    }
    h->slots_valid = h->cs_slots.tab != nullptr;
    if (h->prof_this && h->ev_used > 0) HIP_TRY(hipEventRecord(h->ev1[h->ev_used - 1], h->stream));
    HIP_TRY(hipGetLastError());
    h->cs_interior_pending = false;
    return LBMDEM_OK;
  }
  return fail(LBMDEM_EINVAL, "part must be LBMDEM_CS_EDGES or LBMDEM_CS_INTERIOR");
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
    // the map the fused kernel has just finished with is the next rasterisation's canvas: reset beside the force kernel
    // (single domain; a strip's is reset by the launch that unpacks the neighbours' messages)
    ObstFillJob fill{nullptr, h->L, 0, 0};
    if (!h->dist && !h->cs_interior_pending && h->obst_reset_rows == 0 && !obst_update_planned(h)) {
      fill.map = h->obst[1 - h->ocur];
      fill.row1 = h->L.nxl;
      h->obst_reset_rows = h->L.nxl;
    }
    {
      // the change bits of the map that is painted next (ObstChange) are cleared beside the gather queue
      const int b = 1 - h->ocur;
      if (h->chg[b] && h->chg_on && !h->dist && !h->cs_interior_pending && h->chg_state[b] != 1) {
        fill.clear = h->chg[b];
        fill.nclear = h->chg_windows * h->chg_words;
        h->chg_state[b] = 1;
      }
    }
    launch_forces_slots(h->f[h->fcur], ob, h->L, gview(h), h->fs, h->fscale12, h->fscale3, h->fhf, h->owner,
                        h->force_mode != 0, fill, h->stream);
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

int lbmdem_fused_work_order(lbmdem_handle* h, int* info12) {
  if (!h || !info12) return fail(LBMDEM_EINVAL, "null argument");
  collide_stream_work_order(h->L, info12);
  return LBMDEM_OK;
}

#ifdef LBMDEM_AB
// experiment builds only: per tile of the last k_dem_chain launch {clocks waiting for the halo, clocks working, poll rounds,
// placed | far items << 8, halo grains, list entries, first clock, last clock} (100 MHz)
int lbmdem_debug_chain_times(lbmdem_handle* h, long long* out, int tiles_cap) {
  CHECK_H(h);
  const int tiles = (h->n + DEM_TILE - 1) / DEM_TILE;
  if (!h->chain.dbg) {
    HIP_TRY(hipMalloc((void**)&h->chain.dbg, sizeof(long long) * (16 * (size_t)tiles + 4)));
    HIP_TRY(hipMemset(h->chain.dbg, 0, sizeof(long long) * (16 * (size_t)tiles + 4)));
    return 0;
  }
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int nt = tiles < tiles_cap ? tiles : tiles_cap;
  HIP_TRY(hipMemcpy(out, h->chain.dbg, sizeof(long long) * 16 * (size_t)nt, hipMemcpyDeviceToHost));
  long long paths[4];
  HIP_TRY(hipMemcpy(paths, h->chain.dbg + 16 * (size_t)tiles, sizeof paths, hipMemcpyDeviceToHost));
  fprintf(stderr, "rasterisation by the runs so far: %lld discs still, %lld ring scan, %lld box scan, %lld with partners near\n", paths[0], paths[1], paths[2], paths[3]);
  return nt;
}
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
  // VerletWall moves the right/top DEM walls: main.c:1555-1561
  lbmdem_config& c = h->cfg;
  if (h->nbsteps * c.dt < c.phys.dtt) {
    c.Mdx = 1.e-3 * c.lx / 10;
    c.Mhy = (1.e-3 * c.ly / 10);
  } else {
    c.Mdx = 1.e-3 * c.lx;
    c.Mhy = 1.e-3 * c.ly;
  }
  return lbmdem_verlet_build_lists(h);
}

// the lists alone (pair list + the four wall lists against the walls WHERE THEY ARE): what a rebuild outside the
// reference's updateVerlet cadence may do (the table sub-step of a strip decomposition, lbmdem_dist_table_substep)
int lbmdem_verlet_build_lists(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:verlet_rebuild");
  if (*h->ovf_host) return fail(LBMDEM_ENOMEM, "Verlet list overflow at the previous rebuild (more than %ld symmetric entries)", h->V.cap);
  const int e = launch_verlet_rebuild(h->V, h->kin[h->kcur], h->r, dem_params(h), h->stream);
  if (e != 0) return fail(LBMDEM_EHIP, "Verlet rebuild failed: %s", hipGetErrorString((hipError_t)e));
  // a truncated list is flagged on the device; the flag travels to the host behind the rebuild and is looked at
  // by the next sub-step that finds it set, by the next rebuild and by lbmdem_sync (no stall here)
  HIP_TRY(hipMemcpyAsync((void*)h->ovf_host, h->V.overflow, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  h->verlet_ok = true;
  h->verlet_tracks_positions = true;
  h->list_generation++;
  return LBMDEM_OK;
}

int lbmdem_dem_substep(lbmdem_handle* h) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:dem_substep");
  CHECK_NOT_SPLIT(h);
  if (!h->verlet_ok) return fail(LBMDEM_EINVAL, "lbmdem_dem_substep before the first lbmdem_verlet_rebuild");
  if (*h->ovf_host) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  if (h->dist && CHAIN_FAILED(h)) return fail(LBMDEM_EHIP, CHAIN_FAIL_MSG);
  drop_chain_paint(h);
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
    h->dx.gate = h->chain.gate;
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
  // a slice of the next rasterisation's canvas is reset under this sub-step (npDEM slices between two fluid steps)
  ObstFillJob fill{nullptr, h->L, 0, 0};
  if (!h->obst_pending && !h->cs_interior_pending && h->obst_reset_rows < h->L.nxl && !obst_update_planned(h)) {
    const int slice = (h->L.nxl + h->cfg.npDEM - 1) / h->cfg.npDEM;
    fill.map = h->obst[1 - h->ocur];
    fill.row0 = h->obst_reset_rows;
    fill.row1 = fill.row0 + slice < h->L.nxl ? fill.row0 + slice : h->L.nxl;
    h->obst_reset_rows = fill.row1;
  }
  launch_dem_substep(h->kin[h->kcur], h->kin[1 - h->kcur], h->r, h->m, h->It, h->fhf, h->V, h->gp,
                     P, film, want_diag ? h->diag : nullptr, want_diag ? &h->dx : nullptr,
                     h->dist ? h->dd.active : nullptr, track, h->substep_seq, h->dist ? h->owner : nullptr, fill,
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

// Two launches of k_dem_chain must not share the GPU: each needs ALL its tiles resident at once, and two that are each half
// resident wait for each other's slots until their spins run out. Within a process the launches are therefore chained by an
// event per device whenever they come from different streams (two handles stepped side by side); another PROCESS on the
// same GPU is the caller's business (ranks of a decomposition have few integrated tiles each and fit side by side).
static std::mutex g_chain_mutex;
static hipEvent_t g_chain_done[64] = {};
static hipStream_t g_chain_stream[64] = {};
static bool g_chain_any[64] = {};
// The event is recorded lazily, on the stream of the LAST launch at the moment a launch from another stream arrives (an event
// recorded after every launch held the next dispatch back: 4.4 us per coupled step for something one handle never needs);
// the caller holds g_chain_mutex from here until its own launch is enqueued.
static int chain_serialise_begin(lbmdem_handle* h) {
  const int d = h->cfg.device & 63;
  if (g_chain_any[d] && g_chain_stream[d] != h->stream) {
    if (!g_chain_done[d]) HIP_TRY(hipEventCreateWithFlags(&g_chain_done[d], hipEventDisableTiming));
    HIP_TRY(hipEventRecord(g_chain_done[d], g_chain_stream[d]));
    HIP_TRY(hipStreamWaitEvent(h->stream, g_chain_done[d], 0));
  }
  g_chain_stream[d] = h->stream;
  g_chain_any[d] = true;
  return LBMDEM_OK;
}
// a stream that goes away or is replaced (synchronised by the caller): no later launch has to wait for it
static void chain_forget_stream(int device, hipStream_t st) {
  std::lock_guard<std::mutex> lock(g_chain_mutex);
  if (g_chain_any[device & 63] && g_chain_stream[device & 63] == st) g_chain_any[device & 63] = false;
}

long lbmdem_dem_chain_length(lbmdem_handle* h, long remaining, int fluid) {
  if (h->chain_max < 2 || remaining < 2 || h->diag_always || !h->chain.pub) return 0;
  const lbmdem_config& c = h->cfg;
  const long cap = remaining < h->chain_max ? remaining : h->chain_max;
  long k = 0;
  for (; k < cap; ++k) {
    const long t = h->nbsteps + k;
    if (k > 0 && ((fluid && t % c.npDEM == 0) || t % c.phys.updateVerlet == 0)) break;   // main.c:1710, 1721
    if (t % c.phys.stepFilm == 0) break;                                                  // film law, main.c:1342
    if ((t + 1) % 4000 == 0) break;                                                       // feeds write_DEM, main.c:1773
  }
  if (k < 2) return 0;
  if (!h->chain_checked) {   // once per handle: do all tiles of this packing run at the same time on this GPU?
    h->chain_checked = true;
    // (the census needs the whole GPU like the kernel itself: behind the last launch of any other handle's stream, and nobody
    // launches until it has finished -- beside a running launch neither would get all its workgroups in, and whose spins
    // run out first was a matter of clocks)
    std::lock_guard<std::mutex> chain_lock(g_chain_mutex);
    if (chain_serialise_begin(h) == LBMDEM_OK) (void)dem_chain_census(h->chain, dem_chain_tslots(h->n), h->stream);
  }
  return h->chain.capacity >= dem_chain_tslots(h->n) ? k : 0;
}

static ChainSnap chain_snapshot(const lbmdem_handle* h);

// k ordinary sub-steps (lbmdem_dem_chain_length said so) in one launch
int lbmdem_dem_chain(lbmdem_handle* h, long k, int fluid) {
  CHECK_H(h);
  PhaseRange range_("lbmdem:dem_chain");
  CHECK_NOT_SPLIT(h);
  if (!h->verlet_ok) return fail(LBMDEM_EINVAL, "lbmdem_dem_substep before the first lbmdem_verlet_rebuild");
  if (*h->ovf_host) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  if (h->dist && CHAIN_FAILED(h)) return fail(LBMDEM_EHIP, CHAIN_FAIL_MSG);
  if (!h->dist && h->run_logged) h->chain_pending.push_back(chain_snapshot(h));   // (the caller has set log_idx / done: see run_steps)
  drop_chain_paint(h);
  const DemParams P = dem_params(h);
  // A run that ends where the next fluid step begins rasterises the discs itself (the positions are in the tiles' LDS)
  // -- when the canvas is clean already (reset beside the force kernel), the pair list is the one the rasteriser would use,
  // and nothing about the map is pending.
  ChainPaint paint{};
  if (fluid && h->chain_paint && !h->dist && (h->nbsteps + k) % h->cfg.npDEM == 0 && !h->obst_pending &&
      !h->cs_interior_pending && h->verlet_tracks_positions && !*h->ovf_host && h->cfg.x_begin == 0 && h->cfg.x_end == h->cfg.lx) {
    const int b = 1 - h->ocur;
    // in place (the canvas holds the picture its record describes, nobody has touched it) or onto a clean canvas
    const bool inplace = h->obst_update && h->snap_ok[b] && h->obst_reset_rows == 0 && *h->moved_host != h->list_generation;
    // (a grain that has outrun the pair list: the tail's "alone according to the list" is not to be trusted -- the stand-alone
    // rasteriser with atomics for everybody paints this step)
    const bool clean = h->obst_reset_rows == h->L.nxl && *h->moved_host != h->list_generation;
    if (inplace || clean) {
      RC_TRY(next_paint_epoch(h));
      paint = ChainPaint{h->obst[b], h->L, h->rLB, h->xc, h->yc, h->r2, h->rbl0, h->pk, h->fs.touched, h->mincov, h->paint_epoch,
                         inplace ? h->snap[b][h->snap_cur[b]] : ObstSnap{nullptr, nullptr, nullptr, nullptr},
                         h->snap[b][1 - h->snap_cur[b]], h->r, h->V.xreb, h->V.yreb, (real)(0.5 * h->cfg.phys.distVerlet),
                         h->moved_dev, h->list_generation};
      // ... and, in place, against the picture in the other buffer: the rows in which the two maps of the coming fluid step
      // differ (ObstChange; the bits were cleared beside the last force kernels and nobody has painted this buffer since)
      if (inplace && h->chg[b] && h->chg_on && h->chg_state[b] == 1 && h->snap_ok[1 - b]) {
        paint.chg = ObstChange{h->chg[b], h->chg_words, h->chg_ww, h->chg_off};
        paint.other = h->snap[1 - b][h->snap_cur[1 - b]];
        paint.windows = h->chg_windows;
        h->chg_state[b] = 2;
      } else h->chg_state[b] = 0;
      h->snap_cur[b] = 1 - h->snap_cur[b];
      h->snap_ok[b] = true;
      if (inplace) h->obst_updates++; else h->obst_repaints++;
    }
  }
  ObstFillJob fill{nullptr, h->L, 0, 0};   // k slices of the next rasterisation's canvas (lbmdem_dem_substep: one each)
  if (!h->obst_pending && !h->cs_interior_pending && h->obst_reset_rows < h->L.nxl && !obst_update_planned(h)) {
    const long slice = (h->L.nxl + h->cfg.npDEM - 1) / h->cfg.npDEM;
    fill.map = h->obst[1 - h->ocur];
    fill.row0 = h->obst_reset_rows;
    fill.row1 = fill.row0 + slice * k < h->L.nxl ? (int)(fill.row0 + slice * k) : h->L.nxl;
    h->obst_reset_rows = fill.row1;
  }
  std::lock_guard<std::mutex> chain_lock(g_chain_mutex);
  RC_TRY(chain_serialise_begin(h));
  DemChain chain = h->chain;
#ifdef LBMDEM_AB   // lbmdem_debug_chain_giveup: this launch is made to give up half way (tests/test_gpu_dem_chain.py)
  if (h->chain_giveup_at >= 0 && h->chain_launches == h->chain_giveup_at) chain.capacity = -1;
#endif
  launch_dem_chain(h->kin[h->kcur], h->kin[1 - h->kcur], h->r, h->m, h->It, h->fhf, h->V, h->gp, P,
                   h->dist ? h->dd.active : nullptr, &h->ct, h->substep_seq, h->dist ? h->owner : nullptr, fill, chain,
                   (int)k, paint, h->stream);
  if (paint.obst) { h->chain_painted = true; h->chain_paints++; }
  if (h->dist && h->dist_poison) launch_dist_poison(h->dd, h->kin[0], h->kin[1], h->n, h->stream);
  h->substep_seq += k;
  h->chain_launches++; h->chain_substeps += k;
  h->diag_valid = false;
  HIP_TRY(hipGetLastError());
  h->kcur = 1 - h->kcur;
  h->nbsteps += k;
  return LBMDEM_OK;
}

int lbmdem_set_obst_update(lbmdem_handle* h, int on) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  h->obst_update = on != 0;
  return LBMDEM_OK;
}

int lbmdem_set_change_mask(lbmdem_handle* h, int mode) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  if (mode < 0 || mode > 2) return fail(LBMDEM_EINVAL, "lbmdem_set_change_mask: mode 0 (off), 1 (on) or 2 (on, every use verified)");
  h->chg_on = mode != 0;
  h->chg_verify = mode == 2;
  if (!h->chg_on) h->chg_state[0] = h->chg_state[1] = 0;
  return LBMDEM_OK;
}

int lbmdem_change_mask_stats(lbmdem_handle* h, long* used, long* hidden) {
  CHECK_H(h);
  if (used) *used = h->chg_used;
  if (hidden) {
    int bad[2] = {0, 0};
    if (h->chg_bad) {
      HIP_TRY(hipStreamSynchronize(h->stream));
      HIP_TRY(hipMemcpy(bad, h->chg_bad, 2 * sizeof(int), hipMemcpyDeviceToHost));
    }
    *hidden = (long)bad[0] + ((long)bad[1] << 32);
  }
  return LBMDEM_OK;
}

int lbmdem_obst_stats(lbmdem_handle* h, long* updates, long* repaints) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  if (updates) *updates = h->obst_updates;
  if (repaints) *repaints = h->obst_repaints;
  return LBMDEM_OK;
}

int lbmdem_set_dem_chain(lbmdem_handle* h, int max_substeps) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  if (max_substeps < 0) { h->chain_paint = false; return LBMDEM_OK; }   // (A/B: runs as before, without the rasterisation at their end)
  h->chain_max = max_substeps;
  return LBMDEM_OK;
}

// A rank of a strip decomposition integrates the grains of its strip and a margin: with the grains numbered along the
// packing's rows (the bench packing) those are index ranges, and tiles of consecutive indices are either wholly the rank's or
// not at all -- a rank then runs 27 % more tiles than its share. (Patches along a curve would do as well in principle; with
// several ranks on ONE GPU, the tests' stand-in for several GPUs, their launches did not get their workgroups in side by side.)
int lbmdem_dem_tiles_by_index(lbmdem_handle* h) {
  if (h->dem_tiles_mode == 0) return LBMDEM_OK;
  RC_TRY(dem_tiles_compose(h, nullptr, nullptr, 0));
  if (h->verlet_ok) {
    launch_tile_halo(h->V, h->n, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return LBMDEM_OK;
}

int lbmdem_set_dem_tiles(lbmdem_handle* h, int mode) try {
  CHECK_H(h);
  if (mode != 0 && mode != 1) return fail(LBMDEM_EINVAL, "lbmdem_set_dem_tiles: 0 (by index) or 1 (patches of the packing)");
  const int n = h->n;
  std::vector<real> pos(2 * (size_t)n);
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(pos.data(), h->kin[h->kcur].x1, sizeof(real) * n, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemcpy(pos.data() + n, h->kin[h->kcur].x2, sizeof(real) * n, hipMemcpyDeviceToHost));
  std::vector<double> x1(pos.begin(), pos.begin() + n), x2(pos.begin() + n, pos.end());
  RC_TRY(dem_tiles_compose(h, x1.data(), x2.data(), mode));
  if (h->verlet_ok) {   // what the run's kernel needs of the list depends on the tiles
    launch_tile_halo(h->V, n, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

int lbmdem_dem_chain_stats(lbmdem_handle* h, long* launches, long* substeps, int* tile_slots, int* resident) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  if (launches) *launches = h->chain_launches;
  if (substeps) *substeps = h->chain_substeps;
  if (tile_slots) *tile_slots = dem_chain_tslots(h->n);
  if (resident) *resident = h->chain_checked ? h->chain.capacity : -1;
  return LBMDEM_OK;
}

int lbmdem_dem_chain_paints(lbmdem_handle* h, long* paints) {
  if (!h || !paints) return fail(LBMDEM_EINVAL, "null argument");
  *paints = h->chain_paints;
  return LBMDEM_OK;
}

// ---- runs of sub-steps that a failed launch of k_dem_chain cannot end -----------------------------------------------------
// Every launch of k_dem_chain is remembered with the host's state before it (ChainSnap) until the stream has been seen to
// drain without the failure word; the calls of lbmdem_run / lbmdem_run_dem since then are in the run log. A launch that
// gives up raises the handle's stop word: the kernels queued behind it do nothing, so the device still holds the state the
// launch started from. chain_recover puts the host back there, switches the multi-sub-step kernel off for the handle and
// repeats the logged sub-steps one launch each -- the same bits (tests/test_gpu_dem_chain.py forces a give-up mid-run).
constexpr size_t CHAIN_PENDING_CAP = 256;   // unconfirmed launches before a run loop drains the stream itself

static ChainSnap chain_snapshot(const lbmdem_handle* h) {
  ChainSnap s{};
  s.seq = h->substep_seq;
  s.log_idx = (long)h->runlog.size() - 1;
  s.done = 0;   // (set by run_steps)
  s.fcur = h->fcur; s.ocur = h->ocur; s.kcur = h->kcur; s.obst_reset_rows = h->obst_reset_rows;
  for (int b = 0; b < 2; ++b) { s.snap_cur[b] = h->snap_cur[b]; s.snap_ok[b] = h->snap_ok[b]; s.chg_state[b] = h->chg_state[b]; }
  s.list_generation = h->list_generation;
  s.obst_pending = h->obst_pending; s.diag_valid = h->diag_valid; s.slots_clean = h->slots_clean;
  s.last_forces_from_table = h->last_forces_from_table; s.slots_valid = h->slots_valid; s.verlet_ok = h->verlet_ok;
  s.verlet_tracks_positions = h->verlet_tracks_positions; s.chain_painted = h->chain_painted;
  s.substep_seq = h->substep_seq; s.carry_from = h->carry_from;
  s.gathered = h->fs.gathered; s.gathered_next = h->fs.gathered_next;
  s.nbsteps = h->nbsteps; s.Mdx = h->cfg.Mdx; s.Mhy = h->cfg.Mhy;
  return s;
}

static void chain_restore(lbmdem_handle* h, const ChainSnap& s) {
  h->fcur = s.fcur; h->ocur = s.ocur; h->kcur = s.kcur;
  for (int b = 0; b < 2; ++b) { h->snap_cur[b] = s.snap_cur[b]; h->snap_ok[b] = s.snap_ok[b]; h->chg_state[b] = s.chg_state[b]; }
  h->list_generation = s.list_generation;
  h->obst_pending = s.obst_pending; h->diag_valid = s.diag_valid; h->slots_clean = s.slots_clean;
  h->last_forces_from_table = s.last_forces_from_table; h->slots_valid = s.slots_valid; h->verlet_ok = s.verlet_ok;
  h->verlet_tracks_positions = s.verlet_tracks_positions;
  h->substep_seq = s.substep_seq; h->carry_from = s.carry_from;
  h->fs.gathered = s.gathered; h->fs.gathered_next = s.gathered_next;
  h->nbsteps = s.nbsteps; h->cfg.Mdx = s.Mdx; h->cfg.Mhy = s.Mhy;
  // what the failed launch may have written on its way: slices of the next map's canvas, and -- by the tiles that did
  // finish -- their discs, in place. The map the next fluid step paints starts from a clean canvas again.
  const int b = 1 - s.ocur;
  h->snap_ok[b] = false; h->chg_state[b] = 0;
  h->obst_reset_rows = 0; h->chain_painted = false;
}

static int run_steps(lbmdem_handle* h, int fluid, long n, bool logged, bool resumed = false);

// the stream drained; a launch that gave up undone and everything since repeated. `live`: called from the run loop of the
// log's last entry, which goes on by itself from sub-step *rewind of its call (-1: as it was)
static int chain_settle_impl(lbmdem_handle* h, bool live, long* rewind) {
  if (rewind) *rewind = -1;
  if (h->chain_pending.empty()) return LBMDEM_OK;
  HIP_TRY(hipStreamSynchronize(h->stream));
  const int code = *h->chain.err_host;
  if (code == 0) {   // all of them finished: nothing older than the live call has to be remembered
    h->chain_pending.clear();
    if (live) h->runlog.erase(h->runlog.begin(), h->runlog.end() - 1); else h->runlog.clear();
    return LBMDEM_OK;
  }
  size_t i = 0;
  while (i < h->chain_pending.size() && (int)((unsigned long long)h->chain_pending[i].seq & 0x3FFFFFFFull) + 1 != code) ++i;
  if (i == h->chain_pending.size())
    return fail(LBMDEM_EHIP, "a launch of the multi-sub-step DEM kernel gave up and is not among the %zu the handle remembers (code %d)",
                h->chain_pending.size(), code);
  const ChainSnap s = h->chain_pending[i];
  const std::vector<RunLogEntry> log(h->runlog.begin() + s.log_idx, h->runlog.end());
  chain_restore(h, s);
  h->chain_pending.clear();
  // the multi-sub-step kernel stays off for this handle (lbmdem_set_dem_chain switches it back on, with a new census); its
  // lines hold the give-up marks of the tiles
  h->chain_max = 0;
  h->chain_checked = false; h->chain.capacity = 0;
  h->chain_recoveries++;
  HIP_TRY(hipMemsetAsync(h->chain.gate, 0, sizeof(int), h->stream));
  if (h->chain.pub) HIP_TRY(hipMemsetAsync(h->chain.pub, 0, h->chain.pub_bytes ? h->chain.pub_bytes : 128, h->stream));
  HIP_TRY(hipStreamSynchronize(h->stream));
  *h->chain.err_host = 0;
  // replay: the rest of the call the launch belonged to, then every later call -- except the live one, whose loop goes on
  const bool was_logged = h->run_logged;
  const int was_in = h->in_run;
  h->run_logged = false;
  const size_t upto = live ? log.size() - 1 : log.size();
  int rc = LBMDEM_OK;
  // (the state is that of the moment before the launch: the fluid step and the list rebuild that were due at its first
  // sub-step have been done -- `resumed`)
  for (size_t j = 0; j < upto && rc == LBMDEM_OK; ++j) rc = run_steps(h, log[j].fluid, log[j].n - (j == 0 ? s.done : 0), false, j == 0);
  h->run_logged = was_logged; h->in_run = was_in;
  if (live) {
    h->runlog.erase(h->runlog.begin(), h->runlog.end() - 1);
    if (rewind) *rewind = log.size() == 1 ? s.done : -2;   // -2: the live call starts over (the launch belonged to an earlier one)
  } else h->runlog.clear();
  return rc;
}

int lbmdem_chain_settle(lbmdem_handle* h) { return chain_settle_impl(h, false, nullptr); }

// renderScene() x n (main.c:1697-1765), with or without the fluid (`_FLUIDE_`)
// (`resumed`: the state is that of a launch of k_dem_chain about to be made -- what precedes the first sub-step is done)
static int run_steps(lbmdem_handle* h, int fluid, long n, bool logged, bool resumed) {
  struct Scope {
    lbmdem_handle* h; bool was;
    Scope(lbmdem_handle* h_, bool logged_) : h(h_), was(h_->run_logged) { ++h->in_run; h->run_logged = logged_; }
    ~Scope() { --h->in_run; h->run_logged = was; }
  } scope(h, logged);
  if (logged) h->runlog.push_back(RunLogEntry{fluid, n});
  for (long k = 0; k < n;) {
    size_t cap = CHAIN_PENDING_CAP;
#ifdef LBMDEM_AB   // (tests: a small cap makes the run loop itself find the failed launch)
    static const int env_cap = getenv("LBMDEM_CHAIN_CAP") ? atoi(getenv("LBMDEM_CHAIN_CAP")) : 0;
    if (env_cap > 0) cap = (size_t)env_cap;
#endif
    if (logged && h->chain_pending.size() >= cap) {
      long rewind = -1;
      RC_TRY(chain_settle_impl(h, true, &rewind));
      if (rewind >= 0) { k = rewind; resumed = true; continue; }
      if (rewind == -2) { k = 0; continue; }
    }
    int rc = LBMDEM_OK;
    if (!resumed) {
      if (fluid && h->nbsteps % h->cfg.npDEM == 0) rc = lbmdem_lbm_step(h);                           // main.c:1710-1718
      if (rc == LBMDEM_OK && h->nbsteps % h->cfg.phys.updateVerlet == 0) rc = lbmdem_verlet_rebuild(h);  // main.c:1721-1724
      if (rc != LBMDEM_OK) return rc;
    }
    resumed = false;
    const long run = lbmdem_dem_chain_length(h, n - k, fluid);
    if (run) {
      rc = lbmdem_dem_chain(h, run, fluid);
      if (rc == LBMDEM_OK && logged && !h->dist && !h->chain_pending.empty()) h->chain_pending.back().done = k;
      k += run;
    } else { rc = lbmdem_dem_substep(h); ++k; }                                                     // main.c:1733-1764
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
}

int lbmdem_run(lbmdem_handle* h, long n_dem_steps) {
  CHECK_H_RUNNING(h);   // (no settling here: these are the calls that stay asynchronous)
  return run_steps(h, 1, n_dem_steps, !h->dist && h->in_run == 0);
}

int lbmdem_run_dem(lbmdem_handle* h, long n_dem_steps) {
  CHECK_H_RUNNING(h);
  return run_steps(h, 0, n_dem_steps, !h->dist && h->in_run == 0);
}

int lbmdem_dem_chain_recoveries(lbmdem_handle* h, long* count) {
  if (!h || !count) return fail(LBMDEM_EINVAL, "null argument");
  *count = h->chain_recoveries;
  return LBMDEM_OK;
}

#ifdef LBMDEM_AB
// experiment build: the launch of k_dem_chain with this number (counted from 0 over the handle's life; -1: none) gives up
// half way through its sub-steps, as if a partner's workgroup were not resident
int lbmdem_debug_chain_giveup(lbmdem_handle* h, int launch) {
  if (!h) return fail(LBMDEM_EINVAL, "null handle");
  h->chain_giveup_at = launch;
  return LBMDEM_OK;
}
#endif

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
  drop_chain_paint(h);
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
  chain_forget_stream(h->cfg.device, h->stream);
  h->stream = (hipStream_t)hip_stream;
  return LBMDEM_OK;
}

int lbmdem_use_own_stream(lbmdem_handle* h) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  chain_forget_stream(h->cfg.device, h->stream);
  h->stream = h->own_stream;
  return LBMDEM_OK;
}

int lbmdem_sync(lbmdem_handle* h) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  int ovf = 0;
  HIP_TRY(hipMemcpy(&ovf, h->V.overflow, sizeof(int), hipMemcpyDeviceToHost));
  if (ovf) return fail(LBMDEM_ENOMEM, "Verlet list overflow (more than %ld symmetric entries)", h->V.cap);
  if (h->dist && CHAIN_FAILED(h)) return fail(LBMDEM_EHIP, CHAIN_FAIL_MSG);
  int ferr = 0;
  HIP_TRY(hipMemcpy(&ferr, h->fs.error, sizeof(int), hipMemcpyDeviceToHost));
  if (ferr) return fail(LBMDEM_EINVAL, "hydrodynamic force of a grain cut by a strip boundary could not be formed (code %d: "
                                       "overlapping reduced discs across the cut, or a message capacity exceeded)", ferr);
  return LBMDEM_OK;
}

// bytes copied (read + written: 2 x `bytes`) per second by a plain copy kernel on this handle's device and stream, best of `reps`
int lbmdem_measure_copy(lbmdem_handle* h, size_t bytes, int reps, double* gb_per_s) {
  CHECK_H(h);
  if (!gb_per_s || bytes < 16 || reps < 1) return fail(LBMDEM_EINVAL, "bad lbmdem_measure_copy arguments");
  bytes &= ~(size_t)15;
  void *a = nullptr, *b = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&a, bytes);
  if (e == hipSuccess) e = hipMalloc(&b, bytes);
  if (e == hipSuccess) e = hipMemsetAsync(a, 0x11, bytes, h->stream);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float best = 0.f;
  for (int shape = 0; shape < plain_copy_shapes(); ++shape)   // the best of the copy shapes lbm_lattice.hip knows
    for (int k = 0; k <= reps && e == hipSuccess; ++k) {      // (the first pass of each is a warm-up)
      e = hipEventRecord(e0, h->stream);
      launch_plain_copy(a, b, bytes, h->stream, shape);
      if (e == hipSuccess) e = hipEventRecord(e1, h->stream);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (k > 0 && (best == 0.f || ms < best)) best = ms;
    }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  HIP_TRY(e);
  *gb_per_s = best > 0.f ? 2.0 * (double)bytes / (best * 1e-3) / 1e9 : 0.0;
  return LBMDEM_OK;
}

int lbmdem_profile_enable(lbmdem_handle* h, int on) {
  CHECK_H(h);
  HIP_TRY(hipStreamSynchronize(h->stream));
  h->prof = on != 0;
  h->prof_stride = on > 1 ? on : 1;
  h->prof_count = 0;
  h->prof_this = false;
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
    if (h->ev2_set[k]) {   // the edge rows ran on another stream next to the interior rows
      float ms2 = 0.f;
      HIP_TRY(hipEventSynchronize(h->ev2[k]));
      HIP_TRY(hipEventElapsedTime(&ms2, h->ev0[k], h->ev2[k]));
      if (ms2 > ms) ms = ms2;
    }
    tot += ms;
  }
  if (mean_ms) *mean_ms = h->ev_used ? tot / (double)h->ev_used : 0.0;
  if (launches) *launches = (long)h->ev_used;
  return LBMDEM_OK;
}

}  // extern "C"
#pragma GCC visibility pop
