/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Builds the *unmodified* reference translation unit (cb-geo/2d-lbm-dem, src/main.c) into a
 * shared library so that tests can drive it for a bounded number of steps and read its state.
 * The reference source is #included from where it lies under /root/reference at build time
 * (path passed as -DREF_MAIN_C='"..."'); nothing of it is copied into this repository, and the
 * resulting binary goes to oracle/_ref/ (git-ignored).
 *
 * Why a harness at all: the reference fixes lx/ly at compile time (main.c:27-32), fixes the run
 * length with an unconditional `#define duration 1.5` (main.c:47) and keeps all state in
 * file-scope globals, so a bounded, inspectable run needs (a) main renamed and (b) accessors
 * living in the same translation unit.
 *
 * Two ways to initialise:
 *   ref_init()           - this file's own init, calling the reference's functions in the order
 *                          main() does (main.c:1798-1861). Also zeroes grain.mw, which the
 *                          reference leaves uninitialised (main.c:617-636, used at 1512-1513).
 *   ref_run_real_main()  - runs the reference's real main() untouched; its per-iteration time()
 *                          call (main.c:1882) is routed to a hook that longjmps out after N
 *                          renderScene() calls. Used to prove ref_init() == main()'s init.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <time.h>
#include <setjmp.h>
#include <unistd.h>

#define REF_API __attribute__((visibility("default")))

static jmp_buf ref_jmp;
static long ref_hook_calls = 0;
static long ref_hook_limit = -1;
static time_t ref_hook_time(time_t* out);

/* main() calls time() once during init (main.c:1864) and once after every renderScene()
 * (main.c:1882); route both to the hook. */
#define time ref_hook_time
#define main ref_main
/* The stand-alone reference binary gets zero pages from its large mallocs (fresh mmap), which is
 * what makes the never-initialised grain.mw read as 0 there. Inside a long-lived test process
 * the heap is recycled, so give the reference TU zeroed allocations explicitly. */
#define malloc(n) calloc(1, (n))
#include REF_MAIN_C
#undef malloc
#undef main
#undef time

static time_t ref_hook_time(time_t* out) {
  if (out) *out = 0;
  ++ref_hook_calls;
  /* call #1 = init; call #(k+1) = after the k-th renderScene */
  if (ref_hook_limit >= 0 && ref_hook_calls >= ref_hook_limit + 1) longjmp(ref_jmp, 1);
  return 0;
}

REF_API int ref_lx(void) { return lx; }
REF_API int ref_ly(void) { return ly; }
REF_API int ref_nbgrains(void) { return nbgrains; }
REF_API long ref_nbsteps(void) { return nbsteps; }

/* Own init: same calls, same order as main.c:1798-1861 (stats.data header is not written). */
REF_API int ref_init(const char* sample_path) {
  FILE* probe = fopen(sample_path, "r");
  if (!probe) return -1;
  fclose(probe);
  c_squ = 1. / 3.;
  g = read_sample((char*)sample_path);
  for (int i = 0; i < nbgrains; ++i) g[i].mw = 0.;  /* reference leaves mw indeterminate */
  check_sample(nbgrains, g);

  f = malloc(sizeof(real) * lx * ly * Q);
  obst = malloc(sizeof(int) * lx * ly);
  act = malloc(sizeof(int) * lx * ly);
  delta = malloc(sizeof(real) * lx * ly * Q);
  rLB = malloc(sizeof(real) * nbgrains);
  cumul = calloc(nbgrains, sizeof(int));
  neighbours = calloc((size_t)nbgrains * 6, sizeof(int));
  neighbourWallB = calloc(nbgrains, sizeof(int));
  neighbourWallR = calloc(nbgrains, sizeof(int));
  neighbourWallL = calloc(nbgrains, sizeof(int));
  neighbourWallT = calloc(nbgrains, sizeof(int));
  fhf = malloc(sizeof(struct force) * nbgrains);
  fhf1 = calloc(nbgrains, sizeof(real));
  fhf2 = calloc(nbgrains, sizeof(real));
  fhf3 = calloc(nbgrains, sizeof(real));
  if (!f || !obst || !act || !delta || !rLB || !cumul || !neighbours || !fhf1) return -2;
  /* act/delta are never fully initialised by the reference before first use; make the
   * not-yet-written parts deterministic for dumps (values are overwritten before being read). */
  memset(act, 0, sizeof(int) * lx * ly);
  memset(delta, 0, sizeof(real) * lx * ly * Q);

  init_density(lx, ly, f);
  Mgx = 0.;
  Mdx = 1.e-3 * lx / 10;
  Mhy = 1.e-3 * ly / 10;
  Mby = 0.;
  xG = -G * sin(angleG);
  yG = -G * cos(angleG);
  dx = (1. / scale) * (Mdx - Mgx) / (lx - 1);
  real rMin = minimum_grain_radius(nbgrains, g);
  real dtmax = (1 / iterDEM) * pi * rMin * sqrt(pi * rhoS / kg);
  dtLB = dx * dx * (tau - 0.5) / (3 * nu);
  npDEM = (dtLB / dtmax + 1);
  c = dx / dtLB;
  dt = dtLB / npDEM;
  dt2 = dt * dt;
  for (int i = 0; i <= nbgrains - 1; i++) rLB[i] = reductionR * g[i].r / dx;
  init_obst();
  start = 1;
  nbsteps = 0;
  return 0;
}

/* Real main(), stopped after `nsteps` renderScene() calls. Runs in `workdir` because main()
 * creates stats.data in the cwd (main.c:1867-1868). */
REF_API int ref_run_real_main(const char* sample_path, long nsteps, const char* workdir) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof cwd)) return -1;
  if (workdir && chdir(workdir) != 0) return -2;
  ref_hook_calls = 0;
  ref_hook_limit = nsteps;
  char* argv[3] = {(char*)"lbmdem", (char*)sample_path, NULL};
  if (setjmp(ref_jmp) == 0) ref_main(2, argv);
  ref_hook_limit = -1;
  if (chdir(cwd) != 0) return -3;
  return 0;
}

REF_API void ref_steps(long n) {
  for (long k = 0; k < n; ++k) renderScene();
}

/* the same inside `dir`: renderScene writes DEM%06d.dat/.ps, stats.data (every 4000 steps) and the VTK
 * frames (every 8000) into the cwd (main.c:1767-1776) */
REF_API int ref_steps_in_dir(long n, const char* dir) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof cwd)) return -1;
  if (chdir(dir) != 0) return -2;
  for (long k = 0; k < n; ++k) renderScene();
  if (chdir(cwd) != 0) return -3;
  return 0;
}

/* individual phases, for LBM-only vectors */
REF_API void ref_reinit_obst_density(void) { reinit_obst_density(); }
REF_API void ref_obst_construction(void) { obst_construction(); }
REF_API void ref_collision_streaming(void) { collision_streaming(); }
REF_API void ref_forces_fluid(void) { forces_fluid(lx, ly, f, nbgrains, g); }
REF_API void ref_init_verlet(void) { initVerlet(); VerletWall(); }
REF_API void ref_lbm_steps(int n) {
  for (int k = 0; k < n; ++k) {
    reinit_obst_density();
    obst_construction();
    collision_streaming();
    forces_fluid(lx, ly, f, nbgrains, g);
  }
}

REF_API void ref_get_f(double* out) {
  for (size_t k = 0; k < (size_t)lx * ly * Q; ++k) out[k] = ((real*)f)[k];
}
REF_API void ref_set_f(const double* in) {
  for (size_t k = 0; k < (size_t)lx * ly * Q; ++k) ((real*)f)[k] = in[k];
}
REF_API void ref_get_obst(int* out) { memcpy(out, obst, sizeof(int) * lx * ly); }
REF_API void ref_get_act(int* out) { memcpy(out, act, sizeof(int) * lx * ly); }
REF_API void ref_get_delta(double* out) {
  for (size_t k = 0; k < (size_t)lx * ly * Q; ++k) out[k] = ((real*)delta)[k];
}
REF_API void ref_get_fhf(double* out) {
  for (int i = 0; i < nbgrains; ++i) {
    out[3 * i + 0] = fhf1[i];
    out[3 * i + 1] = fhf2[i];
    out[3 * i + 2] = fhf3[i];
  }
}

/* 30 columns per grain, in struct order (main.c:182-197) */
#define REF_GRAIN_COLS 30
REF_API int ref_grain_cols(void) { return REF_GRAIN_COLS; }
REF_API void ref_get_grains(double* out) {
  for (int i = 0; i < nbgrains; ++i) {
    double* o = out + (size_t)i * REF_GRAIN_COLS;
    o[0] = g[i].x1; o[1] = g[i].x2; o[2] = g[i].x3;
    o[3] = g[i].v1; o[4] = g[i].v2; o[5] = g[i].v3;
    o[6] = g[i].a1; o[7] = g[i].a2; o[8] = g[i].a3;
    o[9] = g[i].r; o[10] = g[i].m; o[11] = g[i].mw; o[12] = g[i].It;
    o[13] = g[i].p; o[14] = g[i].s; o[15] = g[i].f1; o[16] = g[i].f2;
    o[17] = g[i].ifm; o[18] = g[i].fm; o[19] = g[i].fr; o[20] = g[i].ifr;
    o[21] = g[i].M11; o[22] = g[i].M12; o[23] = g[i].M21; o[24] = g[i].M22;
    o[25] = g[i].ice; o[26] = g[i].slip; o[27] = g[i].rw;
    o[28] = g[i].z; o[29] = g[i].zz;
  }
}
/* kinematic state only: x1,x2,x3,v1,v2,v3,a1,a2,a3 (9 per grain) */
REF_API void ref_set_kinematics(const double* in) {
  for (int i = 0; i < nbgrains; ++i) {
    const double* p = in + (size_t)i * 9;
    g[i].x1 = p[0]; g[i].x2 = p[1]; g[i].x3 = p[2];
    g[i].v1 = p[3]; g[i].v2 = p[4]; g[i].v3 = p[5];
    g[i].a1 = p[6]; g[i].a2 = p[7]; g[i].a3 = p[8];
  }
}
REF_API void ref_set_nbsteps(long n) { nbsteps = n; }

/* scalars: dx dtLB dt dt2 c npDEM Mgx Mdx Mby Mhy xG yG */
REF_API void ref_get_scalars(double* out) {
  out[0] = dx; out[1] = dtLB; out[2] = dt; out[3] = dt2; out[4] = c; out[5] = npDEM;
  out[6] = Mgx; out[7] = Mdx; out[8] = Mby; out[9] = Mhy; out[10] = xG; out[11] = yG;
}
REF_API void ref_get_rlb(double* out) {
  for (int i = 0; i < nbgrains; ++i) out[i] = rLB[i];
}

/* Verlet lists: cumul[N], neighbours[6N], counts[4] = {B,T,L,R}, wall lists [N] each */
REF_API void ref_get_verlet(int* cumul_out, int* neigh_out, int* counts, int* wb, int* wt,
                            int* wl, int* wr) {
  memcpy(cumul_out, cumul, sizeof(int) * nbgrains);
  memcpy(neigh_out, neighbours, sizeof(int) * nbgrains * 6);
  counts[0] = nNeighWallb; counts[1] = nNeighWallt; counts[2] = nNeighWallL; counts[3] = nNeighWallR;
  memcpy(wb, neighbourWallB, sizeof(int) * nbgrains);
  memcpy(wt, neighbourWallT, sizeof(int) * nbgrains);
  memcpy(wl, neighbourWallL, sizeof(int) * nbgrains);
  memcpy(wr, neighbourWallR, sizeof(int) * nbgrains);
}

/* write_vtk (main.c:237-338) into `dir`; file names use nFile (main.c:239-249) */
REF_API int ref_write_vtk(const char* dir, int file_index) {
  char cwd[4096];
  if (!getcwd(cwd, sizeof cwd)) return -1;
  if (chdir(dir) != 0) return -2;
  nFile = file_index;
  write_vtk(lx, ly, f, nbgrains, g);
  if (chdir(cwd) != 0) return -3;
  return 0;
}

REF_API double ref_total_density(void) {
  real sum = 0;
  for (int x = 0; x < lx; x++)
    for (int y = 0; y < ly; y++)
      for (int q = 0; q < Q; q++) sum = sum + f[x][y][q];
  return sum;
}
