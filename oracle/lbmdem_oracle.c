/*
 * lbmdem_oracle.c -- CPU ORACLE: TEST INFRASTRUCTURE, NOT PRODUCT CODE (see lbmdem_oracle.h).
 *
 * Serial restatement of the 2D LBM-DEM hot path of cb-geo/2d-lbm-dem. Every routine cites the
 * reference lines (src/main.c) whose arithmetic it follows; expression association is kept
 * identical so that, built with -O2 -ffp-contract=off, results are bit-equal to the reference
 * (checked by tests/test_oracle_vs_reference.py and the dumps under tests/golden/).
 *
 * Differences in form (not in results): runtime lattice size, all state in one struct, grains
 * kept as structure-of-arrays, the Verlet list sized on demand instead of a fixed 6N.
 */
#include "lbmdem_oracle.h"

#include <ctype.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define ORA_API __attribute__((visibility("default")))

enum { NQ = 9, HALF = 4 };

/* D2Q9 direction table and weights (main.c:53-54,70-71) */
static const int EX[NQ] = {0, -1, -1, -1, 0, 1, 1, 1, 0};
static const int EY[NQ] = {0, 1, 0, -1, -1, -1, 0, 1, 1};
static const double WQ[NQ] = {4. / 9, 1. / 36, 1. / 9, 1. / 36, 1. / 9, 1. / 36, 1. / 9, 1. / 36, 1. / 9};

#define REF_PI 3.14159265358979 /* main.c:42 (truncated on purpose) */
#define RHO_S 2650              /* main.c:44 */

struct ora_sim {
  int lx, ly, n;
  int threads;
  double scale;
  /* fluid constants (main.c:74-94) */
  double rho_moy, tau, s2, s3, s5, s7, s8, s9, nu, reductionR;
  double lid_u; /* EXTENSION, not reference-pinned: the top plate's velocity `uw_h` of the terms the reference has
                   commented out at main.c:1129-1130 (`//-uw_h/6`, `//+uw_h/6`); 0 = the reference as it runs */
  /* DEM constants (main.c:97-118) */
  double G, angleG, xG, yG, km, kg, kt, ktm, nug, num, nugt, mu, mum, mumb, murf;
  double distVerlet, dtt, iterDEM, freq, amp, t;
  long updateVerlet;
  int stepFilm;
  /* derived (main.c:1836-1860) */
  double dx, dtLB, c, dt, dt2;
  int npDEM;
  double Mgx, Mdx, Mby, Mhy;
  long nbsteps;
  /* sequential diagnostic carries (main.c:130-131) */
  double pf, pft, pff, ic;
  /* lattice */
  double* f;
  double* delta;
  int* obst;
  int* act;
  /* grains, SoA */
  double *x1, *x2, *x3, *v1, *v2, *v3, *a1, *a2, *a3, *r, *m, *mw, *It;
  double *p, *s, *f1, *f2, *ifm, *fm, *fr, *ifr, *M11, *M12, *M21, *M22, *ice, *slip, *rw;
  int *z, *zz;
  double *rLB, *fhf1, *fhf2, *fhf3;
  /* Verlet lists */
  int* cumul;
  int* neighbours;
  int neigh_cap;
  int *wallB, *wallT, *wallL, *wallR;
  int nB, nT, nL, nR;
};

#define FI(s, x, y, q) ((((size_t)(x)) * (s)->ly + (y)) * NQ + (q))
#define NI(s, x, y) (((size_t)(x)) * (s)->ly + (y))

/* ------------------------------------------------------------------ sample reader */

/* main.c:609-639. fscanf("%le %le %le;\n") accepts the three numbers with or without the ';'. */
ORA_API int ora_read_sample(const char* path, int* n_out, double** r_out, double** x1_out,
                            double** x2_out) {
  FILE* fp = fopen(path, "r");
  if (!fp) return -1;
  char line[256];
  if (!fgets(line, sizeof line, fp)) { fclose(fp); return -2; }
  int n = 0;
  if (fscanf(fp, "%d", &n) != 1 || n <= 0) { fclose(fp); return -3; }
  double* r = malloc(sizeof(double) * n);
  double* x1 = malloc(sizeof(double) * n);
  double* x2 = malloc(sizeof(double) * n);
  if (!r || !x1 || !x2) { fclose(fp); free(r); free(x1); free(x2); return -4; }
  const double unit = 1e-3; /* main.c:114 */
  for (int i = 0; i < n; ++i) {
    double v[3];
    for (int k = 0; k < 3; ++k) {
      int ch;
      while ((ch = fgetc(fp)) != EOF && (isspace(ch) || ch == ';')) {}
      if (ch == EOF) { fclose(fp); free(r); free(x1); free(x2); return -5; }
      ungetc(ch, fp);
      if (fscanf(fp, "%le", &v[k]) != 1) { fclose(fp); free(r); free(x1); free(x2); return -6; }
    }
    r[i] = v[0] * unit;
    x1[i] = v[1] * unit;
    x2[i] = v[2] * unit;
  }
  fclose(fp);
  *n_out = n;
  *r_out = r;
  *x1_out = x1;
  *x2_out = x2;
  return 0;
}

ORA_API void ora_free(void* p) { free(p); }

/* ------------------------------------------------------------------ construction */

static double* dalloc(size_t n) { return (double*)calloc(n ? n : 1, sizeof(double)); }
static int* ialloc(size_t n) { return (int*)calloc(n ? n : 1, sizeof(int)); }

static void paint_initial_obstacles(ora_sim* s);

/* main.c:1836-1860: walls, gravity, dx, dtLB, npDEM, c, dt, rLB */
static void derive_run_constants(ora_sim* s) {
  const int lx = s->lx, ly = s->ly, n = s->n;
  s->Mgx = 0.;
  s->Mdx = 1.e-3 * lx / 10;
  s->Mhy = 1.e-3 * ly / 10;
  s->Mby = 0.;
  s->xG = -s->G * sin(s->angleG);
  s->yG = -s->G * cos(s->angleG);
  s->dx = (1. / s->scale) * (s->Mdx - s->Mgx) / (lx - 1);
  double rMin = s->r[0];
  for (int i = 1; i <= n - 1; i++) rMin = fmin(rMin, s->r[i]);
  double dtmax = (1 / s->iterDEM) * REF_PI * rMin * sqrt(REF_PI * RHO_S / s->kg);
  s->dtLB = s->dx * s->dx * (s->tau - 0.5) / (3 * s->nu);
  s->npDEM = (int)(s->dtLB / dtmax + 1);
  s->c = s->dx / s->dtLB;
  s->dt = s->dtLB / s->npDEM;
  s->dt2 = s->dt * s->dt;
  for (int i = 0; i < n; ++i) s->rLB[i] = s->reductionR * s->r[i] / s->dx;
}

ORA_API ora_sim* ora_create(int lx, int ly, double scale, int n, const double* r, const double* x1,
                            const double* x2) {
  if (lx < 3 || ly < 3 || n < 1) return NULL;
  ora_sim* s = (ora_sim*)calloc(1, sizeof *s);
  if (!s) return NULL;
  s->lx = lx; s->ly = ly; s->n = n; s->threads = 0;
  s->scale = scale;
  s->rho_moy = 1000; s->tau = 0.504;
  s->s2 = 1.5; s->s3 = 1.4; s->s5 = 1.5; s->s7 = 1.5; s->s8 = 1.9841; s->s9 = 1.9841;
  s->nu = 1e-6; s->reductionR = 0.85;
  s->G = 9.81; s->angleG = 0.0;
  s->km = 3e+6; s->kg = 1.6e+6; s->kt = 1.0e+6; s->ktm = 2e+6;
  s->nug = 6.4e+1; s->num = 8.7e+1; s->nugt = 5e-1;
  s->mu = .5317; s->mum = .466; s->mumb = .466; s->murf = 0.01;
  s->distVerlet = 5e-4; s->updateVerlet = 100; s->dtt = 0.; s->iterDEM = 100.;
  s->freq = 5; s->amp = 4.e-4; s->t = 0; s->stepFilm = 8000;

  size_t nn = (size_t)lx * ly;
  s->f = dalloc(nn * NQ); s->delta = dalloc(nn * NQ);
  s->obst = ialloc(nn); s->act = ialloc(nn);
  double** dcols[] = {&s->x1, &s->x2, &s->x3, &s->v1, &s->v2, &s->v3, &s->a1, &s->a2, &s->a3,
                      &s->r, &s->m, &s->mw, &s->It, &s->p, &s->s, &s->f1, &s->f2, &s->ifm, &s->fm,
                      &s->fr, &s->ifr, &s->M11, &s->M12, &s->M21, &s->M22, &s->ice, &s->slip,
                      &s->rw, &s->rLB, &s->fhf1, &s->fhf2, &s->fhf3};
  for (size_t k = 0; k < sizeof dcols / sizeof dcols[0]; ++k) *dcols[k] = dalloc(n);
  s->z = ialloc(n); s->zz = ialloc(n);
  s->cumul = ialloc(n);
  s->neigh_cap = 6 * n > 64 ? 6 * n : 64;
  s->neighbours = ialloc(s->neigh_cap);
  s->wallB = ialloc(n); s->wallT = ialloc(n); s->wallL = ialloc(n); s->wallR = ialloc(n);
  if (!s->f || !s->delta || !s->obst || !s->act || !s->fhf3 || !s->wallR) { ora_destroy(s); return NULL; }

  /* grains: main.c:624-635 (mw: see SURVEY hard part 5 -- zero) */
  for (int i = 0; i < n; ++i) {
    s->r[i] = r[i];
    s->m[i] = RHO_S * REF_PI * s->r[i] * s->r[i];
    s->It[i] = s->m[i] * s->r[i] * s->r[i] / 2;
    s->x1[i] = x1[i];
    s->x2[i] = x2[i];
  }
  /* main.c:1834 */
  for (size_t k = 0; k < nn; ++k)
    for (int q = 0; q < NQ; ++q) s->f[k * NQ + q] = WQ[q];
  derive_run_constants(s);
  paint_initial_obstacles(s);
  return s;
}

/* test-only: replace every physics constant (same order as lbmdem_physics in include/lbmdem_hip.h:
 * rho_moy tau s2 s3 s5 s7 s8 s9 nu reductionR G angleG km kg kt ktm nug num nugt mu mum mumb murf distVerlet
 * dtt iterDEM freq amp t, then updateVerlet, stepFilm) right after ora_create and re-derive the run */
ORA_API void ora_set_physics(ora_sim* s, const double* p, int updateVerlet, int stepFilm) {
  s->rho_moy = p[0]; s->tau = p[1]; s->s2 = p[2]; s->s3 = p[3]; s->s5 = p[4]; s->s7 = p[5]; s->s8 = p[6];
  s->s9 = p[7]; s->nu = p[8]; s->reductionR = p[9]; s->G = p[10]; s->angleG = p[11]; s->km = p[12];
  s->kg = p[13]; s->kt = p[14]; s->ktm = p[15]; s->nug = p[16]; s->num = p[17]; s->nugt = p[18];
  s->mu = p[19]; s->mum = p[20]; s->mumb = p[21]; s->murf = p[22]; s->distVerlet = p[23]; s->dtt = p[24];
  s->iterDEM = p[25]; s->freq = p[26]; s->amp = p[27]; s->t = p[28];
  s->updateVerlet = updateVerlet; s->stepFilm = stepFilm;
  derive_run_constants(s);
  paint_initial_obstacles(s);
}

/* test-only: another reduced-radius factor (the reference's is the global reductionR = 0.85, main.c:94);
 * recomputes rLB (main.c:1858-1860) and repaints the initial obstacle map (main.c:1861) */
/* EXTENSION: lid velocity of the top plate in lattice units (see lid_u above) */
ORA_API void ora_set_lid(ora_sim* s, double uw_h) { s->lid_u = uw_h; }

ORA_API void ora_set_reduction(ora_sim* s, double reductionR) {
  s->reductionR = reductionR;
  for (int i = 0; i < s->n; ++i) s->rLB[i] = s->reductionR * s->r[i] / s->dx;
  paint_initial_obstacles(s);
}

ORA_API void ora_destroy(ora_sim* s) {
  if (!s) return;
  free(s->f); free(s->delta); free(s->obst); free(s->act);
  free(s->x1); free(s->x2); free(s->x3); free(s->v1); free(s->v2); free(s->v3);
  free(s->a1); free(s->a2); free(s->a3); free(s->r); free(s->m); free(s->mw); free(s->It);
  free(s->p); free(s->s); free(s->f1); free(s->f2); free(s->ifm); free(s->fm); free(s->fr);
  free(s->ifr); free(s->M11); free(s->M12); free(s->M21); free(s->M22); free(s->ice);
  free(s->slip); free(s->rw); free(s->z); free(s->zz); free(s->rLB);
  free(s->fhf1); free(s->fhf2); free(s->fhf3);
  free(s->cumul); free(s->neighbours); free(s->wallB); free(s->wallT); free(s->wallL); free(s->wallR);
  free(s);
}

/* bounding box of grain i on the lattice: main.c:1009-1023 (and 686-700, 1296-1303) */
typedef struct { double xc, yc, r2, R2; int xi, xf, yi, yf; } grain_box;

static grain_box grain_bbox(const ora_sim* s, int i) {
  grain_box b;
  b.xc = (s->x1[i] - s->Mgx) / s->dx;
  b.yc = (s->x2[i] - s->Mby) / s->dx;
  b.r2 = s->rLB[i] * s->rLB[i];
  double rbl0 = s->r[i] / s->dx;
  b.R2 = rbl0 * rbl0;
  b.xi = (int)(b.xc - rbl0);
  b.xf = (int)(b.xc + rbl0);
  if (b.xi < 1) b.xi = 1;
  if (b.xf >= s->lx - 1) b.xf = s->lx - 2;
  b.yi = (int)(b.yc - rbl0);
  b.yf = (int)(b.yc + rbl0);
  if (b.yi < 1) b.yi = 1;
  if (b.yf >= s->ly - 1) b.yf = s->ly - 2;
  return b;
}

/* main.c:663-711 */
static void paint_initial_obstacles(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
  for (int x = 1; x < lx - 1; x++)
    for (int y = 1; y < ly - 1; y++) s->obst[NI(s, x, y)] = -1;
  for (int x = 0; x < lx; x++) {
    s->obst[NI(s, x, 0)] = s->obst[NI(s, x, ly - 1)] = s->n;
    s->act[NI(s, x, 0)] = s->act[NI(s, x, ly - 1)] = 0;
  }
  for (int y = 1; y < ly - 1; y++) {
    s->obst[NI(s, 0, y)] = s->obst[NI(s, lx - 1, y)] = s->n;
    s->act[NI(s, 0, y)] = s->act[NI(s, lx - 1, y)] = 0;
  }
  for (int i = 0; i < s->n; i++) {
    grain_box b = grain_bbox(s, i);
    for (int x = b.xi; x <= b.xf; x++)
      for (int y = b.yi; y <= b.yf; y++) {
        double d2 = (x - b.xc) * (x - b.xc) + (y - b.yc) * (y - b.yc);
        if (d2 <= b.R2 && d2 <= b.r2) s->obst[NI(s, x, y)] = i;
      }
  }
}

/* ------------------------------------------------------------------ LBM phases */

/* rigid-body velocity of grain i at lattice node (x,y): main.c:974-975, 1172-1173 */
static inline double wall_ux(const ora_sim* s, int i, int y) {
  return s->v1[i] - (y * s->dx + s->Mby - s->x2[i]) * s->v3[i];
}
static inline double wall_uy(const ora_sim* s, int i, int x) {
  return s->v2[i] + (x * s->dx + s->Mgx - s->x1[i]) * s->v3[i];
}

/* main.c:966-986 */
ORA_API void ora_reinit_obst_density(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
  const double c = s->c;
#pragma omp parallel for if (s->threads)
  for (int x = 1; x < lx - 1; x++) {
    for (int y = 1; y < ly - 1; y++) {
      int i = s->obst[NI(s, x, y)];
      if (i == -1) continue;
      double ux = wall_ux(s, i, y), uy = wall_uy(s, i, x);
      double u_squ = (ux * ux + uy * uy) / (c * c);
      double* fn = s->f + FI(s, x, y, 0);
      for (int q = 0; q < NQ; q++) {
        double eu = (EX[q] * ux + EY[q] * uy) / c;
        fn[q] = WQ[q] * (1. + 3 * eu + 4.5 * eu * eu - 1.5 * u_squ);
      }
    }
  }
}

/* main.c:991-1065 */
ORA_API void ora_obst_construction(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
#pragma omp parallel for if (s->threads)
  for (int x = 1; x < lx - 1; x++) {
    for (int y = 1; y < ly - 1; y++) {
      s->obst[NI(s, x, y)] = -1;
      s->act[NI(s, x, y)] = 1;
      double* d = s->delta + FI(s, x, y, 0);
      for (int q = 1; q < NQ; q++) d[q] = 0;
    }
  }
  /* serial over grains: overlapping discs resolve as "highest index wins" (main.c:1028) */
  for (int i = 0; i < s->n; i++) {
    grain_box b = grain_bbox(s, i);
    for (int y = b.yi; y <= b.yf; y++)
      for (int x = b.xi; x <= b.xf; x++) {
        double d2 = (x - b.xc) * (x - b.xc) + (y - b.yc) * (y - b.yc);
        if (d2 <= b.R2 && d2 <= b.r2) s->obst[NI(s, x, y)] = i;
      }
    /* active solid nodes and the wall distance along each fluid link (main.c:1036-1062) */
    for (int y = b.yi; y <= b.yf; y++)
      for (int x = b.xi; x <= b.xf; x++) {
        if (s->obst[NI(s, x, y)] != i) continue;
        s->act[NI(s, x, y)] = 0;
        for (int q = 1; q < NQ; q++) {
          int nx = x + EX[q], ny = y + EY[q];
          if (s->obst[NI(s, nx, ny)] != -1) continue;
          s->act[NI(s, x, y)] = 1;
          double aa = fabs(EX[q]) + fabs(EY[q]);
          double bb = (x + EX[q] - b.xc) * EX[q] + (y + EY[q] - b.yc) * EY[q];
          double cc = (x + EX[q] - b.xc) * (x + EX[q] - b.xc) +
                      (y + EY[q] - b.yc) * (y + EY[q] - b.yc) - b.r2;
          s->delta[FI(s, x, y, q)] = (bb - sqrt(fabs(bb * bb - aa * cc))) / aa;
        }
      }
  }
}

/* MRT collision of one node, in place: main.c:1082-1116 */
static inline void mrt_collide_node(const ora_sim* s, double* fn) {
  const double a = 1. / 36;
  const double f0 = fn[0], f1 = fn[1], f2 = fn[2], f3 = fn[3], f4 = fn[4], f5 = fn[5], f6 = fn[6],
               f7 = fn[7], f8 = fn[8];
  double rho = f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + f8;
  double e = -4 * f0 + 2 * f1 - f2 + 2 * f3 - f4 + 2 * f5 - f6 + 2 * f7 - f8;
  double eps = 4 * f0 + f1 - 2 * f2 + f3 - 2 * f4 + f5 - 2 * f6 + f7 - 2 * f8;
  double j_x = f5 + f6 + f7 - f1 - f2 - f3;
  double q_x = -f1 + 2 * f2 - f3 + f5 - 2 * f6 + f7;
  double j_y = f1 + f8 + f7 - f3 - f4 - f5;
  double q_y = f1 - f3 + 2 * f4 - f5 + f7 - 2 * f8;
  double p_xx = f2 - f4 + f6 - f8;
  double p_xy = -f1 + f3 - f5 + f7;

  double j_x2 = j_x * j_x;
  double j_y2 = j_y * j_y;

  double eO = e - s->s2 * (e + 2 * rho - 3 * (j_x2 + j_y2) / rho);
  double epsO = eps - s->s3 * (eps - rho + 3 * (j_x2 + j_y2) / rho);
  double q_xO = q_x - s->s5 * (q_x + j_x);
  double q_yO = q_y - s->s7 * (q_y + j_y);
  double p_xxO = p_xx - s->s8 * (p_xx - (j_x2 - j_y2) / rho);
  double p_xyO = p_xy - s->s9 * (p_xy - j_x * j_y / rho);

  fn[0] = a * (4 * rho - 4 * eO + 4 * epsO);
  fn[2] = a * (4 * rho - eO - 2 * epsO - 6 * j_x + 6 * q_xO + 9 * p_xxO);
  fn[4] = a * (4 * rho - eO - 2 * epsO - 6 * j_y + 6 * q_yO - 9 * p_xxO);
  fn[6] = a * (4 * rho - eO - 2 * epsO + 6 * j_x - 6 * q_xO + 9 * p_xxO);
  fn[8] = a * (4 * rho - eO - 2 * epsO + 6 * j_y - 6 * q_yO - 9 * p_xxO);
  fn[1] = a * (4 * rho + 2 * eO + epsO - 6 * j_x - 3 * q_xO + 6 * j_y + 3 * q_yO - 9 * p_xyO);
  fn[3] = a * (4 * rho + 2 * eO + epsO - 6 * j_x - 3 * q_xO - 6 * j_y - 3 * q_yO + 9 * p_xyO);
  fn[5] = a * (4 * rho + 2 * eO + epsO + 6 * j_x + 3 * q_xO - 6 * j_y - 3 * q_yO - 9 * p_xyO);
  fn[7] = a * (4 * rho + 2 * eO + epsO + 6 * j_x + 3 * q_xO + 6 * j_y + 3 * q_yO + 9 * p_xyO);
}

/* main.c:1077-1119 */
ORA_API void ora_collide(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
#pragma omp parallel for if (s->threads)
  for (int x = 1; x < lx - 1; x++)
    for (int y = 1; y < ly - 1; y++)
      if (s->obst[NI(s, x, y)] == -1) mrt_collide_node(s, s->f + FI(s, x, y, 0));
}

/* main.c:1123-1145: resting bounce-back walls on all four lattice edges */
ORA_API void ora_edges(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
  double* f = s->f;
  for (int x = 1; x < lx - 1; x++) {
    f[FI(s, x, 0, 8)] = f[FI(s, x, 1, 4)];
    f[FI(s, x, 0, 7)] = f[FI(s, x + 1, 1, 3)];
    f[FI(s, x, 0, 1)] = f[FI(s, x - 1, 1, 5)];
    f[FI(s, x, ly - 1, 4)] = f[FI(s, x, ly - 2, 8)];
    f[FI(s, x, ly - 1, 3)] = f[FI(s, x - 1, ly - 2, 7)];
    f[FI(s, x, ly - 1, 5)] = f[FI(s, x + 1, ly - 2, 1)];
    if (s->lid_u != 0.0) { /* the commented-out lid terms of main.c:1129-1130 */
      f[FI(s, x, ly - 1, 3)] = f[FI(s, x - 1, ly - 2, 7)] - s->lid_u / 6;
      f[FI(s, x, ly - 1, 5)] = f[FI(s, x + 1, ly - 2, 1)] + s->lid_u / 6;
    }
  }
  for (int y = 1; y < ly - 1; y++) {
    f[FI(s, 0, y, 6)] = f[FI(s, 1, y, 2)];
    f[FI(s, 0, y, 7)] = f[FI(s, 1, y + 1, 3)];
    f[FI(s, 0, y, 5)] = f[FI(s, 1, y - 1, 1)];
    f[FI(s, lx - 1, y, 2)] = f[FI(s, lx - 2, y, 6)];
    f[FI(s, lx - 1, y, 3)] = f[FI(s, lx - 2, y - 1, 7)];
    f[FI(s, lx - 1, y, 1)] = f[FI(s, lx - 2, y + 1, 5)];
  }
  f[FI(s, 0, 0, 7)] = f[FI(s, 1, 1, 3)];
  f[FI(s, lx - 1, 0, 1)] = f[FI(s, lx - 2, 1, 5)];
  f[FI(s, 0, ly - 1, 5)] = f[FI(s, 1, ly - 2, 1)];
  f[FI(s, lx - 1, ly - 1, 3)] = f[FI(s, lx - 2, ly - 2, 7)];
}

/* main.c:1154-1222: interpolated bounce-back with moving-wall term on active solid nodes.
 * In place and in (x outer, y inner) scan order on purpose -- the 0<delta<1/2 branch reads a
 * node two links out which may already have been rewritten by this loop (SURVEY hard part 2). */
ORA_API void ora_grain_ibb(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
  const double c = s->c;
  double* f = s->f;
  for (int x = 1; x < lx - 1; x++) {
    for (int y = 1; y < ly - 1; y++) {
      int i = s->obst[NI(s, x, y)];
      if (i == -1 || s->act[NI(s, x, y)] != 1) continue;
      for (int q = 1; q < NQ; q++) {
        const int qo = (q <= HALF) ? q + HALF : q - HALF;
        int nx = x + EX[q], ny = y + EY[q];
        if (s->obst[NI(s, nx, ny)] != -1) {
          f[FI(s, x, y, q)] = WQ[q];
          continue;
        }
        const double d = s->delta[FI(s, x, y, q)];
        if (d >= 0.5) {
          f[FI(s, x, y, q)] = f[FI(s, nx, ny, qo)] / (2 * d) +
                              (2 * d - 1) * f[FI(s, nx, ny, q)] / (2 * d) +
                              3 * (WQ[q] / c) * (EX[q] * wall_ux(s, i, y) + EY[q] * wall_uy(s, i, x)) / d;
        }
        if (d > 0. && d < 0.5) {
          int nnx = nx + EX[q], nny = ny + EY[q];
          f[FI(s, x, y, q)] = 2 * d * f[FI(s, nx, ny, qo)] +
                              (1 - 2 * d) * f[FI(s, nnx, nny, qo)] +
                              6 * (WQ[q] / c) * (EX[q] * wall_ux(s, i, y) + EY[q] * wall_uy(s, i, x));
        }
      }
    }
  }
}

/* main.c:1224-1242: two-pass "swap" streaming over every node, borders and solids included */
ORA_API void ora_swap_stream(ora_sim* s) {
  const int lx = s->lx, ly = s->ly;
  double* f = s->f;
  for (int x = 0; x < lx; x++)
    for (int y = 0; y < ly; y++) {
      double* fn = f + FI(s, x, y, 0);
      for (int q = 1; q <= HALF; q++) {
        double t = fn[q]; fn[q] = fn[q + HALF]; fn[q + HALF] = t;
      }
    }
  for (int x = 0; x < lx; x++)
    for (int y = 0; y < ly; y++)
      for (int q = 1; q <= HALF; q++) {
        int nx = x + EX[q], ny = y + EY[q];
        if (nx >= 0 && ny >= 0 && nx < lx && ny < ly) {
          double* a = f + FI(s, x, y, q + HALF);
          double* b = f + FI(s, nx, ny, q);
          double t = *a; *a = *b; *b = t;
        }
      }
}

/* main.c:1071-1243 */
ORA_API void ora_collision_streaming(ora_sim* s) {
  ora_collide(s);
  ora_edges(s);
  ora_grain_ibb(s);
  ora_swap_stream(s);
}

/* main.c:1285-1333: momentum exchange over each grain's bounding box, serial x->y->q order */
ORA_API void ora_forces_fluid(ora_sim* s) {
  const int n = s->n, nx_ = s->lx, ny_ = s->ly;
  const double* f = s->f;
  for (int i = 0; i < n; ++i) { s->fhf1[i] = 0; s->fhf2[i] = 0; s->fhf3[i] = 0; }
#pragma omp parallel for if (s->threads)
  for (int i = 0; i < n; ++i) {
    const double xc = (s->x1[i] - s->Mgx) / s->dx;
    const double yc = (s->x2[i] - s->Mby) / s->dx;
    const double rbl0 = s->r[i] / s->dx;
    /* int max(int,int)/min(int,int) in the reference: the double argument is truncated first */
    int xi = (int)(xc - rbl0); if (xi < 1) xi = 1;
    int xf = (int)(xc + rbl0); if (xf > nx_ - 2) xf = nx_ - 2;
    int yi = (int)(yc - rbl0); if (yi < 1) yi = 1;
    int yf = (int)(yc + rbl0); if (yf > ny_ - 2) yf = ny_ - 2;
    double h1 = 0, h2 = 0, h3 = 0;
    for (int x = xi; x <= xf; ++x)
      for (int y = yi; y <= yf; ++y) {
        if (s->obst[NI(s, x, y)] != i) continue;
        for (int q = 1; q < NQ; ++q) {
          int nx = x + EX[q], ny = y + EY[q];
          if (s->obst[NI(s, nx, ny)] == i) continue;
          const int qo = (q <= HALF) ? q + HALF : q - HALF;
          const double fnx = (f[FI(s, x, y, qo)] + f[FI(s, nx, ny, q)]) * EX[qo];
          const double fny = (f[FI(s, x, y, qo)] + f[FI(s, nx, ny, q)]) * EY[qo];
          h1 = h1 + fnx;
          h2 = h2 + fny;
          h3 = h3 - fnx * (y - (s->x2[i] - s->Mby) / s->dx) + fny * (x - (s->x1[i] - s->Mgx) / s->dx);
        }
      }
    s->fhf1[i] = h1; s->fhf2[i] = h2; s->fhf3[i] = h3;
  }
  const double dx = s->dx, tau = s->tau, nu = s->nu, rho_moy = s->rho_moy;
#pragma omp parallel for if (s->threads)
  for (int i = 0; i < n; ++i) {
    s->fhf1[i] *= rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5));
    s->fhf2[i] *= rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5));
    s->fhf3[i] *= dx * rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5));
  }
}

/* Test helpers for the strip-decomposition protocol (tests/strip_backends.py; no counterpart in the reference, which
 * has one address space). The momentum-exchange sums f[P][opp q] + f[N][q] (main.c:1313-1314) of grain i's links
 * whose far end N = P + e_q lies in rows [nlo, nhi), in the reference's scan order, as {x, y, q, sum} ... */
ORA_API int ora_link_sums(ora_sim* s, int i, int nlo, int nhi, double* out4, int cap) {
  const int nx_ = s->lx, ny_ = s->ly;
  const double* f = s->f;
  const double xc = (s->x1[i] - s->Mgx) / s->dx;
  const double yc = (s->x2[i] - s->Mby) / s->dx;
  const double rbl0 = s->r[i] / s->dx;
  int xi = (int)(xc - rbl0); if (xi < 1) xi = 1;
  int xf = (int)(xc + rbl0); if (xf > nx_ - 2) xf = nx_ - 2;
  int yi = (int)(yc - rbl0); if (yi < 1) yi = 1;
  int yf = (int)(yc + rbl0); if (yf > ny_ - 2) yf = ny_ - 2;
  int cnt = 0;
  for (int x = xi; x <= xf; ++x)
    for (int y = yi; y <= yf; ++y) {
      if (s->obst[NI(s, x, y)] != i) continue;
      for (int q = 1; q < NQ; ++q) {
        int nx = x + EX[q], ny = y + EY[q];
        if (s->obst[NI(s, nx, ny)] == i) continue;
        if (nx < nlo || nx >= nhi) continue;
        const int qo = (q <= HALF) ? q + HALF : q - HALF;
        if (cnt < cap) {
          out4[4 * cnt] = x; out4[4 * cnt + 1] = y; out4[4 * cnt + 2] = q;
          out4[4 * cnt + 3] = f[FI(s, x, y, qo)] + f[FI(s, nx, ny, q)];
        }
        ++cnt;
      }
    }
  return cnt;
}

/* ... and the hydrodynamic force of grain i from the complete list of its links' sums in scan order: the additions and
 * the scaling of main.c:1315-1331. */
ORA_API void ora_force_from_link_sums(ora_sim* s, int i, const double* in4, int n) {
  double h1 = 0, h2 = 0, h3 = 0;
  for (int k = 0; k < n; ++k) {
    const int x = (int)in4[4 * k], y = (int)in4[4 * k + 1], q = (int)in4[4 * k + 2];
    const double sum = in4[4 * k + 3];
    const int qo = (q <= HALF) ? q + HALF : q - HALF;
    const double fnx = sum * EX[qo];
    const double fny = sum * EY[qo];
    h1 = h1 + fnx;
    h2 = h2 + fny;
    h3 = h3 - fnx * (y - (s->x2[i] - s->Mby) / s->dx) + fny * (x - (s->x1[i] - s->Mgx) / s->dx);
  }
  const double dx = s->dx, tau = s->tau, nu = s->nu, rho_moy = s->rho_moy;
  s->fhf1[i] = h1 * (rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5)));
  s->fhf2[i] = h2 * (rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5)));
  s->fhf3[i] = h3 * (dx * rho_moy * 9 * nu * nu / (dx * (tau - 0.5) * (tau - 0.5)));
}

ORA_API void ora_lbm_steps(ora_sim* s, int n) {
  for (int k = 0; k < n; ++k) {
    ora_reinit_obst_density(s);
    ora_obst_construction(s);
    ora_collision_streaming(s);
    ora_forces_fluid(s);
  }
}

/* ------------------------------------------------------------------ DEM */

typedef struct { double f1, f2, f3; } force3;

static inline double maxt(double x, double y) { return (x < y) ? 0. : y; } /* main.c:211-216 */

/* regular contact law between grains i<j, with its diagnostics: main.c:729-803 */
static force3 contact_regular(ora_sim* s, int i, int j) {
  force3 F = {0, 0, 0};
  double xij = s->x1[i] - s->x1[j];
  double yij = s->x2[i] - s->x2[j];
  double dist = sqrt(xij * xij + yij * yij);
  double dn = dist - s->r[i] - s->r[j];
  if (dn >= 0) return F;
  double vx = s->v1[i] - s->v1[j];
  double vy = s->v2[i] - s->v2[j];
  double xn = xij / dist;
  double yn = yij / dist;
  double vn = vx * xn + vy * yn;
  double vt = -vx * yn + vy * xn - s->v3[i] * s->r[i] - s->v3[j] * s->r[j];
  double fn = -s->kg * dn - s->nug * vn;
  if (fn < 0) fn = 0.0;
  double ft = -s->kt * vt * s->dt;
  double ftest = s->mu * fn;
  if (fabs(ft) > ftest) ft = (ft < 0.0) ? ftest : -ftest;
  F.f1 = fn * xn - ft * yn;
  F.f2 = fn * yn + ft * xn;
  F.f3 = -maxt(ft * s->r[i], fn * s->murf * s->r[i] * s->r[j]);

  s->p[i] += fn; s->p[j] += fn;
  s->f1[i] += F.f1; s->f2[i] += F.f2;
  s->s[i] += ft; s->s[j] += ft;
  s->slip[i] += fabs(ft) * (fabs(vt * s->dt) + (fabs(ft - s->pft)) / s->kt);
  s->pft = ft;
  s->rw[i] += fabs(F.f3) * (fabs(s->v3[i] * s->dt) + (fabs(F.f3 - s->pff)) / s->kt);
  s->pff = F.f3;
  s->z[i] += 1; s->zz[i] += 1;
  s->ice[i] += s->ic;
  if (fn == 0) s->ifm[i] = 0; else s->ifm[i] += fabs(ft / (s->mu * fn));
  s->M11[i] += F.f1 * xij; s->M12[i] += F.f1 * yij;
  s->M21[i] += F.f2 * xij; s->M22[i] += F.f2 * yij;
  return F;
}

/* the alternate law used on film steps (nbsteps % 8000 == 0): main.c:1365-1417 */
static force3 contact_film(ora_sim* s, int i, int j) {
  force3 F = {0, 0, 0};
  double xij = s->x1[i] - s->x1[j];
  double yij = s->x2[i] - s->x2[j];
  double dist = sqrt(xij * xij + yij * yij);
  double dn = dist - s->r[i] - s->r[j];
  if (dn >= 0) return F;
  double vx = s->v1[i] - s->v1[j];
  double vy = s->v2[i] - s->v2[j];
  double xn = xij / dist;
  double yn = yij / dist;
  double vn = vx * xn + vy * yn;
  double vt = -vx * yn + vy * xn - s->v3[i] * s->r[i] - s->v3[j] * s->r[j];
  double fn = -s->kg * dn - s->nug * vn;
  if (fn < 0) fn = 0.0;
  double ft = s->kt * vt * s->dt;
  double ftest = s->mu * ft; /* sic */
  if (fabs(ft) > ftest) ft = (ft > 0.0) ? ftest : -ftest;
  F.f1 = fn * xn - ft * yn;
  F.f2 = fn * yn + ft * xn;
  F.f3 = -ft * s->r[i] * s->murf;

  s->p[i] += fn; s->p[j] += fn;
  s->s[i] += ft; s->s[j] += ft;
  double dslip = fabs(ft) * (fabs(vt * s->dt) + (fabs(ft - s->pft)) / s->kt);
  s->slip[i] += dslip; s->slip[j] += dslip;
  double drw = fabs(F.f3) * (fabs(s->v3[i] * s->dt) + (fabs(F.f3 - s->pff)) / s->kt);
  s->rw[i] += drw; s->rw[j] += drw;
  s->z[i] += 1;
  s->pff = F.f3;
  s->pft = ft;
  s->M11[i] += F.f1 * xij; s->M12[i] += F.f1 * yij;
  s->M21[i] += F.f2 * xij; s->M22[i] += F.f2 * yij;
  return F;
}

/* main.c:809-845 */
static force3 wall_bottom(ora_sim* s, int i, double dn) {
  force3 F;
  double vn = s->v2[i], vt = s->v1[i];
  double fn = -s->km * dn - s->num * vn;
  if (fn < 0) fn = 0.;
  double ft = s->ktm * vt;
  double ftest = s->mumb * fn;
  if (fabs(ft) > ftest) ft = (ft < 0.0) ? ftest : -ftest;
  F.f1 = ft; F.f2 = fn; F.f3 = -(ft * s->r[i] * s->murf);
  s->p[i] += fn; s->s[i] += ft; s->f1[i] += F.f1; s->z[i] += 1;
  s->M11[i] += 0; s->M12[i] += F.f1 * s->dt; s->M21[i] += 0; s->M22[i] += F.f2 * s->dt;
  s->rw[i] += fabs(F.f3) * (fabs(s->v3[i] * s->dt) + (fabs(F.f3 - s->pff)) / s->kt);
  s->fr[i] += fabs(ft) * (fabs(vt * s->dt) + (fabs(ft - s->pft)) / s->kt);
  s->pff = F.f3; s->pft = ft;
  return F;
}

/* main.c:846-887 */
static force3 wall_top(ora_sim* s, int i, double dn) {
  force3 F;
  double vn = s->v2[i];
  double fn = s->km * dn - s->num * vn;
  s->ic += s->num * vn * vn * s->dt;
  if (fn > 0.) fn = 0.;
  double vt = s->v1[i] + s->v3[i] * s->r[i] - s->amp * s->freq * cos(s->freq * s->t);
  double ft = fabs(s->ktm * vt);
  double ftmax;
  if (vt >= 0) ftmax = s->mumb * fn - s->nugt * vt; else ftmax = s->mumb * fn + s->nugt * vt;
  if (ft > ftmax) ft = ftmax;
  if (vt > 0) ft = -ft;
  F.f1 = ft; F.f2 = fn; F.f3 = ft * s->r[i] * s->murf;
  s->M11[i] += 0; s->M12[i] += F.f1 * fabs(s->dt); s->M21[i] += 0; s->M22[i] += F.f2 * fabs(s->dt);
  s->p[i] += fn; s->s[i] += ft; s->z[i] += 1;
  return F;
}

/* main.c:888-921 */
static force3 wall_left(ora_sim* s, int i, double dn) {
  force3 F;
  double vn = s->v1[i];
  double fn = -s->km * dn + s->num * vn;
  s->ic += s->num * vn * vn * s->dt;
  if (fn < 0.) fn = 0.;
  double vt = s->v2[i];
  double ft = s->mum * fn;
  if (vt > 0) ft = -ft;
  F.f1 = fn; F.f2 = ft; F.f3 = ft * s->r[i] * s->murf;
  s->M11[i] += F.f1 * fabs(s->dt); s->M12[i] += 0; s->M21[i] += F.f2 * fabs(s->dt); s->M22[i] += 0;
  s->p[i] += fn; s->s[i] += ft; s->f1[i] += F.f1; s->z[i] += 1;
  s->ice[i] += s->ic;
  s->rw[i] += fabs(F.f3) * fabs(s->v3[i] * s->dt);
  s->fr[i] += fabs(ft) * (fabs(vt * s->dt) + (fabs(ft - s->pft)) / s->kt);
  s->pft = ft;
  return F;
}

/* main.c:923-951 (ft is taken from fn *before* fn is clamped) */
static force3 wall_right(ora_sim* s, int i, double dn) {
  force3 F;
  double vn = s->v1[i];
  double fn = s->km * dn - s->num * vn;
  double vt = s->v2[i];
  double ft = s->mum * fn;
  if (vt > 0) ft = -ft;
  if (fn > 0.) fn = 0.;
  F.f1 = fn; F.f2 = -ft; F.f3 = ft * s->r[i] * s->murf;
  s->p[i] += fn; s->f1[i] += F.f1;
  s->pft = ft;
  s->M11[i] += F.f1 * fabs(s->dt); s->M12[i] += 0; s->M21[i] += F.f2 * fabs(s->dt); s->M22[i] += 0;
  s->z[i] += 1;
  return F;
}

/* main.c:1336-1516 */
static void acceleration_grains(ora_sim* s) {
  const int n = s->n;
  const int film = (s->nbsteps % s->stepFilm == 0);
  for (int i = 0; i < n; i++) { s->a1[i] = s->fhf1[i]; s->a2[i] = s->fhf2[i]; s->a3[i] = s->fhf3[i]; }
  for (int i = 0; i < n; i++) {
    int jdep = (i == 0) ? 0 : s->cumul[i - 1];
    for (int k = jdep; k < s->cumul[i]; k++) {
      int j = s->neighbours[k];
      force3 F = film ? contact_film(s, i, j) : contact_regular(s, i, j);
      s->a1[i] = s->a1[i] + F.f1; s->a2[i] = s->a2[i] + F.f2; s->a3[i] = s->a3[i] + F.f3;
      s->a1[j] = s->a1[j] - F.f1; s->a2[j] = s->a2[j] - F.f2; s->a3[j] = s->a3[j] + F.f3;
    }
  }
  /* walls, in the order bottom, top, left, right (main.c:1455-1508). The `fr` updates index the
   * grain table with the *list position* k, as the reference does (main.c:1462-1465,1490-1493). */
  for (int k = 0; k < s->nB; k++) {
    int i = s->wallB[k];
    double dn = s->x2[i] - s->r[i] - s->Mby;
    if (dn < 0) {
      force3 F = wall_bottom(s, i, dn);
      s->a1[i] = s->a1[i] + F.f1; s->a2[i] = s->a2[i] + F.f2; s->a3[i] = s->a3[i] + F.f3;
      s->fr[i] += fabs(F.f1) * (fabs(s->dt * s->v1[k]) + fabs(s->dt2 * s->a1[k]) + (fabs(F.f1 - s->pf)) / s->kt);
      s->pf = F.f1;
    }
  }
  for (int k = 0; k < s->nT; k++) {
    int i = s->wallT[k];
    double dn = -s->x2[i] - s->r[i] + s->Mhy;
    if (dn < 0) {
      force3 F = wall_top(s, i, dn);
      s->a1[i] = s->a1[i] + F.f1; s->a2[i] = s->a2[i] + F.f2; s->a3[i] = s->a3[i] + F.f3;
    }
  }
  for (int k = 0; k < s->nL; k++) {
    int i = s->wallL[k];
    double dn = s->x1[i] - s->r[i] - s->Mgx;
    if (dn < 0) {
      force3 F = wall_left(s, i, dn);
      s->a1[i] = s->a1[i] + F.f1; s->a2[i] = s->a2[i] + F.f2; s->a3[i] = s->a3[i] + F.f3;
      s->fr[i] += fabs(F.f2) * (fabs(s->dt * s->v1[k]) + fabs(s->dt2 * s->a1[k]) + (fabs(F.f2 - s->pf)) / s->kt);
      s->pf = F.f2;
    }
  }
  for (int k = 0; k < s->nR; k++) {
    int i = s->wallR[k];
    double dn = -s->x1[i] - s->r[i] + s->Mdx;
    if (dn < 0) {
      force3 F = wall_right(s, i, dn);
      s->a1[i] = s->a1[i] + F.f1; s->a2[i] = s->a2[i] + F.f2; s->a3[i] = s->a3[i] + F.f3;
    }
  }
  /* main.c:1511-1515 */
  for (int i = 0; i < n; i++) {
    s->a1[i] = s->a1[i] / s->m[i] + ((s->m[i] - s->mw[i]) / s->m[i]) * s->xG;
    s->a2[i] = (s->a2[i] / s->m[i]) + ((s->m[i] - s->mw[i]) / s->m[i]) * s->yG;
    s->a3[i] = s->a3[i] / s->It[i];
  }
}

/* main.c:1519-1594 */
ORA_API void ora_verlet_rebuild(ora_sim* s) {
  const int n = s->n;
  const double dV = s->distVerlet;
  int cnt = 0;
  for (int i = 0; i < n; i++) {
    for (int j = i + 1; j < n; j++) {
      double ddx = s->x1[i] - s->x1[j];
      double ddy = s->x2[i] - s->x2[j];
      if (((fabs(ddx) - s->r[i] - s->r[j]) <= dV) && ((fabs(ddy) - s->r[i] - s->r[j]) <= dV)) {
        if ((sqrt(ddx * ddx + ddy * ddy) - s->r[i] - s->r[j]) <= dV) {
          if (cnt == s->neigh_cap) {
            s->neigh_cap *= 2;
            s->neighbours = (int*)realloc(s->neighbours, sizeof(int) * s->neigh_cap);
          }
          s->neighbours[cnt++] = j;
        }
      }
      s->cumul[i] = cnt; /* never written for i = n-1, as in the reference (main.c:1539) */
    }
  }
  /* VerletWall: main.c:1545-1594 */
  s->nB = s->nL = s->nT = s->nR = 0;
  if (s->nbsteps * s->dt < s->dtt) {
    s->Mdx = 1.e-3 * s->lx / 10;
    s->Mhy = (1.e-3 * s->ly / 10);
  } else {
    s->Mdx = 1.e-3 * s->lx;
    s->Mhy = 1.e-3 * s->ly;
  }
  for (int i = 0; i < n; ++i) if (s->x2[i] - s->r[i] - s->Mby < dV) s->wallB[s->nB++] = i;
  for (int i = 0; i < n; ++i) if (-s->x2[i] - s->r[i] + s->Mhy < dV) s->wallT[s->nT++] = i;
  for (int i = 0; i < n; ++i) if (s->x1[i] - s->r[i] - s->Mgx < dV) s->wallL[s->nL++] = i;
  for (int i = 0; i < n; ++i) if (-s->x1[i] - s->r[i] + s->Mdx < dV) s->wallR[s->nR++] = i;
}

/* main.c:1733-1764 */
ORA_API void ora_dem_substep(ora_sim* s) {
  const int n = s->n;
  const double dt = s->dt, dt2 = s->dt2;
  for (int i = 0; i < n; i++) {
    s->p[i] = 0; s->s[i] = 0.; s->ifm[i] = 0; s->f1[i] = 0.; s->f2[i] = 0.; s->ice[i] = 0;
    s->fr[i] = 0.; s->slip[i] = 0; s->rw[i] = 0.;
    s->ic = 0.;
    s->M11[i] = s->M12[i] = s->M21[i] = s->M22[i] = 0.;
    s->z[i] = 0; s->zz[i] = 0;
    s->x1[i] = s->x1[i] + dt * s->v1[i] + dt2 * s->a1[i] / 2.;
    s->x2[i] = s->x2[i] + dt * s->v2[i] + dt2 * s->a2[i] / 2.;
    s->x3[i] = s->x3[i] + dt * s->v3[i] + dt2 * s->a3[i] / 2.;
    s->v1[i] = s->v1[i] + dt * s->a1[i] / 2.;
    s->v2[i] = s->v2[i] + dt * s->a2[i] / 2.;
    s->v3[i] = s->v3[i] + dt * s->a3[i] / 2.;
  }
  acceleration_grains(s);
  for (int i = 0; i < n; i++) {
    s->v1[i] = s->v1[i] + dt * s->a1[i] / 2.;
    s->v2[i] = s->v2[i] + dt * s->a2[i] / 2.;
    s->v3[i] = s->v3[i] + dt * s->a3[i] / 2.;
  }
  s->nbsteps++;
}

/* main.c:1697-1777 without the file writers */
ORA_API void ora_render_scene(ora_sim* s) {
  if (s->nbsteps % s->npDEM == 0) {
    ora_reinit_obst_density(s);
    ora_obst_construction(s);
    ora_collision_streaming(s);
    ora_forces_fluid(s);
  }
  if (s->nbsteps % s->updateVerlet == 0) ora_verlet_rebuild(s);
  ora_dem_substep(s);
}

ORA_API void ora_steps(ora_sim* s, long n) {
  for (long k = 0; k < n; ++k) ora_render_scene(s);
}

/* renderScene with _FLUIDE_ undefined (main.c:16, 1709-1719): no fluid step, the hydrodynamic forces stay what they
 * are (0 from the start), everything else as above */
ORA_API void ora_steps_dry(ora_sim* s, long n) {
  for (long k = 0; k < n; ++k) {
    if (s->nbsteps % s->updateVerlet == 0) ora_verlet_rebuild(s);
    ora_dem_substep(s);
  }
}

/* ------------------------------------------------------------------ access */

ORA_API void ora_set_threads(ora_sim* s, int nthreads) { s->threads = nthreads > 1; }
ORA_API int ora_lx(const ora_sim* s) { return s->lx; }
ORA_API int ora_ly(const ora_sim* s) { return s->ly; }
ORA_API int ora_n(const ora_sim* s) { return s->n; }
ORA_API long ora_nbsteps(const ora_sim* s) { return s->nbsteps; }
ORA_API void ora_set_nbsteps(ora_sim* s, long n) { s->nbsteps = n; }
ORA_API double* ora_f(ora_sim* s) { return s->f; }
ORA_API int* ora_obst(ora_sim* s) { return s->obst; }
ORA_API int* ora_act(ora_sim* s) { return s->act; }
ORA_API double* ora_delta(ora_sim* s) { return s->delta; }

ORA_API void ora_get_fhf(const ora_sim* s, double* out) {
  for (int i = 0; i < s->n; ++i) {
    out[3 * i] = s->fhf1[i]; out[3 * i + 1] = s->fhf2[i]; out[3 * i + 2] = s->fhf3[i];
  }
}

ORA_API void ora_set_fhf(ora_sim* s, const double* in) {
  for (int i = 0; i < s->n; ++i) {
    s->fhf1[i] = in[3 * i]; s->fhf2[i] = in[3 * i + 1]; s->fhf3[i] = in[3 * i + 2];
  }
}

ORA_API void ora_get_grains(const ora_sim* s, double* out) {
  for (int i = 0; i < s->n; ++i) {
    double* o = out + (size_t)i * ORA_GRAIN_COLS;
    o[0] = s->x1[i]; o[1] = s->x2[i]; o[2] = s->x3[i];
    o[3] = s->v1[i]; o[4] = s->v2[i]; o[5] = s->v3[i];
    o[6] = s->a1[i]; o[7] = s->a2[i]; o[8] = s->a3[i];
    o[9] = s->r[i]; o[10] = s->m[i]; o[11] = s->mw[i]; o[12] = s->It[i];
    o[13] = s->p[i]; o[14] = s->s[i]; o[15] = s->f1[i]; o[16] = s->f2[i];
    o[17] = s->ifm[i]; o[18] = s->fm[i]; o[19] = s->fr[i]; o[20] = s->ifr[i];
    o[21] = s->M11[i]; o[22] = s->M12[i]; o[23] = s->M21[i]; o[24] = s->M22[i];
    o[25] = s->ice[i]; o[26] = s->slip[i]; o[27] = s->rw[i];
    o[28] = s->z[i]; o[29] = s->zz[i];
  }
}

ORA_API void ora_set_kinematics(ora_sim* s, const double* in) {
  for (int i = 0; i < s->n; ++i) {
    const double* p = in + (size_t)i * 9;
    s->x1[i] = p[0]; s->x2[i] = p[1]; s->x3[i] = p[2];
    s->v1[i] = p[3]; s->v2[i] = p[4]; s->v3[i] = p[5];
    s->a1[i] = p[6]; s->a2[i] = p[7]; s->a3[i] = p[8];
  }
}

ORA_API void ora_get_scalars(const ora_sim* s, double* out) {
  out[0] = s->dx; out[1] = s->dtLB; out[2] = s->dt; out[3] = s->dt2; out[4] = s->c;
  out[5] = s->npDEM; out[6] = s->Mgx; out[7] = s->Mdx; out[8] = s->Mby; out[9] = s->Mhy;
  out[10] = s->xG; out[11] = s->yG;
}

ORA_API void ora_get_rlb(const ora_sim* s, double* out) {
  memcpy(out, s->rLB, sizeof(double) * s->n);
}

ORA_API int ora_verlet_capacity(const ora_sim* s) { return s->neigh_cap; }

ORA_API void ora_get_verlet(const ora_sim* s, int* cumul, int* neighbours, int* counts, int* wb,
                            int* wt, int* wl, int* wr) {
  memcpy(cumul, s->cumul, sizeof(int) * s->n);
  memcpy(neighbours, s->neighbours, sizeof(int) * s->neigh_cap);
  counts[0] = s->nB; counts[1] = s->nT; counts[2] = s->nL; counts[3] = s->nR;
  memcpy(wb, s->wallB, sizeof(int) * s->n);
  memcpy(wt, s->wallT, sizeof(int) * s->n);
  memcpy(wl, s->wallL, sizeof(int) * s->n);
  memcpy(wr, s->wallR, sizeof(int) * s->n);
}

/* main.c:1249-1273 */
ORA_API double ora_total_density(const ora_sim* s) {
  double sum = 0;
  const size_t tot = (size_t)s->lx * s->ly * NQ;
  for (size_t k = 0; k < tot; ++k) sum = sum + s->f[k];
  return sum;
}

ORA_API long ora_count_act_anomalies(const ora_sim* s) {
  long bad = 0;
  for (int x = 1; x < s->lx - 1; x++)
    for (int y = 1; y < s->ly - 1; y++) {
      const int o = s->obst[NI(s, x, y)];
      if (o == -1) continue;
      grain_box b = grain_bbox(s, o);
      int any = 0;
      for (int q = 1; q < NQ; q++) {
        const int nx = x + EX[q], ny = y + EY[q];
        const int on = s->obst[NI(s, nx, ny)];
        if (on == -1) any = 1;
        else if (on > o && on != s->n) {
          /* covered by a later-painted grain: it was fluid when `o` was painted unless it lies
           * inside o's own disc */
          double d2 = (nx - b.xc) * (nx - b.xc) + (ny - b.yc) * (ny - b.yc);
          if (!(d2 <= b.R2 && d2 <= b.r2)) any = 1;
        }
      }
      if (any != (s->act[NI(s, x, y)] == 1)) ++bad;
    }
  return bad;
}
