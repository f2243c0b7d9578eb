/*
 * lbmdem_oracle.h -- CPU ORACLE: TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * A serial C restatement of the hot path of cb-geo/2d-lbm-dem (reference src/main.c), written for
 * this repository so that (a) the HIP path has a checker that travels to the GPU box (the reference
 * source does not), and (b) bench.py has a CPU baseline to time on the GPU node's host cores.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 *
 * Parity status: PINNED. tests/test_oracle_vs_reference.py compares this restatement bit-for-bit
 * with the unmodified reference TU compiled in-container (oracle/ref_harness.c -> oracle/_ref/),
 * and tests/golden/ holds dumps generated from that reference build (tests/golden/make_golden.py).
 * Build with -O2 -ffp-contract=off (oracle/Makefile) -- any other flags void bit parity.
 *
 * Host layout is the reference's: f[x][y][q] with x the slow axis (main.c:56,1802).
 */
#ifndef LBMDEM_ORACLE_H
#define LBMDEM_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_sim ora_sim;

/* columns of the grain table returned by ora_get_grains (same order as the reference struct,
 * main.c:182-197, so dumps from oracle/ref_harness.c compare 1:1) */
#define ORA_GRAIN_COLS 30

/* Sample reader, restating read_sample (main.c:609-639): line 1 = comment, line 2 = count,
 * then count x "r x y[;]". Values are returned already scaled by the reference's r = 1e-3
 * (main.c:114,624-628). Caller frees the three arrays with ora_free(). Returns <0 on error. */
int ora_read_sample(const char* path, int* n, double** r, double** x1, double** x2);
void ora_free(void* p);

/* Build a simulation exactly as main() does (main.c:1798-1861): derives dx, dtLB, npDEM, c, dt,
 * rLB, fills f with w[q], paints the initial obstacle map. Radii/positions in metres. */
ora_sim* ora_create(int lx, int ly, double scale, int n, const double* r, const double* x1,
                    const double* x2);
void ora_destroy(ora_sim* s);
/* test-only: all physics constants in the order of lbmdem_physics (29 doubles), + the two cadences */
void ora_set_physics(ora_sim* s, const double* p29, int updateVerlet, int stepFilm);
/* test-only: change reductionR (main.c:94) right after ora_create */
void ora_set_reduction(ora_sim* s, double reductionR);

/* one renderScene() (main.c:1697-1777, file output excluded) = one DEM sub-step, with the
 * reference's cadences: LBM step every npDEM calls, Verlet rebuild every 100, film law every 8000 */
void ora_render_scene(ora_sim* s);
void ora_steps(ora_sim* s, long n);

/* the individual phases */
void ora_reinit_obst_density(ora_sim* s);  /* main.c:966-986   */
void ora_obst_construction(ora_sim* s);    /* main.c:991-1065  */
void ora_collision_streaming(ora_sim* s);  /* main.c:1071-1243 */
void ora_forces_fluid(ora_sim* s);         /* main.c:1285-1333 */
void ora_lbm_steps(ora_sim* s, int n);     /* n x (the four above, in that order) */
void ora_verlet_rebuild(ora_sim* s);       /* main.c:1519-1594 (initVerlet + VerletWall) */
void ora_dem_substep(ora_sim* s);          /* main.c:1733-1764 */

/* phase split of collision_streaming, for tests that need the intermediate ("pre-IBB") state */
void ora_collide(ora_sim* s);              /* main.c:1077-1119 */
void ora_edges(ora_sim* s);                /* main.c:1123-1145 */
void ora_grain_ibb(ora_sim* s);            /* main.c:1154-1222 */
void ora_swap_stream(ora_sim* s);          /* main.c:1224-1242 */

/* OpenMP on the six loops the reference annotates (main.c:967,996,1007,1077,1294,1327); only
 * meaningful in the *_fast build used for CPU timing. 0/1. */
void ora_set_threads(ora_sim* s, int nthreads);

/* state access (host layout) */
int ora_lx(const ora_sim* s);
int ora_ly(const ora_sim* s);
int ora_n(const ora_sim* s);
long ora_nbsteps(const ora_sim* s);
void ora_set_nbsteps(ora_sim* s, long n);
double* ora_f(ora_sim* s);        /* [lx][ly][9] */
int* ora_obst(ora_sim* s);        /* [lx][ly]    */
int* ora_act(ora_sim* s);         /* [lx][ly]    */
double* ora_delta(ora_sim* s);    /* [lx][ly][9] */
void ora_get_fhf(const ora_sim* s, double* out3n);                /* interleaved f1,f2,f3 */
void ora_set_fhf(ora_sim* s, const double* in3n);                  /* test-only: strip-decomposition tests */
void ora_get_grains(const ora_sim* s, double* out);               /* n x ORA_GRAIN_COLS */
void ora_set_kinematics(ora_sim* s, const double* in9n);          /* x1 x2 x3 v1 v2 v3 a1 a2 a3 */
/* scalars: dx dtLB dt dt2 c npDEM Mgx Mdx Mby Mhy xG yG */
void ora_get_scalars(const ora_sim* s, double* out12);
void ora_get_rlb(const ora_sim* s, double* out);
/* Verlet lists: cumul[n], neighbours[cap] (cap returned), wall counts {B,T,L,R} and lists */
int ora_verlet_capacity(const ora_sim* s);
void ora_get_verlet(const ora_sim* s, int* cumul, int* neighbours, int* counts4, int* wb, int* wt,
                    int* wl, int* wr);
double ora_total_density(const ora_sim* s);  /* main.c:1249-1273 summation order */
/* EXTENSION (not reference-pinned): enable the lid terms the reference has commented out at main.c:1129-1130 */
void ora_set_lid(ora_sim* s, double uw_h);
/* test-only (strip-decomposition protocol): link sums {x, y, q, f[P][opp q] + f[N][q]} of grain i whose far end N
 * lies in rows [nlo, nhi), scan order; and the force of grain i from its complete ordered list of sums */
int ora_link_sums(ora_sim* s, int i, int nlo, int nhi, double* out4, int cap);
void ora_force_from_link_sums(ora_sim* s, int i, const double* in4, int n);

/* Count of solid nodes whose reference `act` flag (set while grains are painted one after the other,
 * main.c:1039-1052) differs from what the HIP path derives from the FINAL obstacle map: "has a
 * fluid neighbour, or a neighbour covered by a higher-index grain that is outside the owner's own
 * disc". A difference needs three mutually overlapping reduced discs; parity tests
 * assert this is 0 on their inputs. */
long ora_count_act_anomalies(const ora_sim* s);

#ifdef __cplusplus
}
#endif
#endif
