/*
 * lbmdem -- host driver in C over the C ABI (include/lbmdem_hip.h), a drop-in for the reference
 * binary: `lbmdem <sample.data>` (usage check main.c:1791-1794), same console lines
 * (main.c:614,619,656,1845,1856,1259,1885-1889) and the `final_density:` line on stderr that the
 * reference's JUBE benchmark parses (main.c:1272, benchmark.xml:101).
 *
 * The reference fixes the lattice size and the run length at compile time (-Dlx -Dly,
 * `#define duration 1.5`, main.c:27-32,47); here the same defaults apply and can be overridden at
 * run time: --lx N --ly N --scale S --duration SECONDS --steps N_DEM_STEPS --device K, and (absent in
 * the reference, which cannot resume) --checkpoint FILE (written at the end) / --restart FILE.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <math.h>

#include "../../include/lbmdem_hip.h"

#define DIE(rc, what) do { if ((rc) != LBMDEM_OK) { fprintf(stderr, "%s: %s\n", what, lbmdem_last_error()); return EXIT_FAILURE; } } while (0)

int main(int argc, char** argv) {
  int lx = 7826, ly = 2325, device = 0; /* main.c:27-32 */
  double scale = 1., duration = 1.5;   /* main.c:24-26,47 */
  long max_steps = -1;
  const char* sample = NULL;
  const char *ckpt_out = NULL, *ckpt_in = NULL;
  /* device-resident kernel arguments: ~1.2 us less per launch (the HIP runtime reads this when it
   * initialises, i.e. at the first lbmdem_* call below); an explicit setting of the caller wins */
  setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  printf("2D LBM-DEM code\n");
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--lx") && a + 1 < argc) lx = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--ly") && a + 1 < argc) ly = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--scale") && a + 1 < argc) scale = atof(argv[++a]);
    else if (!strcmp(argv[a], "--duration") && a + 1 < argc) duration = atof(argv[++a]);
    else if (!strcmp(argv[a], "--steps") && a + 1 < argc) max_steps = atol(argv[++a]);
    else if (!strcmp(argv[a], "--device") && a + 1 < argc) device = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--checkpoint") && a + 1 < argc) ckpt_out = argv[++a];
    else if (!strcmp(argv[a], "--restart") && a + 1 < argc) ckpt_in = argv[++a];
    else if (argv[a][0] != '-' && !sample) sample = argv[a];
    else { sample = NULL; break; }
  }
  if (!sample) {
    printf("usage: usage %s <filename> [--lx N --ly N --scale S --duration T --steps N --device K]\n", argv[0]);
    exit(EXIT_FAILURE);
  }
  printf("Opening file : %s\n", sample);

  int n = 0;
  double *r = NULL, *x1 = NULL, *x2 = NULL;
  DIE(lbmdem_read_sample(sample, &n, &r, &x1, &x2), "read_sample");
  printf("Nb grains %d\n", n);
  { /* check_sample, main.c:640-658 */
    double xMax = x1[0], xMin = x1[0], yMax = x2[0], yMin = x2[0], mass = 0.;
    for (int i = 0; i < n; ++i) {
      mass += 2650 * 3.14159265358979 * r[i] * r[i];
      xMax = fmax(xMax, x1[i] + r[i]); xMin = fmin(xMin, x1[i] - r[i]);
      yMax = fmax(yMax, x2[i] + r[i]); yMin = fmin(yMin, x2[i] - r[i]);
    }
    double L0 = xMax - xMin, H0 = yMax - yMin;
    printf("L0=%le H0=%le Mass of Grains=%le Phi=%le\n", L0, H0, mass, mass / (2650 * (L0 * H0)));
  }

  lbmdem_config cfg;
  memset(&cfg, 0, sizeof cfg);
  DIE(lbmdem_physics_defaults(&cfg.phys), "physics_defaults");
  DIE(lbmdem_derive(&cfg, lx, ly, scale, n, r), "derive");
  cfg.x_begin = 0; cfg.x_end = lx; cfg.halo = 0; cfg.device = device;
  printf("no space %le\n", cfg.dx);
  {
    double rMin = r[0];
    for (int i = 1; i < n; ++i) rMin = fmin(rMin, r[i]);
    double dtmax = (1 / cfg.phys.iterDEM) * 3.14159265358979 * rMin * sqrt(3.14159265358979 * 2650 / cfg.phys.kg);
    printf("dtLB=%le,  dtmax=%le,   dt=%le,   npDEM=%d,   c=%lf\n", cfg.dtLB, dtmax, cfg.dt, cfg.npDEM, cfg.c);
  }
  lbmdem_handle* h = NULL;
  long nbsteps = 0;
  if (ckpt_in) {
    DIE(lbmdem_checkpoint_load(ckpt_in, device, &h), "checkpoint_load");
    DIE(lbmdem_get_config(h, &cfg), "get_config");
    nbsteps = lbmdem_nbsteps(h);
    printf("Restarted from %s at step %ld\n", ckpt_in, nbsteps);
  } else {
    DIE(lbmdem_create(&cfg, r, x1, x2, &h), "create");
  }
  time_t now = time(NULL);
  printf("Current local time and date: %s", asctime(localtime(&now)));
  if (!ckpt_in) { /* stats.data header, main.c:1867-1877 */
    FILE* st = fopen("stats.data", "w");
    if (st) {
      fprintf(st, "#1_t 2_xfront 3_xgrainmax 4_height 5_zmean 6_energie_x 7_energie_y "
                  "8_energie_teta 9_energie_cin 10_N0 11_N1 12_N2 13_N3 14_N4 15_N5 "
                  "16_energy_Potential 17_Strain_Energy 18_Frictional_Work "
                  "19_Internal_Friction 20_Inelastic_Collision 21_Slip "
                  "22_Rotational_Work\n");
      fclose(st);
    }
  }

  /* main loop, main.c:1879-1890: advance to the next console cadence (updateVerlet steps) at a time */
  struct timespec t0, t1;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  int nFile = (int)(nbsteps / cfg.phys.stepFilm); /* main.c:147 */
  double energies[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* KE, PE, SE, IFR, WF, INCE, TSLIP, TRW of the last write_DEM (main.c:1885-1889) */
  const int chunk = cfg.phys.updateVerlet;
  const int stepConsole = 400; /* main.c:140 */
  int stop = 0;
  do {
    long todo = chunk - (nbsteps % chunk);
    if (max_steps >= 0 && nbsteps + todo > max_steps) todo = max_steps - nbsteps;
    if (todo <= 0) break;
    /* check_density cadence (main.c:1715): printed right after the fluid step of such a DEM step */
    for (long k = 0; k < todo; ++k) {
      int lbm_now = (nbsteps % cfg.npDEM == 0), console_now = (nbsteps % stepConsole == 0);
      if (lbm_now && console_now) {
        DIE(lbmdem_lbm_step(h), "lbm_step");
        double sum = 0;
        DIE(lbmdem_total_density(h, &sum), "total_density");
        printf("Iteration Number %ld, Total density in the system %f\n", nbsteps, sum);
        if (nbsteps % cfg.phys.updateVerlet == 0) DIE(lbmdem_verlet_rebuild(h), "verlet_rebuild");
        DIE(lbmdem_dem_substep(h), "dem_substep");
      } else {
        DIE(lbmdem_run(h, 1), "run");
      }
      ++nbsteps;
      /* output cadence of renderScene (main.c:1767-1772): write_vtk every stepFilm DEM steps */
      if (nbsteps % cfg.phys.stepFilm == 0) {
        DIE(lbmdem_write_vtk(h, ".", nFile), "write_vtk");
        nFile++;
      }
      /* write_DEM and write_forces every stepStrob = 4000 DEM steps (main.c:142,1773-1776) */
      if (nbsteps % 4000 == 0) {
        DIE(lbmdem_write_dem(h, ".", nFile, energies), "write_dem");
        DIE(lbmdem_write_forces(h, ".", nFile), "write_forces");
      }
      /* the reference tests its stop condition after EVERY renderScene() (main.c:1880-1890) */
      if (nbsteps * cfg.dt > duration) { stop = 1; break; }
    }
    if (nbsteps % chunk == 0) {
      now = time(NULL);
      printf("steps %li steps %le KE %le PE %le SE %le WF %le INCE %le SLIP %le RW %le Time %s \n", nbsteps,
             nbsteps * cfg.dt, energies[0], energies[1], energies[2], energies[4], energies[5], energies[6], energies[7], asctime(localtime(&now)));
    }
  } while (!stop && (max_steps < 0 || nbsteps < max_steps));
  DIE(lbmdem_sync(h), "sync");
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (ckpt_out) DIE(lbmdem_checkpoint_save(h, ckpt_out), "checkpoint_save");
  double sum = 0;
  DIE(lbmdem_total_density(h, &sum), "total_density");
  fprintf(stderr, "final_density: %f\n", sum);
  double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  long lbm_steps = (nbsteps + cfg.npDEM - 1) / cfg.npDEM;
  fprintf(stderr, "time: %e\n", secs);
  fprintf(stderr, "dem_steps: %ld\n", nbsteps);
  fprintf(stderr, "MLUPS: %.1f  DEM-steps/s: %.1f\n", 1e-6 * (double)lx * ly * lbm_steps / secs, nbsteps / secs);
  now = time(NULL);
  printf("End local time and date: %s", asctime(localtime(&now)));
  lbmdem_destroy(h);
  lbmdem_free_host(r); lbmdem_free_host(x1); lbmdem_free_host(x2);
  return 0;
}
