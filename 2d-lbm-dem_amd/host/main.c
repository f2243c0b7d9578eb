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
 * --dry: the reference compiled without `#define _FLUIDE_` (main.c:16) -- DEM only: no fluid step and no check_density
 * line (main.c:1709-1719), no VTK frames although nFile still advances (main.c:1767-1772), hydrodynamic forces 0.
 * --gpus N: one process per GPU, rank k on device K + k; --devices a,b,c names the device of every rank instead (the
 * same device may appear twice: that is how the tests run several ranks on a one-GPU box, see tests/rccl_shim).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <math.h>
#include <unistd.h>
#include <sys/wait.h>
#include <signal.h>

#include "../../include/lbmdem_hip.h"

#define DIE(rc, what) do { if ((rc) != LBMDEM_OK) { fprintf(stderr, "%s: %s\n", what, lbmdem_last_error()); return EXIT_FAILURE; } } while (0)

/* --gpus N: one process per GPU (forked before anything touches the HIP runtime), x-strips with the grains
 * distributed, neighbour messages over RCCL (lbmdem_comm_*). Rank 0 creates the RCCL id and hands it to the others
 * through a file in a private temporary directory. Rank 0 prints and writes the VTK frames and DEM tables (merged over the ranks); checkpoints are single-GPU. */
static int g_rank = 0, g_world = 1, g_use_comm = 0, g_dry = 0;
static char g_iddir[256] = "";
static double g_comm_timeout = 180.;   /* seconds the ranks' transport may take to come up (--comm-timeout, LBMDEM_COMM_TIMEOUT) */
static int g_devices[64], g_ndevices = 0;   /* --devices */

static int share_id(unsigned char* id) {
  char path[320], tmp[340];
  snprintf(path, sizeof path, "%s/rccl_id", g_iddir);
  if (g_rank == 0) {
    if (lbmdem_comm_unique_id(id) != LBMDEM_OK) return -1;
    if (g_world == 1) return 0;
    snprintf(tmp, sizeof tmp, "%s.tmp", path);
    FILE* fp = fopen(tmp, "wb");
    if (!fp || fwrite(id, 1, LBMDEM_COMM_ID_BYTES, fp) != LBMDEM_COMM_ID_BYTES) return -1;
    fclose(fp);
    return rename(tmp, path);
  }
  for (int tries = 0; tries < 6000; ++tries) { /* up to 60 s */
    FILE* fp = fopen(path, "rb");
    if (fp) {
      size_t got = fread(id, 1, LBMDEM_COMM_ID_BYTES, fp);
      fclose(fp);
      if (got == LBMDEM_COMM_ID_BYTES) return 0;
    }
    usleep(10000);
  }
  return -1;
}

static int run(int argc, char** argv);

static int check_decomposition(int argc, char** argv, int gpus) {
  int lx = 7826, ly = 2325;
  double scale = 1.;
  const char* sample = NULL;
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--lx") && a + 1 < argc) lx = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--ly") && a + 1 < argc) ly = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--scale") && a + 1 < argc) scale = atof(argv[++a]);
    else if (argv[a][0] == '-' && argv[a][1] == '-' && strcmp(argv[a], "--comm") && strcmp(argv[a], "--dry") && a + 1 < argc) ++a;
    else if (argv[a][0] != '-' && !sample) sample = argv[a];
  }
  if (!sample) return 0;   /* run() prints the usage line */
  int n = 0;
  double *r = NULL, *x1 = NULL, *x2 = NULL;
  if (lbmdem_read_sample(sample, &n, &r, &x1, &x2) != LBMDEM_OK) { fprintf(stderr, "read_sample: %s\n", lbmdem_last_error()); return EXIT_FAILURE; }
  lbmdem_config cfg;
  memset(&cfg, 0, sizeof cfg);
  lbmdem_physics_defaults(&cfg.phys);
  if (lbmdem_derive(&cfg, lx, ly, scale, n, r) != LBMDEM_OK) { fprintf(stderr, "derive: %s\n", lbmdem_last_error()); return EXIT_FAILURE; }
  double rmax = r[0];
  for (int i = 1; i < n; ++i) rmax = fmax(rmax, r[i]);
  lbmdem_free_host(r); lbmdem_free_host(x1); lbmdem_free_host(x2);
  const int margin = lbmdem_dist_margin_for(&cfg, rmax);
  int narrowest = lx;
  for (int k = 0; k < gpus; ++k) {
    int w = (int)((long)(k + 1) * lx / gpus) - (int)((long)k * lx / gpus);
    if (w < narrowest) narrowest = w;
  }
  if (narrowest < margin) {
    fprintf(stderr, "--gpus %d: strips of %d rows are narrower than the margin of %d rows this packing needs "
                    "(npDEM = %d sub-steps per fluid step); use at most %d GPUs for a lattice %d rows long\n",
            gpus, narrowest, margin, cfg.npDEM, lx / margin, lx);
    return EXIT_FAILURE;
  }
  return 0;
}

int main(int argc, char** argv) {
  int gpus = 1;
  if (getenv("LBMDEM_COMM_TIMEOUT")) g_comm_timeout = atof(getenv("LBMDEM_COMM_TIMEOUT"));
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--gpus") && a + 1 < argc) gpus = atoi(argv[a + 1]);
    if (!strcmp(argv[a], "--comm")) g_use_comm = 1;   /* the RCCL path with a single rank */
    if (!strcmp(argv[a], "--dry")) g_dry = 1;
    if (!strcmp(argv[a], "--comm-timeout") && a + 1 < argc) g_comm_timeout = atof(argv[a + 1]);
    if (!strcmp(argv[a], "--devices") && a + 1 < argc) {
      for (const char* p = argv[a + 1]; *p && g_ndevices < 64;) {
        char* end = NULL;
        long v = strtol(p, &end, 10);
        if (end == p || v < 0) { fprintf(stderr, "--devices: a comma-separated list of device ordinals\n"); return EXIT_FAILURE; }
        g_devices[g_ndevices++] = (int)v;
        p = (*end == ',') ? end + 1 : end;
        if (*end && *end != ',') { fprintf(stderr, "--devices: a comma-separated list of device ordinals\n"); return EXIT_FAILURE; }
      }
    }
  }
  if (g_ndevices > 0 && g_ndevices < gpus) { fprintf(stderr, "--devices names %d devices for %d ranks\n", g_ndevices, gpus); return EXIT_FAILURE; }
  if (g_dry && (gpus > 1 || g_use_comm)) { fprintf(stderr, "--dry is a single-GPU mode (the strips exist for the fluid)\n"); return EXIT_FAILURE; }
  if (gpus <= 1) return run(argc, argv);
  g_world = gpus; g_use_comm = 1;
  { /* every strip must be at least one margin wide (lbmdem_dist_enable would refuse on the ranks whose strip is one
     * row narrower, and the others would wait for them): checked here, on the host, before anything is forked */
    int rc = check_decomposition(argc, argv, gpus);
    if (rc != 0) return rc;
  }
  snprintf(g_iddir, sizeof g_iddir, "/tmp/lbmdem_XXXXXX");
  if (!mkdtemp(g_iddir)) { perror("mkdtemp"); return EXIT_FAILURE; }
  pid_t pids[64];
  if (gpus > 64) { fprintf(stderr, "--gpus: at most 64\n"); return EXIT_FAILURE; }
  for (int r = 0; r < gpus; ++r) {
    pids[r] = fork();
    if (pids[r] < 0) { perror("fork"); return EXIT_FAILURE; }
    if (pids[r] == 0) { g_rank = r; _exit(run(argc, argv)); }
  }
  /* the first rank that fails takes the others with it: its peers would otherwise block for ever inside an RCCL call */
  int bad = 0, left = gpus;
  /* ... and so does a transport that never comes up (a rank stuck in communicator creation or in its first exchange
   * with a neighbour cannot report anything): every rank leaves a file once lbmdem_comm_selftest has passed; ranks that
   * have not all done so after --comm-timeout seconds (default 180; LBMDEM_COMM_TIMEOUT) are killed with a message
   * instead of hanging the job. bench.py guards its C driver the same way (a watchdogged trial). */
  {
    const double limit = g_comm_timeout;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    int up = 0;
    while (left > 0 && !bad && !up && limit > 0) {
      int st = 0;
      pid_t p = waitpid(-1, &st, WNOHANG);
      if (p > 0) {
        --left;
        for (int r = 0; r < gpus; ++r) if (pids[r] == p) pids[r] = 0;
        if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
          for (int r = 0; r < gpus; ++r) if (pids[r] > 0) kill(pids[r], SIGKILL);
          bad = 1;
        }
        continue;
      }
      up = 1;
      for (int r = 0; r < gpus && up; ++r) {
        char f[320];
        snprintf(f, sizeof f, "%s/up.%d", g_iddir, r);
        if (access(f, F_OK) != 0) up = 0;
      }
      if (up) break;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec) > limit) {
        fprintf(stderr, "--gpus %d: the ranks' transport (RCCL communicators + a first exchange with both neighbours) did not "
                        "come up within %.0f s: stopping all ranks (--comm-timeout S / LBMDEM_COMM_TIMEOUT to wait longer)\n",
                gpus, limit);
        for (int r = 0; r < gpus; ++r) if (pids[r] > 0) kill(pids[r], SIGKILL);
        bad = 1;
        break;
      }
      struct timespec nap = {0, 20 * 1000 * 1000};
      nanosleep(&nap, NULL);
    }
  }
  while (left > 0) {
    int st = 0;
    pid_t p = waitpid(-1, &st, 0);
    if (p < 0) break;
    --left;
    for (int r = 0; r < gpus; ++r) if (pids[r] == p) pids[r] = 0;
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) {
      if (!bad) for (int r = 0; r < gpus; ++r) if (pids[r] > 0) kill(pids[r], SIGKILL);
      bad = 1;
    }
  }
  char path[320];
  for (int r = 0; r < gpus; ++r) { snprintf(path, sizeof path, "%s/up.%d", g_iddir, r); unlink(path); }
  snprintf(path, sizeof path, "%s/rccl_id", g_iddir); unlink(path); rmdir(g_iddir);
  return bad ? EXIT_FAILURE : 0;
}

#define SAY(...) do { if (g_rank == 0) printf(__VA_ARGS__); } while (0)

/* check_density / final_density with the reference's own bits (main.c:1249-1273): ONE serial chain over the whole
 * lattice. With several strips the chain runs through the ranks in x order: rank r continues from rank r-1's sum
 * (handed on through the all-reduce: everybody else contributes 0). Every rank returns the lattice's sum. */
static double serial_density(lbmdem_handle* h, lbmdem_comm* comm) {
  double s = 0.;
  for (int r = 0; r < g_world; ++r) {
    double v = 0.;
    if (r == g_rank && lbmdem_total_density_serial(h, s, &v, NULL) != LBMDEM_OK) {
      fprintf(stderr, "total_density_serial: %s\n", lbmdem_last_error());
      exit(EXIT_FAILURE);
    }
    if (comm && lbmdem_comm_allreduce_sum(comm, &v, 1) != LBMDEM_OK) {
      fprintf(stderr, "allreduce: %s\n", lbmdem_last_error());
      exit(EXIT_FAILURE);
    }
    s = v;
  }
  return s;
}

static int run(int argc, char** argv) {
  int lx = 7826, ly = 2325, device = 0; /* main.c:27-32 */
  double scale = 1., duration = 1.5;   /* main.c:24-26,47 */
  long max_steps = -1;
  const char* sample = NULL;
  const char *ckpt_out = NULL, *ckpt_in = NULL;
  /* device-resident kernel arguments: ~1.2 us less per launch (the HIP runtime reads this when it
   * initialises, i.e. at the first lbmdem_* call below); an explicit setting of the caller wins */
  setenv("HIP_FORCE_DEV_KERNARG", "1", 0);
  SAY("2D LBM-DEM code\n");
  for (int a = 1; a < argc; ++a) {
    if (!strcmp(argv[a], "--lx") && a + 1 < argc) lx = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--ly") && a + 1 < argc) ly = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--scale") && a + 1 < argc) scale = atof(argv[++a]);
    else if (!strcmp(argv[a], "--duration") && a + 1 < argc) duration = atof(argv[++a]);
    else if (!strcmp(argv[a], "--steps") && a + 1 < argc) max_steps = atol(argv[++a]);
    else if (!strcmp(argv[a], "--device") && a + 1 < argc) device = atoi(argv[++a]);
    else if (!strcmp(argv[a], "--checkpoint") && a + 1 < argc) ckpt_out = argv[++a];
    else if (!strcmp(argv[a], "--restart") && a + 1 < argc) ckpt_in = argv[++a];
    else if (!strcmp(argv[a], "--gpus") && a + 1 < argc) ++a;
    else if (!strcmp(argv[a], "--comm-timeout") && a + 1 < argc) ++a;
    else if (!strcmp(argv[a], "--devices") && a + 1 < argc) ++a;
    else if (!strcmp(argv[a], "--comm")) {}
    else if (!strcmp(argv[a], "--dry")) {}
    else if (argv[a][0] != '-' && !sample) sample = argv[a];
    else { sample = NULL; break; }
  }
  if (!sample) {
    SAY("usage: usage %s <filename> [--lx N --ly N --scale S --duration T --steps N --device K --gpus N --devices a,b,.. --comm-timeout S --dry]\n", argv[0]);
    exit(EXIT_FAILURE);
  }
  SAY("Opening file : %s\n", sample);

  int n = 0;
  double *r = NULL, *x1 = NULL, *x2 = NULL;
  DIE(lbmdem_read_sample(sample, &n, &r, &x1, &x2), "read_sample");
  SAY("Nb grains %d\n", n);
  { /* check_sample, main.c:640-658 */
    double xMax = x1[0], xMin = x1[0], yMax = x2[0], yMin = x2[0], mass = 0.;
    for (int i = 0; i < n; ++i) {
      mass += 2650 * 3.14159265358979 * r[i] * r[i];
      xMax = fmax(xMax, x1[i] + r[i]); xMin = fmin(xMin, x1[i] - r[i]);
      yMax = fmax(yMax, x2[i] + r[i]); yMin = fmin(yMin, x2[i] - r[i]);
    }
    double L0 = xMax - xMin, H0 = yMax - yMin;
    SAY("L0=%le H0=%le Mass of Grains=%le Phi=%le\n", L0, H0, mass, mass / (2650 * (L0 * H0)));
  }

  lbmdem_config cfg;
  memset(&cfg, 0, sizeof cfg);
  DIE(lbmdem_physics_defaults(&cfg.phys), "physics_defaults");
  DIE(lbmdem_derive(&cfg, lx, ly, scale, n, r), "derive");
  cfg.x_begin = 0; cfg.x_end = lx; cfg.halo = 0; cfg.device = device;
  if (g_use_comm) { /* this rank's strip */
    cfg.x_begin = (int)((long)g_rank * lx / g_world);
    cfg.x_end = (int)((long)(g_rank + 1) * lx / g_world);
    cfg.halo = g_world > 1 ? 2 : 0;
    cfg.device = g_ndevices > 0 ? g_devices[g_rank] : device + g_rank;
  }
  /* with several strips every rank keeps its own checkpoint file: FILE.rank<k> */
  char ckpt_in_rank[4096], ckpt_out_rank[4096];
  if (g_use_comm && ckpt_in) { snprintf(ckpt_in_rank, sizeof ckpt_in_rank, "%s.rank%d", ckpt_in, g_rank); ckpt_in = ckpt_in_rank; }
  if (g_use_comm && ckpt_out) { snprintf(ckpt_out_rank, sizeof ckpt_out_rank, "%s.rank%d", ckpt_out, g_rank); ckpt_out = ckpt_out_rank; }
  SAY("no space %le\n", cfg.dx);
  {
    double rMin = r[0];
    for (int i = 1; i < n; ++i) rMin = fmin(rMin, r[i]);
    double dtmax = (1 / cfg.phys.iterDEM) * 3.14159265358979 * rMin * sqrt(3.14159265358979 * 2650 / cfg.phys.kg);
    SAY("dtLB=%le,  dtmax=%le,   dt=%le,   npDEM=%d,   c=%lf\n", cfg.dtLB, dtmax, cfg.dt, cfg.npDEM, cfg.c);
  }
  lbmdem_handle* h = NULL;
  long nbsteps = 0;
  if (ckpt_in) {
    DIE(lbmdem_checkpoint_load(ckpt_in, cfg.device, &h), "checkpoint_load");
    DIE(lbmdem_get_config(h, &cfg), "get_config");
    if (g_use_comm && (cfg.x_begin != (int)((long)g_rank * cfg.lx / g_world) || cfg.x_end != (int)((long)(g_rank + 1) * cfg.lx / g_world))) {
      fprintf(stderr, "%s holds rows [%d, %d): written by a run with another number of strips\n", ckpt_in, cfg.x_begin, cfg.x_end);
      return EXIT_FAILURE;
    }
    nbsteps = lbmdem_nbsteps(h);
    SAY("Restarted from %s at step %ld\n", ckpt_in, nbsteps);
  } else {
    DIE(lbmdem_create(&cfg, r, x1, x2, &h), "create");
  }
  lbmdem_comm* comm = NULL;
  if (g_use_comm) {
    unsigned char id[LBMDEM_COMM_ID_BYTES];
    if (share_id(id) != 0) { fprintf(stderr, "rank %d: no RCCL id: %s\n", g_rank, lbmdem_last_error()); return EXIT_FAILURE; }
    if (!ckpt_in) DIE(lbmdem_dist_enable(h, 0), "dist_enable");   /* a restarted strip comes back distributed */
    DIE(lbmdem_comm_create(id, g_rank, g_world, cfg.device, &comm), "comm_create");
    DIE(lbmdem_comm_selftest(comm, 4096), "comm_selftest");   /* one rank: to itself; several: with both neighbours */
    if (g_iddir[0]) {   /* tell the parent that this rank's transport works (its start-up watchdog) */
      char f[320];
      snprintf(f, sizeof f, "%s/up.%d", g_iddir, g_rank);
      FILE* u = fopen(f, "w");
      if (u) fclose(u);
    }
  }
  time_t now = time(NULL);
  SAY("Current local time and date: %s", asctime(localtime(&now)));
  if (!ckpt_in && g_rank == 0) { /* stats.data header, main.c:1867-1877 */
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
      if (g_dry) {   /* renderScene without its `#ifdef _FLUIDE_` block (main.c:1709-1719) */
        DIE(lbmdem_run_dem(h, 1), "run_dem");
      } else if (lbm_now && console_now) {
        if (comm) DIE(lbmdem_comm_lbm_step(h, comm), "comm_lbm_step"); else DIE(lbmdem_lbm_step(h), "lbm_step");
        double sum = serial_density(h, comm);
        SAY("Iteration Number %ld, Total density in the system %f\n", nbsteps, sum);
        if (nbsteps % cfg.phys.updateVerlet == 0) DIE(lbmdem_verlet_rebuild(h), "verlet_rebuild");
        DIE(lbmdem_dem_substep(h), "dem_substep");
      } else if (comm) {
        DIE(lbmdem_comm_run(h, comm, 1), "comm_run");
      } else {
        DIE(lbmdem_run(h, 1), "run");
      }
      ++nbsteps;
      /* output cadence of renderScene (main.c:1767-1772): write_vtk every stepFilm DEM steps. With several strips
       * the columns are merged over the ranks and rank 0 writes the same five files */
      if (nbsteps % cfg.phys.stepFilm == 0) {
        if (g_dry) {}   /* write_vtk sits inside `#ifdef _FLUIDE_` (main.c:1768-1770); nFile++ does not */
        else if (comm) DIE(lbmdem_comm_write_vtk(h, comm, ".", nFile), "comm_write_vtk");
        else DIE(lbmdem_write_vtk(h, ".", nFile), "write_vtk");
        nFile++;
      }
      /* write_DEM and write_forces every stepStrob = 4000 DEM steps (main.c:142,1773-1776). With several strips the
       * sub-step before was run by rank 0 on a full replica (lbmdem_comm_run): it holds the whole table */
      if (nbsteps % 4000 == 0 && g_rank == 0) {
        DIE(lbmdem_write_dem(h, ".", nFile, energies), "write_dem");
        DIE(lbmdem_write_forces(h, ".", nFile), "write_forces");
      }
      /* the reference tests its stop condition after EVERY renderScene() (main.c:1880-1890) */
      if (nbsteps * cfg.dt > duration) { stop = 1; break; }
    }
    if (nbsteps % chunk == 0) {
      now = time(NULL);
      SAY("steps %li steps %le KE %le PE %le SE %le WF %le INCE %le SLIP %le RW %le Time %s \n", nbsteps,
             nbsteps * cfg.dt, energies[0], energies[1], energies[2], energies[4], energies[5], energies[6], energies[7], asctime(localtime(&now)));
    }
  } while (!stop && (max_steps < 0 || nbsteps < max_steps));
  DIE(lbmdem_sync(h), "sync");
  clock_gettime(CLOCK_MONOTONIC, &t1);
  if (ckpt_out) {
    if (comm) DIE(lbmdem_comm_sync_carries(h, comm), "comm_sync_carries");
    DIE(lbmdem_checkpoint_save(h, ckpt_out), "checkpoint_save");
  }
  double sum = serial_density(h, comm);
  double secs = (t1.tv_sec - t0.tv_sec) + 1e-9 * (t1.tv_nsec - t0.tv_nsec);
  if (comm) { /* the slowest rank's time */
    double tmax[64] = {0};
    tmax[g_rank] = secs;
    DIE(lbmdem_comm_allreduce_sum(comm, tmax, g_world), "allreduce");
    for (int r = 0; r < g_world; ++r) if (tmax[r] > secs) secs = tmax[r];
  }
  long lbm_steps = g_dry ? 0 : (nbsteps + cfg.npDEM - 1) / cfg.npDEM;
  if (g_rank == 0) {
    fprintf(stderr, "final_density: %f\n", sum);
    fprintf(stderr, "time: %e\n", secs);
    fprintf(stderr, "dem_steps: %ld\n", nbsteps);
    fprintf(stderr, "MLUPS: %.1f  DEM-steps/s: %.1f  (%d GPU%s)\n", 1e-6 * (double)lx * ly * lbm_steps / secs, nbsteps / secs,
            g_world, g_world > 1 ? "s" : "");
  }
  now = time(NULL);
  SAY("End local time and date: %s", asctime(localtime(&now)));
  lbmdem_comm_destroy(comm);
  lbmdem_destroy(h);
  lbmdem_free_host(r); lbmdem_free_host(x1); lbmdem_free_host(x2);
  return 0;
}
