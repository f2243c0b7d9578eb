/* sanitize_check.c -- one ASAN/UBSAN run of the CPU oracle (test infrastructure; SURVEY.md section 5: the reference has
 * no sanitizer configuration at all). Built by tests/test_sanitizers.py as
 *     gcc -std=gnu99 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all oracle/sanitize_check.c oracle/lbmdem_oracle.c -lm
 * and run on a packing that exercises every routine: sample reader, obstacle map with grains clipped at all four lattice
 * edges and beyond the lattice, reinit, collide, edges, interpolated bounce-back incl. the order hazard, swap + stream,
 * hydrodynamic forces, Verlet lists, film and regular contact laws, all wall laws, the serial total density. */
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include "lbmdem_oracle.h"

int main(int argc, char** argv) {
  const int lx = 96, ly = 72;
  if (argc < 2) { fprintf(stderr, "usage: %s <sample.data>\n", argv[0]); return 2; }
  int n = 0;
  double *r = NULL, *x1 = NULL, *x2 = NULL;
  if (ora_read_sample(argv[1], &n, &r, &x1, &x2) != 0 || n < 1) { fprintf(stderr, "cannot read %s\n", argv[1]); return 2; }
  ora_sim* s = ora_create(lx, ly, 1.0, n, r, x1, x2);
  if (!s) return 3;
  double* k = calloc((size_t)n * 9, sizeof(double));
  for (int i = 0; i < n; ++i) {
    k[9 * i] = x1[i]; k[9 * i + 1] = x2[i];
    k[9 * i + 3] = 0.03 * sin(1.7 * i); k[9 * i + 4] = -0.02 * cos(0.9 * i); k[9 * i + 5] = 15.0 * sin(0.3 * i);
  }
  ora_set_kinematics(s, k);
  ora_steps(s, 130);                 /* film step 0, Verlet rebuilds at 0 and 100, ~11-13 fluid steps */
  const double mass = ora_total_density(s);
  double* g = malloc(sizeof(double) * (size_t)n * ORA_GRAIN_COLS);
  ora_get_grains(s, g);
  double* fh = malloc(sizeof(double) * 3 * (size_t)n);
  ora_get_fhf(s, fh);
  int bad = !(isfinite(mass) && mass > 0.0);
  for (int i = 0; i < n * ORA_GRAIN_COLS; ++i) if (!isfinite(g[i])) bad = 1;
  printf("sanitize_check: %d grains, %ld steps, total density %.6f%s\n", n, ora_nbsteps(s), mass, bad ? "  BAD" : "");
  free(g); free(fh); free(k);
  ora_destroy(s);
  ora_free(r); ora_free(x1); ora_free(x2);
  return bad;
}
