// lbmdem_output.hip -- the reference's file outputs over the C ABI: write_vtk (main.c:237-338 + visit_writer's binary
// rectilinear path), write_DEM (main.c:340-438: DEM%06d.dat, stats.data), write_forces (main.c:440-478), the per-grain
// diagnostics table they print, and the merged-strip forms of the VTK writer.

#include "lbmdem_handle.h"

#pragma GCC visibility push(default)
extern "C" {
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

int lbmdem_write_vtk_file(const char* path, int nx, int ny, const char* name, int dim, const float* data) {
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
    rc = lbmdem_write_vtk_file(path, L.lx, L.ly, names[k], dims[k], data[k]);
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

// a strip's five fields (block11: the layout of lbmdem_download_vtk_fields for nx columns, back to back) into the
// lattice-sized arrays, columns [x0, x0 + nx)
int lbmdem_vtk_place_block(float* fields11, int lx, int ly, int x0, int nx, const float* block11) {
  const size_t part = (size_t)nx * ly, cnt = (size_t)lx * ly;
  const float* lp[5] = {block11, block11 + part, block11 + 4 * part, block11 + 7 * part, block11 + 8 * part};
  float* fp[5] = {fields11, fields11 + cnt, fields11 + 4 * cnt, fields11 + 7 * cnt, fields11 + 8 * cnt};
  const int dims[5] = {1, 3, 3, 1, 3};
  for (int k = 0; k < 5; ++k)
    for (int y = 0; y < ly; ++y)
      memcpy(fp[k] + ((size_t)y * lx + x0) * dims[k], lp[k] + (size_t)y * nx * dims[k], sizeof(float) * nx * dims[k]);
  return LBMDEM_OK;
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
  (void)cnt;
  lbmdem_vtk_place_block(fields11, L.lx, L.ly, x0, nx, loc.data());
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}


int lbmdem_write_vtk_fields(const char* dir, int nfile, int lx, int ly, const float* fields11) {
  if (!fields11 || lx < 2 || ly < 2) return fail(LBMDEM_EINVAL, "bad lbmdem_write_vtk_fields arguments");
  const size_t cnt = (size_t)lx * ly;
  const char* names[5] = {"grain_pressure", "grain_velocity", "grain_acceleration", "fluid_pressure", "fluid_velocity"};
  const int dims[5] = {1, 3, 3, 1, 3};
  const float* data[5] = {fields11, fields11 + cnt, fields11 + 4 * cnt, fields11 + 7 * cnt, fields11 + 8 * cnt};
  for (int k = 0; k < 5; ++k) {
    char path[4096];
    snprintf(path, sizeof path, "%s/%s_%.6i.vtk", (dir && *dir) ? dir : ".", names[k], nfile);  // main.c:241-249
    const int rc = lbmdem_write_vtk_file(path, lx, ly, names[k], dims[k], data[k]);
    if (rc != LBMDEM_OK) return rc;
  }
  return LBMDEM_OK;
}

}  // extern "C"
#pragma GCC visibility pop
