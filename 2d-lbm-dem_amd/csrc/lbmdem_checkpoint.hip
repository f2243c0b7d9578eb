// lbmdem_checkpoint.hip -- checkpoint / restart of a handle (the reference cannot resume a run): single domain, or one
// file per rank of a strip decomposition with distributed grains.

#include "lbmdem_handle.h"

#include <sys/stat.h>

#pragma GCC visibility push(default)
extern "C" {
// ---- checkpoint / restart ------------------------------------------------------------------------

namespace {
struct CkptHeader {
  char magic[8];       // "LBMDEMC5"
  double lid6;         // lbmdem_set_lid
  int layout;          // device layout of the populations in the file: 1 = 16-node tiles f[x][y/16][q][y%16]
  int force_mode, diag_always, has_carry;
  double carry[3];     // pft, pff, pf of the order-dependent contact diagnostics (main.c:130-131), when has_carry
  lbmdem_config cfg;   // incl. the wall positions VerletWall may have moved
  long nbsteps;
  int verlet_ok, nnbr; // symmetric list length
  long plane;          // sanity: nxl * sy of the writer
  int has_dist, pad;   // a CkptDist section follows the lattice (the writer had its grains distributed over strips)
};
struct CkptDist {       // follows the lattice when the writer had its grains distributed over strips
  char magic[8];        // "LBMDIST1"
  int margin, cap_g, cap_t, cap_l, poison, pad;
};
constexpr int CKPT_LAYOUT = 1;
static bool wr(FILE* fp, const void* p, size_t n) { return fwrite(p, 1, n, fp) == n; }
static bool rd(FILE* fp, void* p, size_t n) { return fread(p, 1, n, fp) == n; }
struct FileCloser {   // closes on every exit path, exceptions included
  FILE* fp;
  ~FileCloser() { if (fp) fclose(fp); }
};
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
  memcpy(H.magic, "LBMDEMC5", 8);
  H.lid6 = h->L.lid6;
  H.layout = CKPT_LAYOUT; H.force_mode = h->force_mode; H.diag_always = h->diag_always ? 1 : 0;
  H.has_carry = 1;
  if (!h->dist && h->carry_from < h->substep_seq) {
    launch_carry_resolve(h->ct, h->carry_from, h->stream);
    HIP_TRY(hipStreamSynchronize(h->stream));
  }
  HIP_TRY(hipMemcpy(H.carry, h->ct.carry, sizeof H.carry, hipMemcpyDeviceToHost));
  H.cfg = h->cfg; H.nbsteps = h->nbsteps; H.verlet_ok = h->verlet_ok ? 1 : 0; H.nnbr = off[n]; H.plane = h->L.plane;
  H.has_dist = h->dist ? 1 : 0;
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
  FileCloser closer{fp};
  CkptHeader H;
  if (!rd(fp, &H, sizeof H) || memcmp(H.magic, "LBMDEMC5", 8) != 0) return fail(LBMDEM_EINVAL, "'%s' is not a checkpoint of this library version", path);
  if (H.layout != CKPT_LAYOUT) return fail(LBMDEM_EINVAL, "checkpoint '%s' holds another device layout (%d)", path, H.layout);
  // the header is not trusted: sizes are checked before anything is allocated from them
  const lbmdem_config& hc = H.cfg;
  if (hc.nbgrains < 1 || hc.nbgrains > (1 << 28) || hc.lx < 3 || hc.ly < 3 || hc.lx > (1 << 24) || hc.ly > (1 << 24) ||
      hc.x_begin < 0 || hc.x_end > hc.lx || hc.x_begin >= hc.x_end || hc.halo < 0 || hc.halo > hc.lx || hc.npDEM < 1 ||
      H.nnbr < 0 || H.nbsteps < 0 || H.plane < 1 || (H.has_dist != 0 && H.has_dist != 1))
    return fail(LBMDEM_EINVAL, "checkpoint '%s': implausible header (grains %d, lattice %d x %d, rows [%d, %d))", path,
                hc.nbgrains, hc.lx, hc.ly, hc.x_begin, hc.x_end);
  const int n = H.cfg.nbgrains;
  {   // nothing is allocated from a header the file itself cannot back: the fixed part alone is this long
    struct stat stt{};
    const long long need = (long long)sizeof H + (long long)sizeof(double) * 14 * n + (long long)sizeof(int) * (n + 1) +
                           (long long)sizeof(int) * H.nnbr + n + (long long)(sizeof(int) + 9 * sizeof(double)) * H.plane;
    if (fstat(fileno(fp), &stt) != 0) return fail(LBMDEM_EINVAL, "checkpoint '%s': cannot take its size (at least %lld bytes needed)", path, need);
    if ((long long)stt.st_size < need)
      return fail(LBMDEM_EINVAL, "checkpoint '%s' is shorter than its header claims (%lld of at least %lld bytes)", path,
                  (long long)stt.st_size, need);
  }
  std::vector<double> r(n), kin(9 * (size_t)n);
  if (!rd(fp, r.data(), sizeof(double) * n) || !rd(fp, kin.data(), sizeof(double) * 9 * n)) return fail(LBMDEM_EINVAL, "checkpoint truncated");
  lbmdem_config cfg = H.cfg;
  cfg.device = device;
  lbmdem_handle* h = nullptr;
  int rc = lbmdem_create(&cfg, r.data(), kin.data(), kin.data() + n, &h);  // x1, x2 are the first two columns
  if (rc != LBMDEM_OK) return rc;
  struct HandleGuard {   // whatever leaves this function early -- a return, a bad_alloc in a staging buffer -- frees the handle
    lbmdem_handle* h;
    ~HandleGuard() { if (h) lbmdem_destroy(h); }
  } guard{h};
  bool ok = h->L.plane == H.plane && H.nnbr <= h->V.cap;
  auto fill = [&](void* dev, size_t bytes) {
    if (!ok || bytes == 0) return;
    std::vector<char> buf(bytes);
    ok = rd(fp, buf.data(), bytes) && hipMemcpy(dev, buf.data(), bytes, hipMemcpyHostToDevice) == hipSuccess;
  };
  if (ok) ok = hipMemcpy(h->kin[0].x1, kin.data(), sizeof(double) * 9 * n, hipMemcpyHostToDevice) == hipSuccess;
  h->kcur = 0;
  fill(h->fhf, sizeof(double) * 3 * n);
  fill(h->gp, sizeof(double) * n);
  {   // the pair list goes straight into kernels that index with it: checked here, on the host
    std::vector<int> off((size_t)n + 1), nb((size_t)H.nnbr);
    if (ok) ok = rd(fp, off.data(), sizeof(int) * off.size()) && rd(fp, nb.data(), sizeof(int) * nb.size());
    if (ok && H.verlet_ok) {
      ok = off[0] == 0 && off[n] == H.nnbr;
      for (int i = 0; i < n && ok; ++i) ok = off[i] <= off[i + 1];
      for (long k = 0; k < H.nnbr && ok; ++k) ok = nb[k] >= 0 && nb[k] < n;
      if (!ok) return fail(LBMDEM_EINVAL, "checkpoint '%s': the pair list is inconsistent", path);
    }
    if (ok) ok = hipMemcpy(h->V.offsets, off.data(), sizeof(int) * off.size(), hipMemcpyHostToDevice) == hipSuccess &&
                 (H.nnbr == 0 || hipMemcpy(h->V.nbr, nb.data(), sizeof(int) * nb.size(), hipMemcpyHostToDevice) == hipSuccess);
  }
  fill(h->V.wallflags, n);
  fill(h->obst[0], sizeof(int) * (size_t)h->L.plane);
  h->ocur = 0; h->obst_pending = false;
  h->chg_state[0] = h->chg_state[1] = 0;
  h->snap_ok[0] = h->snap_ok[1] = false;   // (the loaded map is not the picture lbmdem_create has just painted)
  for (int q = 0; q < 9 && ok; ++q) fill(h->f[0] + (size_t)q * h->L.plane, sizeof(double) * (size_t)h->L.plane);
  h->fcur = 0;
  if (ok && H.has_dist) {   // a strip with distributed grains: masks and message capacities as the writer had them. A
    CkptDist D;             // missing or short section fails the load HERE, not later inside a collective on one rank
    ok = rd(fp, &D, sizeof D) && memcmp(D.magic, "LBMDIST1", 8) == 0 && D.margin > 0 && D.cap_g > 0 && D.cap_t > 0 &&
         D.cap_l > 0 && D.cap_g <= n && D.cap_t <= n && D.cap_l <= n &&
         lbmdem_dist_enable_caps(h, D.margin, D.cap_g, D.cap_t, D.cap_l) == LBMDEM_OK;
    if (ok) { fill(h->dd.active, n); fill(h->dd.fluidmask, n); fill(h->owner, n); h->dist_poison = D.poison != 0; }
  }
  if (!ok) return fail(LBMDEM_EINVAL, "checkpoint '%s' is truncated or from a different decomposition", path);
  h->cfg = cfg;  // wall positions as saved
  h->force_mode = H.force_mode;
  h->L.lid6 = H.lid6;
  h->diag_always = H.diag_always != 0;
  if (H.has_carry) {  // the "previous contact" carries continue across the restart (no records yet: ct.carry stands)
    if (hipMemcpy(h->ct.carry, H.carry, sizeof H.carry, hipMemcpyHostToDevice) != hipSuccess)
      return fail(LBMDEM_EHIP, "checkpoint: carries not restored");
  }
  h->nbsteps = H.nbsteps;
  h->verlet_ok = H.verlet_ok != 0;
  h->verlet_tracks_positions = h->verlet_ok;
  // (the positions the list was built from, V.xreb / V.yreb, are not in the file: until the next rebuild the rasterisers
  // treat the list as outrun -- clear + repaint with atomics, never the list-based plain stores or the update in place)
  *h->moved_host = h->list_generation;
  if (h->verlet_ok) {  // the entry -> grain map is derived from the offsets
    launch_fill_own(h->V, n, h->stream);
    launch_tile_halo(h->V, n, h->stream);
    if (hipStreamSynchronize(h->stream) != hipSuccess) return fail(LBMDEM_EHIP, "k_fill_own failed");
  }
  guard.h = nullptr;
  *out = h;
  return LBMDEM_OK;
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

}  // extern "C"
#pragma GCC visibility pop
