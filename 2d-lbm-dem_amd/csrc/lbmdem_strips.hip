// lbmdem_strips.hip -- x-strip decomposition over the C ABI (one process per GPU): halo rows, the distributed-grain
// protocol (ownership, margin, the three neighbour messages), and the pieces of the drop-in outputs that need every
// rank's grains (the table sub-step on a full replica, the carries agreed over the ranks).

#include "lbmdem_handle.h"

#pragma GCC visibility push(default)
extern "C" {
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
int lbmdem_dist_enable_caps(lbmdem_handle* h, int M, long cap_g, long cap_t, long cap_l) {
  if (dist_alloc(h->dd, h->n, (int)cap_g, (int)cap_t, (int)cap_l) != 0) { dist_free(h->dd); return fail(LBMDEM_ENOMEM, "dist_alloc failed"); }
  RC_TRY(lbmdem_dem_tiles_by_index(h));   // (see there)
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
  return lbmdem_dist_enable_caps(h, M, cap_g, cap_t, cap_l);
} catch (...) {
  return fail(LBMDEM_EINVAL, "unexpected C++ exception");
}

// The "previous contact" carries as agreed over all ranks (strip decomposition: before a checkpoint, see
// lbmdem_comm_sync_carries): they stand until a younger contact is recorded.
int lbmdem_dist_set_carries(lbmdem_handle* h, const double* carry3) {
  SP_UNAVAILABLE("the strip decomposition with distributed grains");   // (ct.carry holds `real`s: 12 bytes in the float build)
  CHECK_H(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (!carry3) return fail(LBMDEM_EINVAL, "null buffer");
  HIP_TRY(hipStreamSynchronize(h->stream));
  HIP_TRY(hipMemcpy(h->ct.carry, carry3, sizeof(double) * 3, hipMemcpyHostToDevice));
  h->carry_from = h->substep_seq;
  return LBMDEM_OK;
}

// this rank's youngest record per carry (keys {0,0} = none) and its carry[] as it stands (only meaningful on the rank
// that ran the last table sub-step)
int lbmdem_dist_export_carries(lbmdem_handle* h, long long* carry_keys, double* carry_vals, double* carry_standing) {
  SP_UNAVAILABLE("the strip decomposition with distributed grains");
  CHECK_H(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
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

int lbmdem_dist_begin_period(lbmdem_handle* h) { return lbmdem_dist_begin_period_packed(h, nullptr, nullptr); }

// kin_lo / kin_hi (C transport): the two kinematics messages are packed by the same launch (null: not)
int lbmdem_dist_begin_period_packed(lbmdem_handle* h, void* kin_lo, void* kin_hi) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  // errors of earlier periods (a truncated message list, overlapping discs across a cut, a clash while merging tables)
  // are flagged on the device; the flag follows every period to pinned host memory and stops the run HERE, at the next
  // period, instead of letting it continue on truncated messages until somebody calls lbmdem_sync
#ifdef LBMDEM_AB   /* experiment build: a rank stepped alone on stale neighbour messages (scripts/strip_proxy*.py) raises the flag by design */
  static const bool ignore_flag = getenv("LBMDEM_IGNORE_DIST_ERRORS") != nullptr;
  if (ignore_flag) *h->ferr_host = 0;
#endif
  if (*h->ferr_host)
    return fail(LBMDEM_EINVAL, "strip decomposition: device error flag %d in an earlier fluid step (4: more grains near a cut "
                               "than the message capacity, 8: two ranks produced the same link sum, others: the force of a "
                               "grain on a cut could not be formed)", (int)*h->ferr_host);
  const lbmdem_config& c = h->cfg;
  DistGeom Gm;
  Gm.lo = (double)c.x_begin; Gm.hi = (double)c.x_end; Gm.margin = (double)h->dist_margin; Gm.dx = c.dx; Gm.Mgx = c.Mgx;
  Gm.has_lo = c.x_begin > 0; Gm.has_hi = c.x_end < c.lx; Gm.first = c.x_begin == 0; Gm.last = c.x_end == c.lx;
  Gm.gx0 = h->L.gx0; Gm.nxl = h->L.nxl;
  { int* t = h->dd.counters; h->dd.counters = h->dd.counters_alt; h->dd.counters_alt = t; }   // the set cleared last period
  h->fs.local_count = h->dd.counters + 6;
  if (kin_lo || kin_hi)
    launch_dist_classify_pack_kin(h->dd, Gm, h->n, h->kin[h->kcur].x1, h->r, h->rLB, h->owner, h->fs.error, h->ferr_mirror,
                                  h->kin[h->kcur], (real*)kin_lo, (real*)kin_hi, h->stream);
  else
    launch_dist_classify(h->dd, Gm, h->n, h->kin[h->kcur].x1, h->r, h->rLB, h->owner, h->fs.error, h->ferr_mirror, h->stream);
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

// C transport: the neighbours' TABLES messages merged, their KIN messages unpacked, and the obstacle map the fused
// kernel has finished with (obst[1 - ocur] after collide_stream) reset for the next rasterisation -- one launch
int lbmdem_dist_unpack_tables_kin_fill(lbmdem_handle* h, const void* tab_lo, const void* tab_hi, const void* kin_lo,
                                       const void* kin_hi) {
  CHECK_H(h);
  CHECK_NOT_SPLIT(h);
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  if (h->obst_pending) return fail(LBMDEM_EINVAL, "between obst_construction and collide_stream");
  launch_dist_unpack_tables_kin_fill(h->fs, (const real*)tab_lo, (const real*)tab_hi, h->dd.cap_t, h->dd, (const real*)kin_lo,
                                     (const real*)kin_hi, h->kin[h->kcur], h->n, h->obst[1 - h->ocur], h->L, h->stream);
  HIP_TRY(hipGetLastError());
  h->obst_reset_rows = h->L.nxl;
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
  // (only the LISTS: VerletWall's move of the right/top walls happens every updateVerlet steps in the reference,
  // main.c:1555-1561,1721 -- this extra rebuild must not move them early when dtt > 0)
  int rc = lbmdem_verlet_build_lists(h);
  if (rc != LBMDEM_OK) return rc;
  if (!h->dx_ready) {
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (diag_extra_alloc(h->dx, h->n, h->V.cap, h->ct.carry) != 0) return fail(LBMDEM_ENOMEM, "diagnostic buffers: hipMalloc failed");
    h->dx_ready = true;
  }
  const int film = (h->nbsteps % h->cfg.phys.stepFilm == 0) ? 1 : 0;
  const DemParams P = dem_params(h);
  launch_dem_substep(h->kin[h->kcur], h->kin[1 - h->kcur], h->r, h->m, h->It, h->fhf, h->V, h->gp, P, film, h->diag,
                     &h->dx, nullptr, nullptr, h->substep_seq, nullptr, ObstFillJob{nullptr, h->L, 0, 0}, h->stream);
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

}  // extern "C"
#pragma GCC visibility pop
