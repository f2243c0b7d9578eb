// lbmdem_comm.hip -- the library's own RCCL transport for the distributed-grain strips (the C host driver's
// `--gpus N`; strips.py does the same over torch.distributed).

#include "lbmdem_handle.h"

// ---- RCCL transport for the distributed-grain strips (the C host driver; strips.py does the same over
// torch.distributed) ---------------------------------------------------------------------------------------------
// RCCL is loaded with dlopen when the first communicator is made: processes that never call lbmdem_comm_* (the
// single-GPU driver, Python with torch's own RCCL) do not load a second copy of the library.


#include <rccl/rccl.h>

namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.lib) return LBMDEM_OK;
  // LBMDEM_RCCL_LIBRARY names the RCCL build to use (a non-standard install; the tests' several-ranks-on-one-GPU stand-in,
  // tests/rccl_shim). Otherwise a copy that is already in the process (PyTorch-ROCm ships its own as "librccl.so") is
  // reused: one RCCL per process
  void* lib = nullptr;
  const char* named = getenv("LBMDEM_RCCL_LIBRARY");
  if (named && *named) {
    lib = dlopen(named, RTLD_NOW | RTLD_LOCAL);
    if (!lib) return fail(LBMDEM_EHIP, "cannot load LBMDEM_RCCL_LIBRARY=%s: %s", named, dlerror());
    // said once, loudly: a production run must not use a stand-in for RCCL without anybody noticing
    fprintf(stderr, "liblbmdem_hip: the communication library is LBMDEM_RCCL_LIBRARY=%s, not the system's RCCL\n", named);
  }
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
  if (!lib) lib = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
  if (!lib) return fail(LBMDEM_EHIP, "cannot load RCCL: %s", dlerror());
#define RCCL_SYM(field, name)                                                                  \
  g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(lib, name));                  \
  if (!g_rccl.field) { dlclose(lib); return fail(LBMDEM_EHIP, "RCCL lacks %s", name); }
  RCCL_SYM(GetUniqueId, "ncclGetUniqueId") RCCL_SYM(CommInitRank, "ncclCommInitRank") RCCL_SYM(CommDestroy, "ncclCommDestroy")
  RCCL_SYM(Send, "ncclSend") RCCL_SYM(Recv, "ncclRecv") RCCL_SYM(GroupStart, "ncclGroupStart") RCCL_SYM(GroupEnd, "ncclGroupEnd")
  RCCL_SYM(AllReduce, "ncclAllReduce") RCCL_SYM(GetErrorString, "ncclGetErrorString")
#undef RCCL_SYM
  g_rccl.lib = lib;
  return LBMDEM_OK;
}
}  // namespace

#define NCCL_TRY(expr)                                                                                      \
  do {                                                                                                      \
    ncclResult_t r_ = (expr);                                                                               \
    if (r_ != ncclSuccess) return fail(LBMDEM_EHIP, "%s failed: %s", #expr, g_rccl.GetErrorString(r_));    \
  } while (0)

// message classes that can be in flight at the same time each have their own side stream
enum { LANE_KIN = 0, LANE_HALO, LANE_TAB, LANE_FHF, LANE_COUNT };

struct lbmdem_comm {
  // one communicator per lane: messages of different lanes are in flight at the same time, and RCCL orders the
  // operations of ONE communicator
  ncclComm_t nccl[LANE_COUNT] = {};
  int rank = 0, world = 1, device = 0;
  hipStream_t side[LANE_COUNT] = {};
  hipEvent_t ready[LANE_COUNT] = {}, done[LANE_COUNT] = {};
  // device buffers for one handle: [kind or halo][side][send/recv]
  lbmdem_handle* bound = nullptr;
  double* buf[3][2][2] = {};
  size_t count[3] = {};   // doubles per message: KIN, FHF, TABLES
  double* scratch = nullptr;
  float* vtk_dev = nullptr;   // staging of one strip's VTK fields (lbmdem_comm_write_vtk), kept between frames
  size_t vtk_floats = 0;
  bool stale = false;         // experiment build only (lbmdem_comm_debug_stale): the step runs without its transfers
};
#ifdef LBMDEM_AB
#define COMM_STALE(c) ((c)->stale)
#else
#define COMM_STALE(c) false
#endif

#pragma GCC visibility push(default)
extern "C" {

int lbmdem_comm_unique_id(void* id128) {
  if (!id128) return fail(LBMDEM_EINVAL, "null buffer");
  int rc = rccl_load();
  if (rc != LBMDEM_OK) return rc;
  static_assert(sizeof(ncclUniqueId) * LANE_COUNT == LBMDEM_COMM_ID_BYTES, "one ncclUniqueId per lane");
  for (int l = 0; l < LANE_COUNT; ++l) NCCL_TRY(g_rccl.GetUniqueId(reinterpret_cast<ncclUniqueId*>(id128) + l));
  return LBMDEM_OK;
}

int lbmdem_comm_destroy(lbmdem_comm* c) {
  if (!c) return LBMDEM_OK;
  (void)hipSetDevice(c->device);
  for (int l = 0; l < LANE_COUNT; ++l) {
    if (c->side[l]) { (void)hipStreamSynchronize(c->side[l]); (void)hipStreamDestroy(c->side[l]); }
    if (c->ready[l]) (void)hipEventDestroy(c->ready[l]);
    if (c->done[l]) (void)hipEventDestroy(c->done[l]);
  }
  for (auto& k : c->buf) for (auto& s : k) for (double*& p : s) if (p) (void)hipFree(p);
  if (c->scratch) (void)hipFree(c->scratch);
  if (c->vtk_dev) (void)hipFree(c->vtk_dev);
  for (int l = 0; l < LANE_COUNT; ++l) if (c->nccl[l]) (void)g_rccl.CommDestroy(c->nccl[l]);
  delete c;
  return LBMDEM_OK;
}

int lbmdem_comm_create(const void* id128, int rank, int world, int device, lbmdem_comm** out) try {
  SP_UNAVAILABLE("the RCCL transport");
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_create arguments");
  *out = nullptr;
  int rc = rccl_load();
  if (rc != LBMDEM_OK) return rc;
  HIP_TRY(hipSetDevice(device));
  lbmdem_comm* c = new lbmdem_comm();
  c->rank = rank; c->world = world; c->device = device;
  for (int l = 0; l < LANE_COUNT; ++l) {   // every rank creates them in the same order
    ncclUniqueId id;
    memcpy(&id, static_cast<const char*>(id128) + l * sizeof id, sizeof id);
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl[l], world, id, rank);
    if (r != ncclSuccess) { lbmdem_comm_destroy(c); return fail(LBMDEM_EHIP, "ncclCommInitRank failed: %s", g_rccl.GetErrorString(r)); }
  }
  for (int l = 0; l < LANE_COUNT; ++l) {
    if (hipStreamCreateWithFlags(&c->side[l], hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&c->ready[l], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->done[l], hipEventDisableTiming) != hipSuccess) {
      lbmdem_comm_destroy(c);
      return fail(LBMDEM_EHIP, "stream / event creation failed");
    }
  }
  if (hipMalloc((void**)&c->scratch, sizeof(double) * 1024) != hipSuccess) { lbmdem_comm_destroy(c); return fail(LBMDEM_ENOMEM, "hipMalloc"); }
  *out = c;
  return LBMDEM_OK;
} catch (...) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

// transfers of one lane with both neighbours: they depend on what the main stream has enqueued so far, not on
// what it enqueues next; `done` is what the main stream waits for later
static int comm_begin(lbmdem_comm* c, hipStream_t main, int lane, int kind, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  HIP_TRY(hipEventRecord(c->ready[lane], main));
  HIP_TRY(hipStreamWaitEvent(c->side[lane], c->ready[lane], 0));
  if (!COMM_STALE(c)) {
    NCCL_TRY(g_rccl.GroupStart());
    for (int s = 0; s < 2; ++s) {
      if (!has[s]) continue;
      const int peer = s == 0 ? c->rank - 1 : c->rank + 1;
      NCCL_TRY(g_rccl.Send(c->buf[kind][s][0], c->count[kind], ncclDouble, peer, c->nccl[lane], c->side[lane]));
      NCCL_TRY(g_rccl.Recv(c->buf[kind][s][1], c->count[kind], ncclDouble, peer, c->nccl[lane], c->side[lane]));
    }
    NCCL_TRY(g_rccl.GroupEnd());
  }
  HIP_TRY(hipEventRecord(c->done[lane], c->side[lane]));
  return LBMDEM_OK;
}
// A transfer nothing can overlap with (link-sum tables, forces: the next kernel needs them) simply takes its place in the
// main stream: measured with lbmdem_comm_exchange_probe, the two event hand-overs of the side-stream form cost ~25 us
// more than the transfer itself (~10 us).
static int comm_inline(lbmdem_comm* c, hipStream_t main, int lane, int kind, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  if (COMM_STALE(c)) return LBMDEM_OK;
  NCCL_TRY(g_rccl.GroupStart());
  for (int s = 0; s < 2; ++s) {
    if (!has[s]) continue;
    const int peer = s == 0 ? c->rank - 1 : c->rank + 1;
    NCCL_TRY(g_rccl.Send(c->buf[kind][s][0], c->count[kind], ncclDouble, peer, c->nccl[lane], main));
    NCCL_TRY(g_rccl.Recv(c->buf[kind][s][1], c->count[kind], ncclDouble, peer, c->nccl[lane], main));
  }
  NCCL_TRY(g_rccl.GroupEnd());
  return LBMDEM_OK;
}
static int comm_end(lbmdem_comm* c, hipStream_t main, int lane, const bool has[2]) {
  if (!has[0] && !has[1]) return LBMDEM_OK;
  HIP_TRY(hipStreamWaitEvent(main, c->done[lane], 0));
  return LBMDEM_OK;
}

static int comm_bind(lbmdem_comm* c, lbmdem_handle* h) {
  if (c->bound == h) return LBMDEM_OK;
  if (c->bound) return fail(LBMDEM_EINVAL, "a communicator serves one handle");
  if (!h->dist) return fail(LBMDEM_EINVAL, "lbmdem_dist_enable first");
  c->count[LBMDEM_MSG_KIN] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_KIN);
  c->count[LBMDEM_MSG_FHF] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_FHF);
  c->count[LBMDEM_MSG_TABLES] = (size_t)lbmdem_dist_message_doubles(h, LBMDEM_MSG_TABLES);
  for (int k = 0; k < 3; ++k)   // (the f halo rows travel straight between the lattices: halo_exchange)
    for (int s = 0; s < 2; ++s)
      for (int d = 0; d < 2; ++d) {
        HIP_TRY(hipMalloc((void**)&c->buf[k][s][d], sizeof(double) * (c->count[k] ? c->count[k] : 1)));
        HIP_TRY(hipMemset(c->buf[k][s][d], 0, sizeof(double) * (c->count[k] ? c->count[k] : 1)));
      }
  HIP_TRY(hipDeviceSynchronize());
  c->bound = h;
  return LBMDEM_OK;
}


// The f halo travels straight out of and into the lattice: in the device layout f[xl][y / 16][q][y % 16] a lattice row is
// one contiguous slab of 9 * sy reals, so the `halo` owned rows next to a cut ARE the message and the neighbour's halo rows
// ARE the receive buffer -- no pack / unpack kernels, no staging copies.
static int halo_exchange(lbmdem_handle* h, lbmdem_comm* c, hipStream_t st, const bool has[2]) {
  const LatticeView& L = h->L;
  const int H = h->cfg.halo;
  if (H < 1 || L.xo1 - L.xo0 < H) return fail(LBMDEM_EINVAL, "strip narrower than its halo");
  if (COMM_STALE(c)) return LBMDEM_OK;
  real* f = h->f[h->fcur];                     // the lattice the edge rows have just been written to
  const size_t row = (size_t)L.sy * 9, cnt = row * H;
  NCCL_TRY(g_rccl.GroupStart());
  if (has[0]) {
    NCCL_TRY(g_rccl.Send(f + row * L.xo0, cnt, ncclDouble, c->rank - 1, c->nccl[LANE_HALO], st));
    NCCL_TRY(g_rccl.Recv(f + row * (L.xo0 - H), cnt, ncclDouble, c->rank - 1, c->nccl[LANE_HALO], st));
  }
  if (has[1]) {
    NCCL_TRY(g_rccl.Send(f + row * (L.xo1 - H), cnt, ncclDouble, c->rank + 1, c->nccl[LANE_HALO], st));
    NCCL_TRY(g_rccl.Recv(f + row * L.xo1, cnt, ncclDouble, c->rank + 1, c->nccl[LANE_HALO], st));
  }
  NCCL_TRY(g_rccl.GroupEnd());
  return LBMDEM_OK;
}

// One fluid step of a strip with its neighbours. Launches on the critical path (main stream), per period:
//   classification + both kinematics messages (1) . rasteriser (1; the map was reset by the unpack launch of the period
//   before) . interior rows of the fused kernel (1) . this rank's part of the neighbours' link-sum tables (1) . [TABLES
//   exchange] . unpack: tables merged + kinematics unpacked + dead obstacle map reset (1) . force table + gather queue (2)
//   . forces packed (1) . [FHF exchange] . forces unpacked (1) . npDEM sub-steps.
// Off the critical path: the edge rows of the fused kernel run on the halo lane's stream NEXT TO the interior rows (both
// read the old lattice and write disjoint rows of the new one), followed there by the halo exchange straight between the
// lattices; the kinematics travel on their own lane under the whole fluid step.
int lbmdem_comm_lbm_step(lbmdem_handle* h, lbmdem_comm* c) {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  RC_TRY(comm_bind(c, h));
  const bool has[2] = {h->cfg.x_begin > 0, h->cfg.x_end < h->cfg.lx};
  const bool cut = has[0] || has[1];
  hipStream_t main = h->stream;
  double* (*B)[2][2] = c->buf;
  // ownership + message lists from the current positions, and the two kinematics messages, in one launch
  RC_TRY(lbmdem_dist_begin_period_packed(h, has[0] ? B[LBMDEM_MSG_KIN][0][0] : nullptr, has[1] ? B[LBMDEM_MSG_KIN][1][0] : nullptr));
  RC_TRY(comm_begin(c, main, LANE_KIN, LBMDEM_MSG_KIN, has));      // margin refresh / migration, under the fluid step
  RC_TRY(lbmdem_obst_construction(h));
  if (cut) {
    RC_TRY(lbmdem_collide_stream_prepare(h));
    hipStream_t side = c->side[LANE_HALO];
#ifdef LBMDEM_AB   /* A/B: the edge rows through the main stream, only the exchange on the lane's stream */
    static const bool edges_main = getenv("LBMDEM_COMM_EDGES_MAIN") != nullptr;
    if (edges_main) RC_TRY(lbmdem_collide_stream_part_on(h, LBMDEM_CS_EDGES, main));
#else
    const bool edges_main = false;
#endif
    HIP_TRY(hipEventRecord(c->ready[LANE_HALO], main));
    HIP_TRY(hipStreamWaitEvent(side, c->ready[LANE_HALO], 0));
    if (!edges_main) RC_TRY(lbmdem_collide_stream_part_on(h, LBMDEM_CS_EDGES, side));      // the rows the neighbours wait for ...
    RC_TRY(halo_exchange(h, c, side, has));
    HIP_TRY(hipEventRecord(c->done[LANE_HALO], side));
    RC_TRY(lbmdem_collide_stream_part_on(h, LBMDEM_CS_INTERIOR, main));   // ... next to the bulk of the rows
    HIP_TRY(hipStreamWaitEvent(main, c->done[LANE_HALO], 0));
  } else {
    RC_TRY(lbmdem_collide_stream(h));
  }
  RC_TRY(lbmdem_dist_pack2(h, LBMDEM_MSG_TABLES, has[0] ? B[LBMDEM_MSG_TABLES][0][0] : nullptr, has[1] ? B[LBMDEM_MSG_TABLES][1][0] : nullptr));
  RC_TRY(comm_inline(c, main, LANE_TAB, LBMDEM_MSG_TABLES, has));  // link sums of the grains the neighbours own
  RC_TRY(comm_end(c, main, LANE_KIN, has));                        // (arrived long ago)
  if (cut)
    RC_TRY(lbmdem_dist_unpack_tables_kin_fill(h, has[0] ? B[LBMDEM_MSG_TABLES][0][1] : nullptr, has[1] ? B[LBMDEM_MSG_TABLES][1][1] : nullptr,
                                              has[0] ? B[LBMDEM_MSG_KIN][0][1] : nullptr, has[1] ? B[LBMDEM_MSG_KIN][1][1] : nullptr));
  RC_TRY(lbmdem_forces_fluid(h));
  RC_TRY(lbmdem_dist_pack2(h, LBMDEM_MSG_FHF, has[0] ? B[LBMDEM_MSG_FHF][0][0] : nullptr, has[1] ? B[LBMDEM_MSG_FHF][1][0] : nullptr));
  RC_TRY(comm_inline(c, main, LANE_FHF, LBMDEM_MSG_FHF, has));    // forces of the margin grains, from their owners
  RC_TRY(lbmdem_dist_unpack2(h, LBMDEM_MSG_FHF, has[0] ? B[LBMDEM_MSG_FHF][0][1] : nullptr, has[1] ? B[LBMDEM_MSG_FHF][1][1] : nullptr));
  return LBMDEM_OK;
}

// Bitwise merge of host buffers whose non-zero bits are DISJOINT across the ranks (every grain has one owner, every
// lattice column one rank): an integer SUM all-reduce then is a bitwise OR (no bit position receives two ones, so no
// carries). Not on the step path (output cadence). In place; every rank gets the merged buffer.
int lbmdem_comm_allreduce_bits(lbmdem_comm* c, void* host_buf, size_t nbytes) {
  if (!c || !host_buf || nbytes == 0) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_allreduce_bits arguments");
  HIP_TRY(hipSetDevice(c->device));
  const size_t words = (nbytes + 7) / 8;
  unsigned long long* d = nullptr;
  HIP_TRY(hipMalloc((void**)&d, words * 8));
  hipError_t e = hipMemset(d, 0, words * 8);
  if (e == hipSuccess) e = hipMemcpy(d, host_buf, nbytes, hipMemcpyHostToDevice);
  if (e != hipSuccess) { (void)hipFree(d); HIP_TRY(e); }
  const ncclResult_t r = g_rccl.AllReduce(d, d, words, ncclUint64, ncclSum, c->nccl[0], c->side[0]);
  if (r != ncclSuccess) { (void)hipFree(d); return fail(LBMDEM_EHIP, "ncclAllReduce failed: %s", g_rccl.GetErrorString(r)); }
  e = hipStreamSynchronize(c->side[0]);
  if (e == hipSuccess) e = hipMemcpy(host_buf, d, nbytes, hipMemcpyDeviceToHost);
  (void)hipFree(d);
  HIP_TRY(e);
  return LBMDEM_OK;
}

// The sub-step that feeds write_DEM (the one that brings the step counter to a multiple of 4000, main.c:1773) over
// the ranks: exports merged with lbmdem_comm_allreduce_bits, the youngest carry record picked over all ranks, rank 0
// runs lbmdem_dist_table_substep on the full replica, the others their ordinary sub-step.
static int comm_table_substep(lbmdem_handle* h, lbmdem_comm* c) try {
  const int n = h->n, W = c->world;
  std::vector<double> st(12 * (size_t)n), vals(3 * (size_t)W, 0.0);
  std::vector<unsigned char> owned(n);
  std::vector<long long> keys(6 * (size_t)W, 0);
  RC_TRY(lbmdem_dist_export_owned(h, st.data(), owned.data(), keys.data() + 6 * (size_t)c->rank, vals.data() + 3 * (size_t)c->rank));
  RC_TRY(lbmdem_comm_allreduce_bits(c, st.data(), sizeof(double) * st.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, owned.data(), owned.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, keys.data(), sizeof(long long) * keys.size()));
  RC_TRY(lbmdem_comm_allreduce_bits(c, vals.data(), sizeof(double) * vals.size()));
  for (int i = 0; i < n; ++i)
    if (owned[i] != 1) return fail(LBMDEM_EINVAL, "grain %d has %d owners at sub-step %ld", i, (int)owned[i], h->nbsteps);
  if (c->rank != 0) return lbmdem_dem_substep(h);
  double best_val[3] = {0, 0, 0};
  int has[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) {
    long long b0 = 0, b1 = 0;
    for (int r = 0; r < W; ++r) {
      const long long k0 = keys[6 * (size_t)r + 2 * k], k1 = keys[6 * (size_t)r + 2 * k + 1];
      if (k0 > b0 || (k0 == b0 && k0 != 0 && k1 > b1)) { b0 = k0; b1 = k1; best_val[k] = vals[3 * (size_t)r + k]; has[k] = 1; }
    }
  }
  return lbmdem_dist_table_substep(h, st.data(), best_val, has);
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

// Before a checkpoint: every rank learns the carries as the reference holds them now -- per carry the youngest record
// over all ranks, else what rank 0 has kept since the last table sub-step -- and they stand from here on.
int lbmdem_comm_sync_carries(lbmdem_handle* h, lbmdem_comm* c) try {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  const int W = c->world;
  std::vector<long long> keys(6 * (size_t)W, 0);
  std::vector<double> vals(3 * (size_t)W, 0.0), standing(3 * (size_t)W, 0.0);
  RC_TRY(lbmdem_dist_export_carries(h, keys.data() + 6 * (size_t)c->rank, vals.data() + 3 * (size_t)c->rank,
                                    standing.data() + 3 * (size_t)c->rank));
  if (W > 1) {
    RC_TRY(lbmdem_comm_allreduce_bits(c, keys.data(), sizeof(long long) * keys.size()));
    RC_TRY(lbmdem_comm_allreduce_bits(c, vals.data(), sizeof(double) * vals.size()));
    RC_TRY(lbmdem_comm_allreduce_bits(c, standing.data(), sizeof(double) * standing.size()));
  }
  double out[3];
  for (int k = 0; k < 3; ++k) {
    out[k] = standing[k];   // rank 0's
    long long b0 = 0, b1 = 0;
    for (int r = 0; r < W; ++r) {
      const long long k0 = keys[6 * (size_t)r + 2 * k], k1 = keys[6 * (size_t)r + 2 * k + 1];
      if (k0 > b0 || (k0 == b0 && k0 != 0 && k1 > b1)) { b0 = k0; b1 = k1; out[k] = vals[3 * (size_t)r + k]; }
    }
  }
  return lbmdem_dist_set_carries(h, out);
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

int lbmdem_comm_run(lbmdem_handle* h, lbmdem_comm* c, long n_dem_steps) {
  CHECK_H(h);
  for (long k = 0; k < n_dem_steps; ++k) {
    if (h->nbsteps % h->cfg.npDEM == 0) RC_TRY(lbmdem_comm_lbm_step(h, c));                     // main.c:1710-1718
    if (h->nbsteps % h->cfg.phys.updateVerlet == 0) RC_TRY(lbmdem_verlet_rebuild(h));            // main.c:1721-1724
    if ((h->nbsteps + 1) % 4000 == 0) { RC_TRY(comm_table_substep(h, c)); continue; }            // feeds write_DEM, main.c:1773
    const long run = lbmdem_dem_chain_length(h, n_dem_steps - k, 1);
    if (run) { RC_TRY(lbmdem_dem_chain(h, run, 1)); k += run - 1; }
    else RC_TRY(lbmdem_dem_substep(h));                                                          // main.c:1733-1764
  }
  return LBMDEM_OK;
}

// write_vtk (main.c:237-338) of the whole lattice: every rank renders the five fields of its own columns and SENDS them
// to rank 0 (point to point, one reused device staging buffer of the widest strip), which places them and writes the
// five files. Only rank 0 holds lattice-sized host arrays (as the single-GPU writer does).
int lbmdem_comm_write_vtk(lbmdem_handle* h, lbmdem_comm* c, const char* dir, int nfile) try {
  CHECK_H(h);
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  const int lx = h->cfg.lx, ly = h->cfg.ly, W = c->world;
  const LatticeView& L = h->L;
  const int nx = L.xo1 - L.xo0, x0 = L.gx0 + L.xo0;
  const size_t part = (size_t)nx * ly;
  // Every local step that can fail comes BEFORE a vote over all ranks: a rank that returned early would leave rank 0
  // blocked in its receive for ever (no time-out in RCCL).
  std::vector<float> loc;
  int rc_local = LBMDEM_OK;
  try { loc.resize(11 * part); } catch (const std::bad_alloc&) { rc_local = fail(LBMDEM_ENOMEM, "host memory allocation failed"); }
  if (rc_local == LBMDEM_OK)
    rc_local = lbmdem_download_vtk_fields(h, loc.data(), loc.data() + part, loc.data() + 4 * part, loc.data() + 7 * part, loc.data() + 8 * part);
  // who owns which columns; last entry: ranks that failed so far
  if (2 * W + 1 > 1024) return fail(LBMDEM_EINVAL, "too many ranks");
  std::vector<double> strips(2 * (size_t)W + 1, 0.0);
  strips[2 * (size_t)c->rank] = x0; strips[2 * (size_t)c->rank + 1] = nx;
  strips[2 * (size_t)W] = rc_local == LBMDEM_OK ? 0.0 : 1.0;
  if (W > 1) RC_TRY(lbmdem_comm_allreduce_sum(c, strips.data(), 2 * W + 1));
  if (strips[2 * (size_t)W] != 0.0)
    return rc_local != LBMDEM_OK ? rc_local : fail(LBMDEM_EHIP, "write_vtk: %d rank(s) could not render their columns", (int)strips[2 * (size_t)W]);
  size_t widest = 0, covered = 0;
  for (int r = 0; r < W; ++r) { const size_t w = (size_t)strips[2 * (size_t)r + 1]; if (w > widest) widest = w; covered += w; }
  if (covered != (size_t)lx) return fail(LBMDEM_EINVAL, "the ranks' strips cover %zu of %d columns", covered, lx);   // (the same on every rank)
  if (W > 1) {
    hipError_t e = hipSuccess;
    if (c->vtk_floats < 11 * widest * ly) {
      if (c->vtk_dev) { (void)hipFree(c->vtk_dev); c->vtk_dev = nullptr; c->vtk_floats = 0; }
      e = hipMalloc((void**)&c->vtk_dev, sizeof(float) * 11 * widest * ly);
      if (e == hipSuccess) c->vtk_floats = 11 * widest * ly;
    }
    if (e == hipSuccess && c->rank != 0) e = hipMemcpy(c->vtk_dev, loc.data(), sizeof(float) * loc.size(), hipMemcpyHostToDevice);
    double bad = e == hipSuccess ? 0.0 : 1.0;
    RC_TRY(lbmdem_comm_allreduce_sum(c, &bad, 1));   // second vote: the staging buffers stand on every rank
    if (bad != 0.0) return fail(LBMDEM_EHIP, "write_vtk: staging buffer on %d rank(s) failed%s%s", (int)bad, e == hipSuccess ? "" : ": ", e == hipSuccess ? "" : hipGetErrorString(e));
  }
  if (c->rank != 0) {
    NCCL_TRY(g_rccl.Send(c->vtk_dev, loc.size(), ncclFloat, 0, c->nccl[0], c->side[0]));
    HIP_TRY(hipStreamSynchronize(c->side[0]));
    return LBMDEM_OK;
  }
  std::vector<float> fields(11 * (size_t)lx * ly, 0.f);
  lbmdem_vtk_place_block(fields.data(), lx, ly, x0, nx, loc.data());
  for (int r = 1; r < W; ++r) {
    const int rx0 = (int)strips[2 * (size_t)r], rnx = (int)strips[2 * (size_t)r + 1];
    const size_t n = 11 * (size_t)rnx * ly;
    if (loc.size() < n) loc.resize(n);
    NCCL_TRY(g_rccl.Recv(c->vtk_dev, n, ncclFloat, r, c->nccl[0], c->side[0]));
    HIP_TRY(hipStreamSynchronize(c->side[0]));
    HIP_TRY(hipMemcpy(loc.data(), c->vtk_dev, sizeof(float) * n, hipMemcpyDeviceToHost));
    lbmdem_vtk_place_block(fields.data(), lx, ly, rx0, rnx, loc.data());
  }
  return lbmdem_write_vtk_fields(dir, nfile, lx, ly, fields.data());
} catch (const std::bad_alloc&) {
  return fail(LBMDEM_ENOMEM, "host memory allocation failed");
}

int lbmdem_comm_allreduce_sum(lbmdem_comm* c, double* values, int n) {
  if (!c || !values || n < 1 || n > 1024) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_allreduce_sum arguments");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(hipMemcpy(c->scratch, values, sizeof(double) * n, hipMemcpyHostToDevice));
  NCCL_TRY(g_rccl.AllReduce(c->scratch, c->scratch, (size_t)n, ncclDouble, ncclSum, c->nccl[0], c->side[0]));
  HIP_TRY(hipStreamSynchronize(c->side[0]));
  HIP_TRY(hipMemcpy(values, c->scratch, sizeof(double) * n, hipMemcpyDeviceToHost));
  return LBMDEM_OK;
}

/* A send to and a receive from THIS rank, grouped on a side stream while the caller's stream is busy: the
 * transport of lbmdem_comm_lbm_step exercised with a single rank. */
int lbmdem_comm_selftest(lbmdem_comm* c, int doubles) {
  if (!c || doubles < 1) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_selftest arguments");
  HIP_TRY(hipSetDevice(c->device));
  double *a = nullptr, *b = nullptr;
  HIP_TRY(hipMalloc((void**)&a, sizeof(double) * doubles));
  HIP_TRY(hipMalloc((void**)&b, sizeof(double) * doubles));
  std::vector<double> ha(doubles), hb(doubles, -1.0);
  for (int k = 0; k < doubles; ++k) ha[k] = 0.5 * k + 1.0;
  hipStream_t main = nullptr;
  hipError_t e = hipStreamCreateWithFlags(&main, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipMemcpyAsync(a, ha.data(), sizeof(double) * doubles, hipMemcpyHostToDevice, main);
  if (e == hipSuccess) e = hipEventRecord(c->ready[LANE_HALO], main);
  if (e == hipSuccess) e = hipStreamWaitEvent(c->side[LANE_HALO], c->ready[LANE_HALO], 0);
  ncclResult_t r = ncclSuccess;
  if (e == hipSuccess) {
    r = g_rccl.GroupStart();
    if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[LANE_HALO], c->side[LANE_HALO]);
    if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[LANE_HALO], c->side[LANE_HALO]);
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  if (e == hipSuccess && r == ncclSuccess) e = hipEventRecord(c->done[LANE_HALO], c->side[LANE_HALO]);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamWaitEvent(main, c->done[LANE_HALO], 0);
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpyAsync(hb.data(), b, sizeof(double) * doubles, hipMemcpyDeviceToHost, main);
  if (e == hipSuccess && r == ncclSuccess) e = hipStreamSynchronize(main);
  if (main) (void)hipStreamDestroy(main);
  (void)hipFree(a); (void)hipFree(b);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL self send/recv failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  for (int k = 0; k < doubles; ++k) if (hb[k] != ha[k]) return fail(LBMDEM_EHIP, "self send/recv returned wrong data at %d", k);
  if (c->world == 1) return LBMDEM_OK;
  // several ranks: the step's own pattern -- on every lane one grouped exchange with both neighbours, all lanes in
  // flight at once -- with a payload that names sender and lane
  const int left = c->rank - 1, right = c->rank + 1 < c->world ? c->rank + 1 : -1;
  double* d = nullptr;   // [lane][send L, send R, recv L, recv R][doubles]
  HIP_TRY(hipMalloc((void**)&d, sizeof(double) * doubles * 4 * LANE_COUNT));
  std::vector<double> host((size_t)doubles * 4 * LANE_COUNT, -1.0);
  auto value = [&](int rank, int lane, int to_right, int k) { return 1000.0 * rank + 100.0 * lane + 10.0 * to_right + 1e-3 * k; };
  for (int l = 0; l < LANE_COUNT; ++l)
    for (int sd = 0; sd < 2; ++sd)
      for (int k = 0; k < doubles; ++k) host[((size_t)l * 4 + sd) * doubles + k] = value(c->rank, l, sd, k);
  e = hipMemcpy(d, host.data(), sizeof(double) * host.size(), hipMemcpyHostToDevice);
  r = ncclSuccess;
  for (int l = 0; l < LANE_COUNT && e == hipSuccess && r == ncclSuccess; ++l) {
    double* base = d + (size_t)l * 4 * doubles;
    r = g_rccl.GroupStart();
    if (left >= 0 && r == ncclSuccess) r = g_rccl.Send(base, (size_t)doubles, ncclDouble, left, c->nccl[l], c->side[l]);
    if (left >= 0 && r == ncclSuccess) r = g_rccl.Recv(base + 2 * (size_t)doubles, (size_t)doubles, ncclDouble, left, c->nccl[l], c->side[l]);
    if (right >= 0 && r == ncclSuccess) r = g_rccl.Send(base + (size_t)doubles, (size_t)doubles, ncclDouble, right, c->nccl[l], c->side[l]);
    if (right >= 0 && r == ncclSuccess) r = g_rccl.Recv(base + 3 * (size_t)doubles, (size_t)doubles, ncclDouble, right, c->nccl[l], c->side[l]);
    ncclResult_t r2 = g_rccl.GroupEnd();
    if (r == ncclSuccess) r = r2;
  }
  for (int l = 0; l < LANE_COUNT; ++l) {
    const hipError_t e2 = hipStreamSynchronize(c->side[l]);
    if (e == hipSuccess) e = e2;
  }
  if (e == hipSuccess && r == ncclSuccess) e = hipMemcpy(host.data(), d, sizeof(double) * host.size(), hipMemcpyDeviceToHost);
  (void)hipFree(d);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL neighbour exchange failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  for (int l = 0; l < LANE_COUNT; ++l)
    for (int k = 0; k < doubles; ++k) {
      // the left neighbour's message "to the right" lands in recv L, the right neighbour's "to the left" in recv R
      if (left >= 0 && host[((size_t)l * 4 + 2) * doubles + k] != value(left, l, 1, k))
        return fail(LBMDEM_EHIP, "lane %d: wrong data from rank %d at %d", l, left, k);
      if (right >= 0 && host[((size_t)l * 4 + 3) * doubles + k] != value(right, l, 0, k))
        return fail(LBMDEM_EHIP, "lane %d: wrong data from rank %d at %d", l, right, k);
    }
  return LBMDEM_OK;
}

#ifdef LBMDEM_AB
/* Measurement helper, only in the experiment build (not declared in include/lbmdem_hip.h; scripts/strip_proxy_c.py): from
 * here on the fluid step of this rank runs WITHOUT its transfers -- the message buffers and halo rows keep what arrived
 * last (same sizes, same work) -- so that one rank of a decomposition can be stepped and timed alone on a one-GPU box. */
int lbmdem_comm_debug_stale(lbmdem_comm* c, int on) {
  if (!c) return fail(LBMDEM_EINVAL, "null communicator");
  c->stale = on != 0;
  return LBMDEM_OK;
}

/* Measurement helper, only in the experiment build (make AB=1 -> liblbmdem_hip_ab.so; bound by scripts/exchange_probe.py,
 * not declared in include/lbmdem_hip.h): what one exchange on the step's critical path costs on this stack. `iters` times
 * { small kernel on a main stream; ready event -> side stream; grouped send + receive of `doubles` values to this rank
 * itself; done event -> main stream; small kernel on the main stream }, timed with events on the main stream, and the
 * same loop without the exchange, and with the send + receive enqueued on the main stream itself (no events).
 * us[0] = mean with the exchange on the side stream, us[1] = without, us[2] = with it in line. */
int lbmdem_comm_exchange_probe(lbmdem_comm* c, int doubles, int iters, double* us) {
  if (!c || doubles < 1 || iters < 1 || !us) return fail(LBMDEM_EINVAL, "bad lbmdem_comm_exchange_probe arguments");
  HIP_TRY(hipSetDevice(c->device));
  double *a = nullptr, *b = nullptr;
  hipStream_t main = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc((void**)&a, sizeof(double) * doubles);
  if (e == hipSuccess) e = hipMalloc((void**)&b, sizeof(double) * doubles);
  if (e == hipSuccess) e = hipStreamCreateWithFlags(&main, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  ncclResult_t r = ncclSuccess;
  const int lane = LANE_TAB;
  for (int with = 2; with >= 0 && e == hipSuccess && r == ncclSuccess; --with) {
    for (int pass = 0; pass < 2 && e == hipSuccess && r == ncclSuccess; ++pass) {   // pass 0 warms up
      const int n = pass == 0 ? 10 : iters;
      if (pass == 1) e = hipEventRecord(e0, main);
      for (int k = 0; k < n && e == hipSuccess && r == ncclSuccess; ++k) {
        e = hipMemsetAsync(a, 0, 8, main);                       // the producer of the message
        if (with == 2) {   // in line: the transfer simply takes its place in the main stream
          if (e == hipSuccess) {
            r = g_rccl.GroupStart();
            if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], main);
            if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], main);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
          }
        } else if (with == 1) {
          if (e == hipSuccess) e = hipEventRecord(c->ready[lane], main);
          if (e == hipSuccess) e = hipStreamWaitEvent(c->side[lane], c->ready[lane], 0);
          if (e == hipSuccess) {
            r = g_rccl.GroupStart();
            if (r == ncclSuccess) r = g_rccl.Send(a, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], c->side[lane]);
            if (r == ncclSuccess) r = g_rccl.Recv(b, (size_t)doubles, ncclDouble, c->rank, c->nccl[lane], c->side[lane]);
            const ncclResult_t r2 = g_rccl.GroupEnd();
            if (r == ncclSuccess) r = r2;
          }
          if (e == hipSuccess && r == ncclSuccess) e = hipEventRecord(c->done[lane], c->side[lane]);
          if (e == hipSuccess && r == ncclSuccess) e = hipStreamWaitEvent(main, c->done[lane], 0);
        }
        if (e == hipSuccess && r == ncclSuccess) e = hipMemsetAsync(b, 0, 8, main);   // its consumer
      }
      if (pass == 1 && e == hipSuccess && r == ncclSuccess) {
        e = hipEventRecord(e1, main);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        float ms = 0.f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        us[with == 1 ? 0 : (with == 0 ? 1 : 2)] = 1e3 * ms / iters;
      } else if (e == hipSuccess) e = hipStreamSynchronize(main);
    }
  }
  if (main) { (void)hipStreamSynchronize(main); (void)hipStreamDestroy(main); }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  (void)hipFree(a); (void)hipFree(b);
  if (r != ncclSuccess) return fail(LBMDEM_EHIP, "RCCL self send/recv failed: %s", g_rccl.GetErrorString(r));
  HIP_TRY(e);
  return LBMDEM_OK;
}
#endif  // LBMDEM_AB

}  // extern "C"
#pragma GCC visibility pop
