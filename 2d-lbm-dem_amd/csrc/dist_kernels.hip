// dist_kernels.hip -- x-strip decomposition with the GRAINS distributed over the ranks (one process per GPU).
//
// The reference has one address space. Under strip decomposition every rank keeps arrays for
// all n grains (global index = array index: the Verlet list, the ascending-partner summation order and every
// formula stay exactly as on one GPU) but INTEGRATES only the grains it owns (centre in its rows) plus a margin on
// either side. Between two fluid steps the npDEM sub-steps run without communication: an error at the outer edge
// of the margin (its grains miss partners that nobody here integrates) travels at most one Verlet-list edge per
// sub-step, and the margin is npDEM edges deep, so it never reaches an owned grain or a grain whose disc touches
// this rank's rows. Once per fluid step the margin is refreshed by the two neighbours (kinematics, then the
// hydrodynamic forces of the same grains); a grain that crosses a cut simply changes owner -- both sides hold its
// exact state, the refresh message is its migration. Nothing here is a collective.
//
// Kernels of this file: ownership / masks / message lists from the current positions, and the packing and
// unpacking of the three point-to-point messages (kinematics, hydrodynamic forces, link-sum tables).

#include "lbmdem_internal.h"

namespace {

// error bits OR-ed into ForceSlots::error
constexpr int ERR_LIST_OVERFLOW = 4;   // more grains near a cut than the message capacity
constexpr int ERR_MERGE_CLASH = 8;     // two ranks produced the same link sum

struct Buf2 { real* p[2]; };
struct CBuf2 { const real* p[2]; };

// kinematics message: {count; count x {id, x1 x2 x3 v1 v2 v3 a1 a2 a3}} in the order of the side's send list
__device__ __forceinline__ void pack_kin_entry(const Kin& K, int i, real* __restrict__ o) {
  o[0] = (real)i;
  o[1] = K.x1[i]; o[2] = K.x2[i]; o[3] = K.x3[i]; o[4] = K.v1[i]; o[5] = K.v2[i]; o[6] = K.v3[i];
  o[7] = K.a1[i]; o[8] = K.a2[i]; o[9] = K.a3[i];
}
__device__ __forceinline__ void pack_kin_side(const int* __restrict__ list, int cnt, const Kin& K, real* __restrict__ buf,
                                              int first, int stride) {
  if (first == 0) buf[0] = (real)cnt;
  for (int k = first; k < cnt; k += stride) pack_kin_entry(K, list[k], buf + 1 + (long)k * 10);
}

// `KB` (C transport): the same launch also writes the two kinematics messages.
__global__ __launch_bounds__(1024) void k_dist_classify(DistDevice D, DistGeom Gm, int n, const real* __restrict__ x1,
                                                       const real* __restrict__ r, const real* __restrict__ rLB,
                                                       unsigned char* __restrict__ owner, int* __restrict__ error,
                                                       int* __restrict__ error_mirror, Kin K, Buf2 KB) {
  __shared__ int sCnt[5], sBase[5];
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  // the other set of counters (the period before used it; every reader is earlier in the stream) is zeroed for the next period
  if (blockIdx.x == 0 && threadIdx.x < 8) D.counters_alt[threadIdx.x] = 0;
  // the error flag of the periods before, as it stands now, to its mirror in pinned host memory (the host looks at it when
  // it opens the NEXT period): a store from here instead of a copy command of its own in the stream
  if (error_mirror && blockIdx.x == 0 && threadIdx.x == 8) *error_mirror = *error;
  const bool in = i < n;
  const bool act = in && D.active[i] != 0;   // integrated during the last period: exact wherever it matters below
  const real xc = in ? (x1[i] - Gm.Mgx) / Gm.dx : 0.0;   // the rasteriser's lattice coordinate of the centre (main.c:1009)
  const bool own = act && (Gm.first || xc >= Gm.lo) && (Gm.last || xc < Gm.hi);
  // grains that may cover nodes of this rank's rows (+ halo) or have links ending there
  const real reach = in ? r[i] / Gm.dx + 3.0 : 0.0;
  const bool near = act && xc + reach >= (real)Gm.gx0 && xc - reach <= (real)(Gm.gx0 + Gm.nxl);
  if (in) {
    owner[i] = own ? 1 : 0;
    D.active[i] = own ? 1 : 0;           // the margin joins when the neighbours' messages are unpacked
    D.fluidmask[i] = near ? 1 : 0;
  }
  // one global atomic per workgroup and list (same-address atomics from hundreds of wavefronts serialise at ~30 ns
  // each on this GPU); the order within a list is immaterial
  const real ring = in ? rLB[i] + 2.0 : 0.0;   // reduced disc + one node: the grain's ring of boundary links
  const bool want[5] = {own && Gm.has_lo && xc < Gm.lo + Gm.margin,                      // kinematics to the low neighbour
                        own && Gm.has_hi && xc >= Gm.hi - Gm.margin,                     // ... the high neighbour
                        act && !own && Gm.has_lo && xc < Gm.lo && xc + ring >= Gm.lo,    // its link sums reach this rank's rows
                        act && !own && Gm.has_hi && xc >= Gm.hi && xc - ring < Gm.hi,
                        near};
  int pos[5];
  if (threadIdx.x < 5) sCnt[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 5; ++w) pos[w] = want[w] ? atomicAdd(&sCnt[w], 1) : 0;
  __syncthreads();
  if (threadIdx.x < 5 && sCnt[threadIdx.x] > 0)
    sBase[threadIdx.x] = atomicAdd(&D.counters[threadIdx.x == 4 ? 6 : threadIdx.x], sCnt[threadIdx.x]);
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 5; ++w) {
    if (!want[w]) continue;
    int* list = w == 4 ? D.local_list : (w < 2 ? D.send_list[w] : D.strad_list[w - 2]);
    const int cap = w == 4 ? D.cap_l : (w < 2 ? D.cap_g : D.cap_t);
    const int k = sBase[w] + pos[w];
    if (k < cap) list[k] = i; else atomicOr(error, ERR_LIST_OVERFLOW);
    // C transport: the grain's entry of the kinematics message goes out with its list slot (same index)
    if (w < 2 && k < cap && KB.p[w]) pack_kin_entry(K, i, KB.p[w] + 1 + (long)k * 10);
  }
  if (!KB.p[0] && !KB.p[1]) return;
  // ... and the workgroup that finishes last writes the two message headers (the final counts). Only atomics cross
  // workgroups here -- this workgroup's additions to the counters have returned (sBase) before it takes its ticket -- so
  // no device-wide fence is needed (a fence means a write-back of the whole L2 on this GPU: ~40 us in this launch, measured).
  __syncthreads();
  if (threadIdx.x == 0 && atomicAdd(&D.counters[7], 1) == (int)gridDim.x - 1) {
    for (int side = 0; side < 2; ++side) {
      if (!KB.p[side]) continue;
      const int cnt = atomicAdd(&D.counters[side], 0);
      KB.p[side][0] = (real)(cnt < D.cap_g ? cnt : D.cap_g);   // an overflow was flagged above
    }
  }
}

// The per-side kernels below take both sides in one launch: blockIdx.y = side (0 low, 1 high); a null buffer
// skips the side.
__global__ void k_pack_kin(DistDevice D, Kin K, Buf2 B) {
  const int side = blockIdx.y;
  if (!B.p[side]) return;
  const int cnt = D.counters[side] < D.cap_g ? D.counters[side] : D.cap_g;   // an overflow is flagged by k_dist_classify
  pack_kin_side(D.send_list[side], cnt, K, B.p[side], blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

__device__ __forceinline__ void unpack_kin_side(const DistDevice& D, int side, const real* __restrict__ buf, const Kin& K,
                                                int n, int* __restrict__ error, int first, int stride) {
  const int cap = D.cap_g;
  unsigned char* __restrict__ active = D.active;
  int* __restrict__ recv_ids = D.recv_ids[side];
  int* __restrict__ recv_count = D.counters + 4 + side;
  int cnt = (int)buf[0];
  if (cnt < 0 || cnt > cap) { cnt = 0; if (first == 0) atomicOr(error, ERR_LIST_OVERFLOW); }
  if (first == 0) *recv_count = cnt;
  for (int k = first; k < cnt; k += stride) {
    const real* o = buf + 1 + (long)k * 10;
    const int i = (int)o[0];
    if (i < 0 || i >= n) continue;
    recv_ids[k] = i;
    K.x1[i] = o[1]; K.x2[i] = o[2]; K.x3[i] = o[3]; K.v1[i] = o[4]; K.v2[i] = o[5]; K.v3[i] = o[6];
    K.a1[i] = o[7]; K.a2[i] = o[8]; K.a3[i] = o[9];
    active[i] = 1;
  }
}

__global__ void k_unpack_kin(DistDevice D, CBuf2 B, Kin K, int n, int* __restrict__ error) {
  const int side = blockIdx.y;
  if (!B.p[side]) return;
  unpack_kin_side(D, side, B.p[side], K, n, error, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

__global__ void k_pack_fhf(DistDevice D, const real* __restrict__ fhf, int n, Buf2 B) {
  const int side = blockIdx.y;
  real* __restrict__ buf = B.p[side];
  if (!buf) return;
  const int* __restrict__ list = D.send_list[side];
  const int cnt = D.counters[side] < D.cap_g ? D.counters[side] : D.cap_g;   // an overflow is flagged by k_dist_classify
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x) {
    const int i = list[k];
    buf[3 * (long)k] = fhf[i]; buf[3 * (long)k + 1] = fhf[n + i]; buf[3 * (long)k + 2] = fhf[2 * (long)n + i];
  }
}

__global__ void k_unpack_fhf(DistDevice D, real* __restrict__ fhf, int n, CBuf2 B) {
  const int side = blockIdx.y;
  const real* __restrict__ buf = B.p[side];
  if (!buf) return;
  const int* __restrict__ ids = D.recv_ids[side];
  const int cnt = D.counters[4 + side];
  for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < cnt; k += gridDim.x * blockDim.x) {
    const int i = ids[k];
    fhf[i] = buf[3 * (long)k]; fhf[n + i] = buf[3 * (long)k + 1]; fhf[2 * (long)n + i] = buf[3 * (long)k + 2];
  }
}

// the neighbour's part of the link-sum tables of grains this rank owns: slot by slot into the local table
// (wave = this wavefront's index among the nwaves that share the side)
__device__ __forceinline__ void merge_tables_side(const ForceSlots& S, const real* __restrict__ buf, int cap, int wave,
                                                  int nwaves, int lane) {
  int cnt = (int)buf[0];
  if (cnt < 0 || cnt > cap) { cnt = 0; if (wave == 0 && lane == 0) atomicOr(S.error, ERR_LIST_OVERFLOW); }
  const int nslot = 8 * S.spd;
  const unsigned long long* b = reinterpret_cast<const unsigned long long*>(buf);
  unsigned long long* tab = reinterpret_cast<unsigned long long*>(S.tab);
  for (int e = wave; e < cnt; e += nwaves) {
    const unsigned long long* ent = b + 1 + (long)e * (1 + nslot);
    const int i = (int)reinterpret_cast<const real*>(ent)[0];
    if (i < 0) continue;   // the sender could not complete this grain (flagged on its side)
    for (int k = lane; k < nslot; k += 64) {
      const unsigned long long v = ent[1 + k];
      if (v == LBMDEM_SLOT_EMPTY) continue;
      unsigned long long* t = tab + (long)i * nslot + k;
      if (*t != LBMDEM_SLOT_EMPTY) atomicOr(S.error, ERR_MERGE_CLASH);
      *t = v;
    }
  }
}

__global__ void k_merge_tables(ForceSlots S, CBuf2 B, int cap) {
  if (!B.p[blockIdx.y]) return;
  merge_tables_side(S, B.p[blockIdx.y], cap, (blockIdx.x * blockDim.x + threadIdx.x) >> 6, (gridDim.x * blockDim.x) >> 6,
                    threadIdx.x & 63);
}

// C transport: everything that follows the arrival of the TABLES message in ONE launch -- the neighbours' table parts
// merged (workgroups [0, 2 MB)), their kinematics messages unpacked ([2 MB, 4 MB): they arrived during the fluid kernels),
// and the obstacle map the fused kernel has just finished with reset for the next rasterisation (the rest; main.c:997-999).
constexpr int UNPACK_MB = 16;   // workgroups per side and message
__global__ __launch_bounds__(256) void k_unpack_tables_kin_fill(ForceSlots S, CBuf2 TB, int cap_t, DistDevice D, CBuf2 KB,
                                                                Kin K, int n, int* __restrict__ dead_obst, LatticeView L) {
  const int b = blockIdx.x;
  if (b < 2 * UNPACK_MB) {
    const int side = b / UNPACK_MB, wb = b % UNPACK_MB;
    if (TB.p[side]) merge_tables_side(S, TB.p[side], cap_t, (wb * 256 + (int)threadIdx.x) >> 6, UNPACK_MB * 4, threadIdx.x & 63);
  } else if (b < 4 * UNPACK_MB) {
    const int side = (b - 2 * UNPACK_MB) / UNPACK_MB, wb = b % UNPACK_MB;
    if (KB.p[side]) unpack_kin_side(D, side, KB.p[side], K, n, S.error, wb * 256 + (int)threadIdx.x, UNPACK_MB * 256);
  } else if (dead_obst) {
    obst_fill_range(dead_obst, L, (long)(b - 4 * UNPACK_MB) * 256 + threadIdx.x, (long)(gridDim.x - 4 * UNPACK_MB) * 256);
  }
}

// diagnostic: grains this rank does not integrate hold NaN -- a use of stale state shows up in the results
__global__ void k_poison(const unsigned char* __restrict__ active, Kin a, Kin b, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || active[i]) return;
  const real q = __longlong_as_double(0x7FF8000000000BADll);
  a.x1[i] = a.x2[i] = a.x3[i] = a.v1[i] = a.v2[i] = a.v3[i] = a.a1[i] = a.a2[i] = a.a3[i] = q;
  b.x1[i] = b.x2[i] = b.x3[i] = b.v1[i] = b.v2[i] = b.v3[i] = b.a1[i] = b.a2[i] = b.a3[i] = q;
}

}  // namespace

int dist_alloc(DistDevice& D, int n, int cap_g, int cap_t, int cap_l) {
  D = DistDevice{};
  D.cap_g = cap_g; D.cap_t = cap_t; D.cap_l = cap_l;
  hipError_t e = hipSuccess;
  auto A = [&](void** p, size_t bytes) { if (e == hipSuccess) e = hipMalloc(p, bytes ? bytes : 16); };
  A((void**)&D.active, n); A((void**)&D.fluidmask, n);
  for (int s = 0; s < 2; ++s) {
    A((void**)&D.send_list[s], sizeof(int) * cap_g);
    A((void**)&D.strad_list[s], sizeof(int) * cap_t);
    A((void**)&D.recv_ids[s], sizeof(int) * cap_g);
  }
  A((void**)&D.local_list, sizeof(int) * cap_l);
  A((void**)&D.counters, sizeof(int) * 16);   // two sets of 8, used by alternate periods
  if (e != hipSuccess) return -1;
  D.counters_alt = D.counters + 8;
  if (hipMemset(D.active, 1, n) != hipSuccess || hipMemset(D.fluidmask, 1, n) != hipSuccess ||
      hipMemset(D.counters, 0, sizeof(int) * 16) != hipSuccess || hipDeviceSynchronize() != hipSuccess)
    return -1;
  return 0;
}

void dist_free(DistDevice& D) {
  void* ps[] = {D.active, D.fluidmask, D.send_list[0], D.send_list[1], D.strad_list[0], D.strad_list[1],
                D.recv_ids[0], D.recv_ids[1], D.counters < D.counters_alt ? D.counters : D.counters_alt, D.local_list};
  for (void* p : ps) if (p) (void)hipFree(p);
  D = DistDevice{};
}

void launch_dist_classify(const DistDevice& D, const DistGeom& Gm, int n, const real* x1, const real* r,
                          const real* rLB, unsigned char* owner, int* error, int* error_mirror, hipStream_t st) {
  // D.counters is all zero here: the classification of the period before cleared it (the received counts [4], [5] are
  // SET by the next unpack). The counts may exceed the capacities (flagged): every reader clamps.
  hipLaunchKernelGGL(k_dist_classify, dim3((n + 1023) / 1024), dim3(1024), 0, st, D, Gm, n, x1, r, rLB, owner, error,
                     error_mirror, Kin{}, Buf2{{nullptr, nullptr}});
}

void launch_dist_classify_pack_kin(const DistDevice& D, const DistGeom& Gm, int n, const real* x1, const real* r,
                                   const real* rLB, unsigned char* owner, int* error, int* error_mirror, const Kin& K,
                                   real* kin_lo, real* kin_hi, hipStream_t st) {
  hipLaunchKernelGGL(k_dist_classify, dim3((n + 1023) / 1024), dim3(1024), 0, st, D, Gm, n, x1, r, rLB, owner, error,
                     error_mirror, K, Buf2{{kin_lo, kin_hi}});
}

void launch_dist_unpack_tables_kin_fill(const ForceSlots& S, const real* tab_lo, const real* tab_hi, int cap_t,
                                        const DistDevice& D, const real* kin_lo, const real* kin_hi, const Kin& K, int n,
                                        int* dead_obst, const LatticeView& L, hipStream_t st) {
  const long fill4 = dead_obst ? (long)L.nxl * L.sy / 4 : 0;
  long fb = (fill4 + 256 * 8 - 1) / (256 * 8);      // eight 16-byte stores per thread
  if (fb > 1024) fb = 1024;
  hipLaunchKernelGGL(k_unpack_tables_kin_fill, dim3(4 * UNPACK_MB + (unsigned)fb), dim3(256), 0, st, S, CBuf2{{tab_lo, tab_hi}},
                     cap_t, D, CBuf2{{kin_lo, kin_hi}}, K, n, dead_obst, L);
}

void launch_dist_pack_kin(const DistDevice& D, const Kin& K, real* lo, real* hi, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_kin, dim3(32, 2), dim3(256), 0, st, D, K, Buf2{{lo, hi}});
}

void launch_dist_unpack_kin(const DistDevice& D, const Kin& K, const real* lo, const real* hi, int n, int* error,
                            hipStream_t st) {
  hipLaunchKernelGGL(k_unpack_kin, dim3(32, 2), dim3(256), 0, st, D, CBuf2{{lo, hi}}, K, n, error);
}

void launch_dist_pack_fhf(const DistDevice& D, const real* fhf, int n, real* lo, real* hi, hipStream_t st) {
  hipLaunchKernelGGL(k_pack_fhf, dim3(32, 2), dim3(256), 0, st, D, fhf, n, Buf2{{lo, hi}});
}

void launch_dist_unpack_fhf(const DistDevice& D, real* fhf, int n, const real* lo, const real* hi, hipStream_t st) {
  hipLaunchKernelGGL(k_unpack_fhf, dim3(32, 2), dim3(256), 0, st, D, fhf, n, CBuf2{{lo, hi}});
}

void launch_dist_merge_tables(const ForceSlots& S, const real* lo, const real* hi, int cap, hipStream_t st) {
  hipLaunchKernelGGL(k_merge_tables, dim3(32, 2), dim3(256), 0, st, S, CBuf2{{lo, hi}}, cap);
}

void launch_dist_poison(const DistDevice& D, const Kin& a, const Kin& b, int n, hipStream_t st) {
  hipLaunchKernelGGL(k_poison, dim3((n + 255) / 256), dim3(256), 0, st, D.active, a, b, n);
}
