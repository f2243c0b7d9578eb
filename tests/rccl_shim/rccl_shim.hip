// rccl_shim.hip -- TEST INFRASTRUCTURE ONLY: a stand-in for librccl.so that lets SEVERAL RANKS SHARE ONE GPU.
//
// Why: the library's own transport (2d-lbm-dem_amd/csrc/lbmdem_comm.hip) dlopen()s RCCL and uses nine of its entry
// points. Real RCCL refuses two ranks on one device ("Duplicate GPU detected"), and the boxes this repo is tested on have
// one GPU -- so without this file lbmdem_comm.hip only ever runs with world == 1, where every neighbour flag is false.
// This shim implements exactly those nine entry points (same signatures, <rccl/rccl.h> types) between processes that all
// sit on the same GPU, so that `lbmdem --gpus 2` and lbmdem_comm_run run their real multi-rank code paths on one device.
// It is never linked into, shipped with, or looked up by the product: tests point the library at it explicitly
// (LBMDEM_RCCL_LIBRARY=<this .so>, or a directory holding it as librccl.so.1 first on LD_LIBRARY_PATH).
//
// How: one POSIX shared-memory segment per communicator (named after the ncclUniqueId, unlinked once every rank has
// mapped it), pinned and mapped into each process's GPU address space (hipHostRegister). For every ordered pair of
// ranks the segment holds a one-slot channel {flag, ack, data[CHUNK]}.
//   ncclSend  = one kernel on the caller's stream: wait until the receiver acknowledged the previous chunk, copy the
//               chunk into the channel, system-scope fence, publish flag = sequence number;
//   ncclRecv  = one kernel on the caller's stream: wait for flag == sequence number, copy the chunk out, publish ack.
// Both are STREAM-ORDERED like the real thing (nothing blocks the host), messages larger than a chunk take several
// rounds, and inside a group the rounds of all sends and receives are interleaved (send round k, receive round k, ...)
// so that a grouped exchange between two ranks cannot wait on itself. A kernel that waits longer than
// RCCL_SHIM_TIMEOUT_S seconds (default 20) gives up, raises the segment's error word and every later call of every rank
// fails: an exchange issued in the wrong order FAILS THE TEST instead of hanging the GPU.
//   ncclAllReduce (ncclSum; used off the step path only) synchronises the stream and reduces through the segment on
//   the host, chunk by chunk, with a host barrier.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <errno.h>
#include <fcntl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <vector>

namespace {

constexpr size_t CHUNK = 128u << 10;      // bytes per channel slot (8 ranks x 4 communicators stay below 40 MB of /dev/shm)
constexpr int MAX_RANKS = 8;

struct alignas(128) Word { volatile uint32_t v; };

struct Channel {      // src -> dst
  Word flag;          // sequence number of the chunk src published last
  Word ack;           // sequence number of the chunk dst consumed last
  alignas(128) unsigned char data[CHUNK];
};

struct Header {
  Word arrived;             // ranks that mapped the segment
  Word error;               // != 0: a device-side wait timed out (value = 1 + waiting rank)
  Word bar_count, bar_gen;  // host barrier of the collectives
};

// layout of the shared file: Header; Channel chan[W * W]; unsigned char slot[W][CHUNK] (host-side all-reduce)
size_t segment_bytes(int W) { return sizeof(Header) + sizeof(Channel) * (size_t)W * W + CHUNK * (size_t)W; }

double now_s() {
  timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return t.tv_sec + 1e-9 * t.tv_nsec;
}

double timeout_s() {
  const char* e = getenv("RCCL_SHIM_TIMEOUT_S");
  const double v = e ? atof(e) : 0.0;
  return v > 0 ? v : 20.0;
}

thread_local char g_msg[256] = "no error";

}  // namespace

struct ncclComm {
  int rank = 0, world = 1;
  unsigned char* base = nullptr;    // host mapping
  unsigned char* dbase = nullptr;   // the same bytes as the GPU sees them
  size_t bytes = 0;
  uint32_t send_seq[MAX_RANKS] = {}, recv_seq[MAX_RANKS] = {};   // per peer, counted by this process
  long long wait_ticks = 0;
  Header* hdr() const { return reinterpret_cast<Header*>(base); }
  Channel* chan_host(int src, int dst) const { return reinterpret_cast<Channel*>(base + sizeof(Header)) + (size_t)src * world + dst; }
  Channel* chan_dev(int src, int dst) const { return reinterpret_cast<Channel*>(dbase + sizeof(Header)) + (size_t)src * world + dst; }
  unsigned char* slot(int r) const { return base + sizeof(Header) + sizeof(Channel) * (size_t)world * world + CHUNK * (size_t)r; }
  Word* err_dev() const { return &reinterpret_cast<Header*>(dbase)->error; }
};

namespace {

// ---- device side -------------------------------------------------------------------------------------------------
__device__ bool wait_for(volatile uint32_t* w, uint32_t want, long long limit_ticks, volatile uint32_t* err, uint32_t who) {
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(const_cast<uint32_t*>(w), __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != want) {
    if (__hip_atomic_load(const_cast<uint32_t*>(err), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0) return false;   // somebody else gave up
    if (wall_clock64() - t0 > limit_ticks) {
      __hip_atomic_store(const_cast<uint32_t*>(err), who, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      return false;
    }
    __builtin_amdgcn_s_sleep(32);
  }
  return true;
}

__device__ void copy_bytes(unsigned char* dst, const unsigned char* src, size_t n) {
  // both ends are at least 8-byte aligned in every use of this shim (hipMalloc'ed buffers, 128-byte aligned slots)
  const size_t n8 = n / 8;
  unsigned long long* d8 = reinterpret_cast<unsigned long long*>(dst);
  const unsigned long long* s8 = reinterpret_cast<const unsigned long long*>(src);
  for (size_t k = threadIdx.x; k < n8; k += blockDim.x) d8[k] = s8[k];
  for (size_t k = n8 * 8 + threadIdx.x; k < n; k += blockDim.x) dst[k] = src[k];
}

__global__ __launch_bounds__(1024) void k_shim_send(Channel* ch, uint32_t seq, const unsigned char* src, size_t n,
                                                     long long limit, volatile uint32_t* err, uint32_t who) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = wait_for(&ch->ack.v, seq - 1, limit, err, who) ? 1 : 0;   // the slot is free again
  __syncthreads();
  if (!ok) return;
  copy_bytes(ch->data, src, n);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint32_t*>(&ch->flag.v), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(1024) void k_shim_recv(Channel* ch, uint32_t seq, unsigned char* dst, size_t n,
                                                     long long limit, volatile uint32_t* err, uint32_t who) {
  __shared__ int ok;
  if (threadIdx.x == 0) ok = wait_for(&ch->flag.v, seq, limit, err, who) ? 1 : 0;
  __syncthreads();
  if (!ok) return;
  __threadfence_system();
  copy_bytes(dst, ch->data, n);
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_store(const_cast<uint32_t*>(&ch->ack.v), seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---- host side ---------------------------------------------------------------------------------------------------
struct Op { bool send; ncclComm* c; unsigned char* buf; size_t bytes; int peer; hipStream_t st; };
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

ncclResult_t check(ncclComm* c) {
  if (!c || !c->base) { snprintf(g_msg, sizeof g_msg, "rccl shim: null communicator"); return ncclInvalidArgument; }
  const uint32_t e = c->hdr()->error.v;
  if (e != 0) {
    snprintf(g_msg, sizeof g_msg, "rccl shim: rank %u waited longer than %.0f s for its peer inside a send/recv kernel "
             "(an exchange was issued in the wrong order, or a rank died)", e - 1, timeout_s());
    return ncclSystemError;
  }
  return ncclSuccess;
}

ncclResult_t flush(std::vector<Op>& ops) {
  size_t rounds = 0;
  for (const Op& o : ops) { const size_t r = (o.bytes + CHUNK - 1) / CHUNK; if (r > rounds) rounds = r; }
  for (size_t k = 0; k < rounds; ++k)
    for (int pass = 0; pass < 2; ++pass)            // all sends of round k, then all receives of round k
      for (const Op& o : ops) {
        if (o.send != (pass == 0)) continue;
        const size_t off = k * CHUNK;
        if (off >= o.bytes) continue;
        const size_t n = o.bytes - off < CHUNK ? o.bytes - off : CHUNK;
        ncclComm* c = o.c;
        if (o.send) {
          const uint32_t seq = ++c->send_seq[o.peer];
          hipLaunchKernelGGL(k_shim_send, dim3(1), dim3(1024), 0, o.st, c->chan_dev(c->rank, o.peer), seq, o.buf + off, n,
                             c->wait_ticks, &c->err_dev()->v, (uint32_t)(1 + c->rank));
        } else {
          const uint32_t seq = ++c->recv_seq[o.peer];
          hipLaunchKernelGGL(k_shim_recv, dim3(1), dim3(1024), 0, o.st, c->chan_dev(o.peer, c->rank), seq, o.buf + off, n,
                             c->wait_ticks, &c->err_dev()->v, (uint32_t)(1 + c->rank));
        }
        if (hipGetLastError() != hipSuccess) { snprintf(g_msg, sizeof g_msg, "rccl shim: kernel launch failed"); return ncclUnhandledCudaError; }
      }
  return ncclSuccess;
}

ncclResult_t enqueue(bool send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm* c, hipStream_t st) {
  ncclResult_t r = check(c);
  if (r != ncclSuccess) return r;
  const size_t tb = type_bytes(t);
  if (!buf || tb == 0 || peer < 0 || peer >= c->world) { snprintf(g_msg, sizeof g_msg, "rccl shim: bad send/recv arguments"); return ncclInvalidArgument; }
  Op o{send, c, const_cast<unsigned char*>(static_cast<const unsigned char*>(buf)), count * tb, peer, st};
  if (g_depth > 0) { g_ops.push_back(o); return ncclSuccess; }
  std::vector<Op> one{o};
  return flush(one);
}

// barrier of the ranks' HOST threads (collectives only)
ncclResult_t host_barrier(ncclComm* c) {
  Header* h = c->hdr();
  const uint32_t gen = __atomic_load_n(&h->bar_gen.v, __ATOMIC_ACQUIRE);
  if (__atomic_add_fetch(&h->bar_count.v, 1, __ATOMIC_ACQ_REL) == (uint32_t)c->world) {
    __atomic_store_n(&h->bar_count.v, 0, __ATOMIC_RELAXED);
    __atomic_store_n(&h->bar_gen.v, gen + 1, __ATOMIC_RELEASE);
    return ncclSuccess;
  }
  const double t0 = now_s(), lim = 6 * timeout_s();
  while (__atomic_load_n(&h->bar_gen.v, __ATOMIC_ACQUIRE) == gen) {
    if (now_s() - t0 > lim) { snprintf(g_msg, sizeof g_msg, "rccl shim: host barrier timed out (rank %d)", c->rank); return ncclSystemError; }
    sched_yield();
  }
  return ncclSuccess;
}

template <typename T>
void add_into(void* acc, const void* x, size_t n) {
  T* a = static_cast<T*>(acc);
  const T* b = static_cast<const T*>(x);
  for (size_t k = 0; k < n; ++k) a[k] = a[k] + b[k];
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

const char* ncclGetErrorString(ncclResult_t r) { return r == ncclSuccess ? "no error" : g_msg; }

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof *id);
  static int counter = 0;
  uint64_t rnd = 0;
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd >= 0) { if (read(fd, &rnd, sizeof rnd) != (ssize_t)sizeof rnd) rnd = 0; close(fd); }
  snprintf(id->internal, sizeof id->internal, "/rcclshim-%d-%d-%016llx", (int)getpid(), counter++, (unsigned long long)rnd);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->base) {
    (void)hipHostUnregister(c->base);
    munmap(c->base, c->bytes);
  }
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks || id.internal[0] != '/') {
    snprintf(g_msg, sizeof g_msg, "rccl shim: bad ncclCommInitRank arguments (at most %d ranks)", MAX_RANKS);
    return ncclInvalidArgument;
  }
  id.internal[sizeof id.internal - 1] = 0;
  const size_t bytes = segment_bytes(nranks);
  int fd = shm_open(id.internal, O_CREAT | O_RDWR, 0600);
  if (fd < 0) { snprintf(g_msg, sizeof g_msg, "rccl shim: shm_open(%s): %s", id.internal, strerror(errno)); return ncclSystemError; }
  if (ftruncate(fd, (off_t)bytes) != 0) { snprintf(g_msg, sizeof g_msg, "rccl shim: ftruncate: %s", strerror(errno)); close(fd); return ncclSystemError; }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { snprintf(g_msg, sizeof g_msg, "rccl shim: mmap: %s", strerror(errno)); return ncclSystemError; }
  ncclComm* c = new ncclComm();
  c->rank = rank; c->world = nranks; c->base = static_cast<unsigned char*>(p); c->bytes = bytes;
  c->wait_ticks = (long long)(timeout_s() * 1e8);   // wall_clock64 counts at 100 MHz
  // every rank has mapped the (zero-filled) segment before anybody uses it; then the name can go
  Header* h = c->hdr();
  __atomic_add_fetch(&h->arrived.v, 1, __ATOMIC_ACQ_REL);
  const double t0 = now_s(), lim = 6 * timeout_s();
  while (__atomic_load_n(&h->arrived.v, __ATOMIC_ACQUIRE) < (uint32_t)nranks) {
    if (now_s() - t0 > lim) {
      snprintf(g_msg, sizeof g_msg, "rccl shim: rank %d waited %.0f s for the other ranks in ncclCommInitRank", rank, lim);
      shm_unlink(id.internal); munmap(p, bytes); delete c;
      return ncclSystemError;
    }
    usleep(200);
  }
  if (host_barrier(c) != ncclSuccess) { shm_unlink(id.internal); munmap(p, bytes); delete c; return ncclSystemError; }
  if (rank == 0) shm_unlink(id.internal);
  hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped | hipHostRegisterPortable);
  void* dp = nullptr;
  if (e == hipSuccess) e = hipHostGetDevicePointer(&dp, p, 0);
  if (e != hipSuccess) {
    snprintf(g_msg, sizeof g_msg, "rccl shim: cannot map the shared segment into the GPU: %s", hipGetErrorString(e));
    munmap(p, bytes); delete c;
    return ncclUnhandledCudaError;
  }
  c->dbase = static_cast<unsigned char*>(dp);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { ++g_depth; return ncclSuccess; }

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) { snprintf(g_msg, sizeof g_msg, "rccl shim: ncclGroupEnd without ncclGroupStart"); return ncclInvalidUsage; }
  if (--g_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(g_ops);
  return flush(ops);
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return enqueue(true, buf, count, t, peer, c, st);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t c, hipStream_t st) {
  return enqueue(false, buf, count, t, peer, c, st);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t t, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t st) {
  ncclResult_t r = check(c);
  if (r != ncclSuccess) return r;
  const size_t tb = type_bytes(t);
  if (!sendbuff || !recvbuff || op != ncclSum || (tb != 8 && tb != 4)) {
    snprintf(g_msg, sizeof g_msg, "rccl shim: ncclAllReduce supports ncclSum over 4- and 8-byte types");
    return ncclInvalidArgument;
  }
  if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
  const size_t total = count * tb;
  std::vector<unsigned char> acc(CHUNK);
  for (size_t off = 0; off < total; off += CHUNK) {
    const size_t n = total - off < CHUNK ? total - off : CHUNK;
    if (hipMemcpy(c->slot(c->rank), static_cast<const unsigned char*>(sendbuff) + off, n, hipMemcpyDeviceToHost) != hipSuccess)
      return ncclUnhandledCudaError;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
    if ((r = host_barrier(c)) != ncclSuccess) return r;
    memcpy(acc.data(), c->slot(0), n);
    for (int k = 1; k < c->world; ++k) {   // rank order: every rank forms the same sum
      switch (t) {
        case ncclFloat64: add_into<double>(acc.data(), c->slot(k), n / 8); break;
        case ncclInt64: case ncclUint64: add_into<uint64_t>(acc.data(), c->slot(k), n / 8); break;
        case ncclFloat32: add_into<float>(acc.data(), c->slot(k), n / 4); break;
        default: add_into<uint32_t>(acc.data(), c->slot(k), n / 4); break;
      }
    }
    if ((r = host_barrier(c)) != ncclSuccess) return r;   // nobody overwrites a slot that is still being read
    if (hipMemcpy(static_cast<unsigned char*>(recvbuff) + off, acc.data(), n, hipMemcpyHostToDevice) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

#pragma GCC visibility pop
}  // extern "C"
