// Context, communicator and the host side of the two-stage deterministic reductions.
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <execinfo.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <algorithm>

#include "core.hpp"

namespace po {

static thread_local char g_err[1024] = "";
thread_local int g_last_code = 0;

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  fprintf(stderr, "paropt_amd: %s\n", g_err);
}
const char *last_error() { return g_err; }

static int g_dbg_switch[SW_COUNT] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
double host_now() {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
int dbg_switch(int id, const char *env, int dflt) {
  if (id >= 0 && id < SW_COUNT && g_dbg_switch[id] >= 0) return g_dbg_switch[id];
  const char *e = env ? getenv(env) : nullptr;
  return e ? atoi(e) : dflt;
}
void dbg_switch_set(int id, int value) {
  if (id >= 0 && id < SW_COUNT) g_dbg_switch[id] = value;
}

int launch_reduce_final(Ctx *c, int nblocks, int nslots, int nsum, int nmin, int dst_off);
int launch_reduce_final_multi(Ctx *c, const Ctx::PendingRed *pend, int count);
int launch_red_publish(Ctx *c, const double *src, int count);
// (Re)allocates the pinned result buffer and the alias the device writes through (see Ctx::h_red_dev).
static int alloc_h_red(Ctx *c, size_t doubles) {
  if (c->h_red) (void)hipHostFree(c->h_red);
  c->h_red = nullptr;
  c->h_red_dev = nullptr;
  // Coherent (fine-grained), mapped pinned memory, asked for explicitly: a kernel's system-scope stores there are
  // visible to a polling host thread without a stream synchronisation (the completion flag below relies on it)
  if (hipHostMalloc((void **)&c->h_red, sizeof(double) * doubles, hipHostMallocMapped | hipHostMallocCoherent) !=
      hipSuccess) {
    (void)hipGetLastError();
    c->h_red = nullptr;
    PO_HIP(hipHostMalloc((void **)&c->h_red, sizeof(double) * doubles, hipHostMallocDefault));
    c->h_red_coherent = 0;
  } else {
    c->h_red_coherent = 1;
  }
  if (getenv("PAROPT_AMD_NO_DIRECT_RED") ||
      hipHostGetDevicePointer((void **)&c->h_red_dev, c->h_red, 0) != hipSuccess) {
    (void)hipGetLastError();
    c->h_red_dev = nullptr;
  }
  // A reallocation that lost the coherence guarantee (or the device alias) takes the flag away with it: results
  // behind a flag the host has seen must not sit in memory that may be read stale.  Nothing is in
  // flight here: the buffer is resized at communicator set-up only.
  if (c->h_flag && (!c->h_red_coherent || !c->h_red_dev)) {
    (void)hipHostFree(c->h_flag);
    c->h_flag = c->h_flag_dev = nullptr;
    if (c->d_ticket) (void)hipFree(c->d_ticket);
    c->d_ticket = nullptr;
  }
  // the completion flag of the final reduction stages (see Ctx::h_flag): allocated once
  // (without the explicit coherence guarantee the flag is not used at all: results behind a flag the host has seen
  // could themselves be stale)
  if (c->h_red_dev && !c->h_flag && c->h_red_coherent && !getenv("PAROPT_AMD_NO_FLAG_POLL")) {
    if (hipHostMalloc((void **)&c->h_flag, 64, hipHostMallocMapped | hipHostMallocCoherent) == hipSuccess &&
        hipHostGetDevicePointer((void **)&c->h_flag_dev, c->h_flag, 0) == hipSuccess &&
        hipMalloc((void **)&c->d_ticket, sizeof(unsigned)) == hipSuccess &&
        hipMemset(c->d_ticket, 0, sizeof(unsigned)) == hipSuccess) {
      *c->h_flag = 0;
    } else {
      (void)hipGetLastError();
      if (c->h_flag) (void)hipHostFree(c->h_flag);
      if (c->d_ticket) (void)hipFree(c->d_ticket);
      c->h_flag = c->h_flag_dev = nullptr;
      c->d_ticket = nullptr;
    }
  }
  return PO_OK;
}


static int comm_self_test(Ctx *c);
static void free_overflow(Ctx *c, const double *keep);

// ---- RCCL through dlopen: the library is only needed for world sizes > 1 ----------------------
struct Id128 {  // ncclUniqueId: 128 opaque bytes passed BY VALUE to ncclCommInitRank
  char b[PO_RCCL_ID_BYTES];
};
struct RcclApi {
  void *handle = nullptr;
  int (*GetUniqueId)(void *) = nullptr;
  int (*CommInitRank)(void **, int, Id128, int) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
  int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
  int (*CommDestroy)(void *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GetVersion)(int *) = nullptr;
  int version = 0;
};
static RcclApi g_rccl;

static int load_rccl() {
  if (g_rccl.handle) return PO_OK;
  // PAROPT_AMD_RCCL_LIB names the library file explicitly (a C++ program that does not run under torch's loader
  // finds RCCL on the default search path or under /opt/rocm/lib only)
  const char *names[] = {getenv("PAROPT_AMD_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
  void *h = nullptr;
  for (const char *nm : names) {
    if (!nm || !*nm) continue;
    h = dlopen(nm, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) {
    set_error("cannot dlopen librccl.so: %s", dlerror());
    return PO_ERR_COMM;
  }
  g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
  g_rccl.CommInitRank = (int (*)(void **, int, Id128, int))dlsym(h, "ncclCommInitRank");
  g_rccl.AllGather =
      (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
  g_rccl.AllReduce =
      (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclAllReduce");
  g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
  g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
  g_rccl.GetVersion = (int (*)(int *))dlsym(h, "ncclGetVersion");
  if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.AllReduce ||
      !g_rccl.CommDestroy) {
    set_error("librccl.so lacks an expected symbol");
    dlclose(h);
    return PO_ERR_COMM;
  }
  // The prototypes above are hand-declared (no rccl.h at build time).  What they assume -- ncclUniqueId is 128 opaque
  // bytes passed BY VALUE to ncclCommInitRank, ncclDouble == 8, ncclSum == 0 -- holds for every NCCL/RCCL 2.x release from 2.10 on; anything else is refused here
  // instead of being called with a wrong layout.
  int version = 0;
  if (!g_rccl.GetVersion || g_rccl.GetVersion(&version) != 0) {
    set_error("librccl.so: ncclGetVersion unavailable; cannot verify the ABI the hand-declared prototypes assume");
    dlclose(h);
    return PO_ERR_COMM;
  }
  // version code: major * 10000 + minor * 100 + patch from 2.9 on (major * 1000 + ... before)
  const int major = version >= 10000 ? version / 10000 : version / 1000;
  const int minor = version >= 10000 ? (version / 100) % 100 : (version / 100) % 10;
  if (major != 2 || minor < 10) {
    set_error("librccl.so reports version code %d (need NCCL/RCCL 2.x, x >= 10: 128-byte by-value unique id, ncclDouble == 8)",
              version);
    dlclose(h);
    return PO_ERR_COMM;
  }
  g_rccl.version = version;
  g_rccl.handle = h;
  return PO_OK;
}

int rccl_version() { return g_rccl.version; }

int rccl_unique_id(void *id128) {
  PO_TRY(load_rccl());
  Id128 id;
  memset(&id, 0, sizeof(id));
  int rc = g_rccl.GetUniqueId(&id);
  if (rc != 0) {
    set_error("ncclGetUniqueId failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return PO_ERR_COMM;
  }
  memcpy(id128, &id, sizeof(id));
  return PO_OK;
}

int comm_init_rccl(Ctx *c, int rank, int size, const void *id128) {
  if (size < 1 || rank < 0 || rank >= size) {
    set_error("bad rank/size %d/%d", rank, size);
    return PO_ERR_ARG;
  }
  c->rank = rank;
  c->size = size;
  // PAROPT_AMD_FORCE_RCCL=1 routes even a single rank through ncclCommInitRank / ncclAllGather, so
  // the RCCL plumbing (dlopen, by-value unique id, stream use) can be exercised on a 1-GPU box
  if (size == 1 && !getenv("PAROPT_AMD_FORCE_RCCL")) {
    c->comm_kind = COMM_SELF;
    return PO_OK;
  }
  PO_TRY(load_rccl());
  PO_HIP(hipSetDevice(c->device));
  Id128 id;
  memcpy(&id, id128, sizeof(id));
  void *comm = nullptr;
  int rc = g_rccl.CommInitRank(&comm, size, id, rank);
  if (rc != 0) {
    set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
    return PO_ERR_COMM;
  }
  c->rccl_comm = comm;
  c->comm_kind = COMM_RCCL;
  // PAROPT_AMD_RCCL_ALLGATHER=1: every reduction through the rank-ordered all-gather (bit-identical across
  // rank-to-GPU placements); default: pure-sum payloads use ncclAllReduce, mixed SUM/MIN/MAX the all-gather
  c->rccl_allreduce = getenv("PAROPT_AMD_RCCL_ALLGATHER") ? 0 : 1;
  // gather buffers sized for the communicator
  if (c->d_gather) (void)hipFree(c->d_gather);
  PO_HIP(hipMalloc((void **)&c->d_gather, sizeof(double) * (size_t)size * kMaxRed));
  PO_TRY(alloc_h_red(c, (size_t)size * kMaxRed));
  // The first collectives a fresh communicator runs are known-answer ones: a wrong enum value, a wrong rank order or
  // a communicator that does not span the ranks it claims fails HERE with a message, not as a diverging solve.
  const int rc_test = comm_self_test(c);
  if (rc_test != PO_OK) {
    g_rccl.CommDestroy(c->rccl_comm);
    c->rccl_comm = nullptr;
    c->comm_kind = COMM_SELF;
    c->size = 1;
    c->rank = 0;
    return rc_test;
  }
  return PO_OK;
}

int comm_init_callback(Ctx *c, int rank, int size, po_allgather_fn fn, void *user) {
  if (size < 1 || rank < 0 || rank >= size || (size > 1 && !fn)) {
    set_error("bad callback communicator %d/%d", rank, size);
    return PO_ERR_ARG;
  }
  c->rank = rank;
  c->size = size;
  c->cb_allgather = fn;
  c->cb_user = user;
  c->comm_kind = size > 1 ? COMM_CALLBACK : COMM_SELF;
  PO_TRY(alloc_h_red(c, (size_t)(size + 1) * kMaxRed));
  return PO_OK;
}

// development aid: see Ctx::host_trace (contexts that are never destroyed report at exit)
static std::vector<Ctx *> g_traced;
static void host_trace_report(Ctx *c) {
  if (c->rank != 0) return;
  fprintf(stderr,
          "paropt_amd host trace: %ld synchronising reductions; host time between a result and the next launch "
          "%.1f us avg (%ld gaps below 500 us, %ld of them above 20 us, %.3f s; %ld longer ones, %.3f s); inside launch "
          "calls %.2f us avg (%ld launches, %.3f s); waiting for results %.3f s; host time between two launches without "
          "a synchronisation in between %.1f us avg (%ld intervals, %ld above 30 us: %.3f s)\n",
          c->n_reductions, c->host_gap_n ? 1e6 * c->host_gap_s / c->host_gap_n : 0.0, c->host_gap_n, c->host_gap_n20,
          c->host_gap_s, c->host_gap_long_n, c->host_gap_long_s,
          c->host_launch_n ? 1e6 * c->host_launch_s / c->host_launch_n : 0.0, c->host_launch_n, c->host_launch_s,
          c->host_wait_s, c->host_inter_n ? 1e6 * c->host_inter_s / c->host_inter_n : 0.0, c->host_inter_n,
          c->host_inter_n30, c->host_inter_s30);
}
static void host_trace_atexit() {
  for (Ctx *c : g_traced)
    if (c) host_trace_report(c);
}
int ctx_create(int device, Ctx **out) {
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev == 0) {
    set_error("no HIP device available (%s): paropt_amd has no CPU fallback",
              e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    return PO_ERR_NO_DEVICE;
  }
  if (device < 0 || device >= ndev) {
    set_error("device %d out of range (have %d)", device, ndev);
    return PO_ERR_ARG;
  }
  PO_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  PO_HIP(hipGetDeviceProperties(&prop, device));
  po_ctx_s *c = new po_ctx_s();
  c->device = device;
  c->host_trace = getenv("PAROPT_AMD_HOST_TRACE") != nullptr;
  if (c->host_trace) {
    if (g_traced.empty()) atexit(host_trace_atexit);
    g_traced.push_back(c);
  }
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->max_blocks = c->num_cu * 8;  // upper bound used only to size the partials buffer
  PO_HIP(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking));
  PO_HIP(hipMalloc((void **)&c->d_red, sizeof(double) * kMaxRed));
  PO_TRY(alloc_h_red(c, kMaxRed));
  PO_HIP(hipEventCreate(&c->ev0));
  PO_HIP(hipEventCreate(&c->ev1));
  PO_HIP(hipEventCreate(&c->ev_mdot0));
  PO_HIP(hipEventCreate(&c->ev_mdot1));
  c->partials_cap = 0;
  if (getenv("PAROPT_AMD_NO_BATCH")) c->batch_enabled = 0;
  *out = c;
  return ensure_partials(c, (size_t)c->max_blocks * 64);
}

int ctx_destroy(Ctx *c) {
  if (!c) return PO_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->stream);
  if (c->host_trace) {
    for (auto &p : g_traced)
      if (p == c) p = nullptr;
    host_trace_report(c);
  }
  if (c->rccl_comm && g_rccl.CommDestroy) g_rccl.CommDestroy(c->rccl_comm);
  free_overflow(c, nullptr);
  if (c->partials_base) (void)hipFree(c->partials_base);
  if (c->d_red) (void)hipFree(c->d_red);
  if (c->d_gather) (void)hipFree(c->d_gather);
  if (c->h_red) (void)hipHostFree(c->h_red);
  if (c->h_flag) (void)hipHostFree(c->h_flag);
  if (c->d_ticket) (void)hipFree(c->d_ticket);
  for (int i = 0; i < 2; i++)
    if (c->wide_scratch[i]) (void)hipFree(c->wide_scratch[i]);
  if (c->ev0) (void)hipEventDestroy(c->ev0);
  if (c->ev1) (void)hipEventDestroy(c->ev1);
  if (c->ev_mdot0) (void)hipEventDestroy(c->ev_mdot0);
  if (c->ev_mdot1) (void)hipEventDestroy(c->ev_mdot1);
  (void)hipStreamDestroy(c->stream);
  delete static_cast<po_ctx_s *>(c);
  return PO_OK;
}

static int grow_partials(Ctx *c, size_t doubles) {
  // growth happens outside hot loops in practice (first use of a wider kernel); nothing is queued when it happens
  PO_HIP(hipStreamSynchronize(c->stream));
  if (c->partials_base) PO_HIP(hipFree(c->partials_base));
  c->partials_base = nullptr;
  size_t cap = doubles + doubles / 4;
  if (cap < ((size_t)1 << 22)) cap = (size_t)1 << 22;  // 32 MB: room for the regions of a whole batch
  PO_HIP(hipMalloc((void **)&c->partials_base, cap * sizeof(double)));
  c->partials_cap = cap;
  c->partials_cursor = 0;
  return PO_OK;
}
// Region for the first-stage partials of the launch about to be issued (c->d_partials).  Outside a batch: the arena
// itself.  Inside a batch: a region of its own, so that the final stage can wait for the flush.  When the arena is
// full the region is a chunk of its own (freed after the flush): whether THIS rank's arena is full depends on its
// local grid sizes, so a flush here -- a collective -- could be issued by one rank and not by its peers.
static void free_overflow(Ctx *c, const double *keep) {
  std::vector<double *> kept;
  for (double *p : c->partials_overflow) {
    if (p == keep) {
      kept.push_back(p);
    } else {
      (void)hipFree(p);
    }
  }
  c->partials_overflow.swap(kept);
}
int ensure_partials(Ctx *c, size_t doubles) {
  const bool queued = c->batch_depth > 0 || !c->batch_pend.empty();
  if (!queued) {
    if (doubles > c->partials_cap) PO_TRY(grow_partials(c, doubles));
    c->partials_cursor = 0;
    c->d_partials = c->partials_base;
    return PO_OK;
  }
  const size_t need = (doubles + 63) & ~(size_t)63;  // 512-byte granules
  if (c->partials_cursor + need > c->partials_cap) {
    double *chunk = nullptr;
    PO_HIP(hipMalloc((void **)&chunk, need * sizeof(double)));
    c->partials_overflow.push_back(chunk);
    c->d_partials = chunk;
    c->partials_last = need;
    return PO_OK;
  }
  c->d_partials = c->partials_base + c->partials_cursor;
  c->partials_cursor += need;
  c->partials_last = need;
  return PO_OK;
}

// One collective + device-to-host copy + host sync for `total` rank-local values in d_red[0..total); on return
// *parts points at nparts consecutive copies (one per contributing rank, rank order; 1 when the collective has
// already summed them) in h_red.
// One place decides where the final reduction stages write: without a device-side collective (one rank, or the
// host-callback communicator) straight into the pinned host buffer -- no device-to-host copy command before the sync.
bool red_direct(const Ctx *c) { return c->comm_kind != COMM_RCCL && c->h_red_dev != nullptr; }

// Host side of the completion flag: poll the sequence number the last flagged launch raises behind its results.
// Bounded (200 ms of spinning with pause instructions, far beyond any queue of kernels ahead of the final stage);
// past the bound the stream is synchronised -- always correct, and counted (Ctx::n_flag_timeouts, po_ctx_sync_counters)
// so that a flag that is never raised shows up in a soak test instead of degrading silently.
static int wait_results(Ctx *c) {
  volatile unsigned long long *f = c->h_flag;
  const unsigned long long want = c->red_seq;
  bool seen = false;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  c->n_flag_waits++;
  for (long spin = 0;; spin++) {
    if (*f == want) {
      seen = true;
      break;
    }
#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
    __builtin_ia32_pause();
#endif
    if ((spin & 1023) == 1023) {
      struct timespec t1;
      clock_gettime(CLOCK_MONOTONIC, &t1);
      if ((double)(t1.tv_sec - t0.tv_sec) + 1e-9 * (double)(t1.tv_nsec - t0.tv_nsec) > 0.2) break;
    }
  }
  __atomic_thread_fence(__ATOMIC_ACQUIRE);  // the results in h_red are read after the flag, also by the compiler
  c->red_seq_seen = want;
  if (!seen) {
    c->n_flag_timeouts++;
    PO_HIP(hipStreamSynchronize(c->stream));
  }
  return PO_OK;
}

static int exchange_reduced(Ctx *c, int total, bool pure_sum, const double **parts_out, int *nparts_out) {
  const size_t bytes = sizeof(double) * (size_t)total;
  int nparts = 1;
  const double *parts = c->h_red;
  c->n_reductions++;
  const double ht0 = c->host_trace ? host_now() : 0.0;
  static const bool trace = getenv("PAROPT_AMD_SYNC_TRACE") != nullptr;  // development aid: who synchronises?
  if (trace) {
    void *frames[12];
    const int nf = backtrace(frames, 12);
    fprintf(stderr, "== host sync %ld (%d values)\n", c->n_reductions, total);
    backtrace_symbols_fd(frames + 1, nf > 1 ? nf - 1 : 0, 2);
  }
  const bool can_poll = c->h_red_dev && c->h_flag_dev && c->h_flag;
  if (c->comm_kind == COMM_RCCL && c->rccl_allreduce && pure_sum) {
    // the MPI_Allreduce(SUM) sites of the reference (dot/mdot/Gram entries, src/ParOptVec.cpp:124-170,
    // src/ParOptInteriorPoint.cpp:1957) -> ONE ncclAllReduce over xGMI on the solver's stream, in place in d_red;
    // every rank receives the same bits (the reduced value is produced once per chunk and broadcast)
    int rc = g_rccl.AllReduce(c->d_red, c->d_red, (size_t)total, /*ncclDouble*/ 8, /*ncclSum*/ 0, c->rccl_comm,
                              c->stream);
    if (rc != 0) {
      set_error("ncclAllReduce failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
      return PO_ERR_COMM;
    }
    c->n_allreduce++;
    // results -> pinned host memory + completion flag by a one-workgroup kernel behind the collective, polled by the
    // host (round 5: the completion path of the single-rank runs; it replaces copy command + stream synchronisation
    // exactly where the exchanges ARE the scaling loss, at n / 8 per rank)
    if (can_poll) {
      PO_TRY(launch_red_publish(c, c->d_red, total));
      PO_TRY(wait_results(c));
    } else {
      PO_HIP(hipMemcpyAsync(c->h_red, c->d_red, bytes, hipMemcpyDeviceToHost, c->stream));
      PO_HIP(hipStreamSynchronize(c->stream));
    }
  } else if (c->comm_kind == COMM_RCCL) {
    c->n_allgather++;
    int rc = g_rccl.AllGather(c->d_red, c->d_gather, (size_t)total, /*ncclDouble*/ 8, c->rccl_comm, c->stream);
    if (rc != 0) {
      set_error("ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?");
      return PO_ERR_COMM;
    }
    if (can_poll) {
      PO_TRY(launch_red_publish(c, c->d_gather, total * c->size));
      PO_TRY(wait_results(c));
    } else {
      PO_HIP(hipMemcpyAsync(c->h_red, c->d_gather, bytes * c->size, hipMemcpyDeviceToHost, c->stream));
      PO_HIP(hipStreamSynchronize(c->stream));
    }
    nparts = c->size;
  } else {
    // (red_direct: the final stages have written their results into h_red themselves, and the last of them has raised
    // the flag behind its results: poll it -- bounded, then the stream is synchronised as before)
    if (!red_direct(c)) PO_HIP(hipMemcpyAsync(c->h_red, c->d_red, bytes, hipMemcpyDeviceToHost, c->stream));
    if (red_direct(c) && c->h_flag && c->red_seq > c->red_seq_seen) {
      PO_TRY(wait_results(c));
    } else {
      PO_HIP(hipStreamSynchronize(c->stream));
    }
    if (c->comm_kind == COMM_CALLBACK) {
      double *all = c->h_red + kMaxRed;
      int rc = c->cb_allgather(c->h_red, all, total, c->cb_user);
      if (rc != 0) {
        set_error("allgather callback failed with code %d", rc);
        return PO_ERR_COMM;
      }
      parts = all;
      nparts = c->size;
    }
  }
  *parts_out = parts;
  *nparts_out = nparts;
  if (c->host_trace) {
    c->host_t_sync = host_now();
    c->host_wait_s += c->host_t_sync - ht0;
    c->host_gap_open = true;
  }
  return PO_OK;
}

static void combine_segment(const double *parts, int nparts, int stride, int off, int nsum, int nmin, int nmax,
                            double *host_out);

// Known-answer collectives over the solver's own exchange path (the code every reduction of the interior point goes
// through), run once when an RCCL communicator is created -- also for the forced single-rank communicator of
// PAROPT_AMD_FORCE_RCCL=1.  Rank r contributes (r + 1) everywhere.
//   pure-sum payload  -> ncclAllReduce:                      N (N + 1) / 2                  [MPI_Allreduce(SUM) sites]
//   mixed payload     -> rank-ordered ncclAllGather + host:  sum N (N + 1) / 2, min 1, max N, and slot r of the
//                        gathered buffer must hold r + 1 (rank order = the order the host combine assumes)
static int comm_self_test(Ctx *c) {
  const int N = c->size;
  const double mine = (double)(c->rank + 1), want_sum = 0.5 * N * (N + 1.0);
  const double *parts = nullptr;
  int nparts = 1;
  const long red0 = c->n_reductions, ar0 = c->n_allreduce, ag0 = c->n_allgather;
  auto put = [&](int count) -> int {
    for (int i = 0; i < count; i++) c->h_red[i] = mine;
    PO_HIP(hipMemcpyAsync(c->d_red, c->h_red, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
    PO_HIP(hipStreamSynchronize(c->stream));
    return PO_OK;
  };
  int rc = PO_OK;
  // 1. the all-reduce form (two slots: a payload, not a scalar)
  if (c->rccl_allreduce) {
    PO_TRY(put(2));
    PO_TRY(exchange_reduced(c, 2, true, &parts, &nparts));
    double got[2] = {0.0, 0.0};
    combine_segment(parts, nparts, 2, 0, 2, 0, 0, got);
    if (got[0] != want_sum || got[1] != want_sum) {
      set_error("RCCL self-test: all-reduce(SUM) of rank+1 over %d ranks gave %.17g / %.17g on rank %d, expected %.17g "
                "(librccl version code %d; wrong datatype/op enum or a communicator that does not span %d ranks)",
                N, got[0], got[1], c->rank, want_sum, g_rccl.version, N);
      rc = PO_ERR_COMM;
    }
  }
  // 2. the rank-ordered all-gather form with a SUM, a MIN and a MAX segment
  if (rc == PO_OK) {
    PO_TRY(put(3));
    PO_TRY(exchange_reduced(c, 3, false, &parts, &nparts));
    if (nparts != N) {
      set_error("RCCL self-test: the all-gather returned %d contributions for %d ranks", nparts, N);
      rc = PO_ERR_COMM;
    }
    for (int r = 0; rc == PO_OK && r < N; r++) {
      for (int s2 = 0; s2 < 3; s2++) {
        if (parts[(size_t)r * 3 + s2] != (double)(r + 1)) {
          set_error("RCCL self-test: slot %d of the gathered buffer holds %.17g on rank %d, expected %d (rank order)",
                    r, parts[(size_t)r * 3 + s2], c->rank, r + 1);
          rc = PO_ERR_COMM;
          break;
        }
      }
    }
    if (rc == PO_OK) {
      double got[3] = {0.0, 0.0, 0.0};
      combine_segment(parts, nparts, 3, 0, 1, 1, 1, got);
      if (got[0] != want_sum || got[1] != 1.0 || got[2] != (double)N) {
        set_error("RCCL self-test: SUM/MIN/MAX over %d ranks gave %.17g / %.17g / %.17g, expected %.17g / 1 / %d", N,
                  got[0], got[1], got[2], want_sum, N);
        rc = PO_ERR_COMM;
      }
    }
  }
  // the self-test does not show up in the statistics the bench reports
  c->n_reductions = red0;
  c->n_allreduce = ar0;
  c->n_allgather = ag0;
  return rc;
}

// Latency of the solver's collective path as the interior point uses it (bench.py's `collective_us`): `reps`
// exchanges of `count` doubles, each = final-stage payload in d_red -> collective -> device-to-host copy -> host
// sync; out = {median, min, max} host wall microseconds per exchange.
int comm_bench(Ctx *c, int count, int pure_sum, int reps, double out_us[3]) {
  if (count < 1 || count > kMaxRed || reps < 1) {
    set_error("comm_bench: bad arguments");
    return PO_ERR_ARG;
  }
  std::vector<double> t((size_t)reps, 0.0);
  PO_TRY(batch_flush(c));
  PO_TRY(ensure_partials(c, (size_t)count));
  PO_HIP(hipMemsetAsync(c->d_partials, 0, sizeof(double) * count, c->stream));
  PO_HIP(hipMemsetAsync(c->d_red, 0, sizeof(double) * count, c->stream));
  PO_HIP(hipStreamSynchronize(c->stream));
  const long red0 = c->n_reductions, ar0 = c->n_allreduce, ag0 = c->n_allgather, la0 = c->n_launches;
  for (int i = 0; i < reps + 2; i++) {  // two untimed warm-up exchanges
    const double *parts = nullptr;
    int nparts = 1;
    struct timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    // a real final stage (one block of partials per slot) in front of the exchange, as every reduction has it: the
    // number is "final-stage launch -> collective -> (copy) -> host sync", not a bare stream sync
    PO_TRY(launch_reduce_final(c, 1, count, count, 0, 0));
    PO_TRY(exchange_reduced(c, count, pure_sum != 0, &parts, &nparts));
    clock_gettime(CLOCK_MONOTONIC, &b);
    if (i >= 2) t[i - 2] = 1e6 * (double)(b.tv_sec - a.tv_sec) + 1e-3 * (double)(b.tv_nsec - a.tv_nsec);
  }
  c->n_reductions = red0;
  c->n_allreduce = ar0;
  c->n_allgather = ag0;
  c->n_launches = la0;
  std::sort(t.begin(), t.end());
  out_us[0] = t[t.size() / 2];
  out_us[1] = t.front();
  out_us[2] = t.back();
  return PO_OK;
}

// MPI_Allreduce on host values for user code (a problem's evalObjCon summing its rank-local objective parts, as
// examples/rosenbrock/rosenbrock.cpp:103-106 does with MPI_Allreduce): op 0 SUM, 1 MIN, 2 MAX; in place; collective.
int comm_allreduce_host(Ctx *c, double *values, int count, int op) {
  if (count < 0 || count > kMaxRed || op < 0 || op > 2) {
    set_error("po_ctx_allreduce: count %d (max %d) / op %d out of range", count, kMaxRed, op);
    return PO_ERR_ARG;
  }
  if (count == 0 || c->comm_kind == COMM_SELF) return PO_OK;
  PO_TRY(batch_flush(c));  // d_red is shared with queued reductions
  memcpy(c->h_red, values, sizeof(double) * count);
  PO_HIP(hipMemcpyAsync(c->d_red, c->h_red, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  const double *parts = nullptr;
  int nparts = 1;
  PO_TRY(exchange_reduced(c, count, op == 0, &parts, &nparts));
  std::vector<double> out((size_t)count);
  combine_segment(parts, nparts, count, 0, op == 0 ? count : 0, op == 1 ? count : 0, op == 2 ? count : 0, out.data());
  memcpy(values, out.data(), sizeof(double) * count);
  return PO_OK;
}

// combine the rank contributions of one segment in rank order: bit-identical on every rank
static void combine_segment(const double *parts, int nparts, int stride, int off, int nsum, int nmin, int nmax,
                            double *host_out) {
  const int nslots = nsum + nmin + nmax;
  for (int s = 0; s < nslots; s++) {
    double acc = parts[off + s];
    for (int r = 1; r < nparts; r++) {
      const double v = parts[(size_t)r * stride + off + s];
      if (s < nsum) {
        acc += v;
      } else if (s < nsum + nmin) {
        acc = fmin(acc, v);
      } else {
        acc = fmax(acc, v);
      }
    }
    host_out[s] = acc;
  }
}

void batch_abort(Ctx *c) {
  if (!c->partials_overflow.empty()) {
    (void)hipStreamSynchronize(c->stream);
    free_overflow(c, nullptr);
  }
  c->batch_pend.clear();
  c->batch_after.clear();
  c->batch_cursor = 0;
  c->partials_cursor = 0;
  c->mdot_timing_pending = false;
}

int batch_flush(Ctx *c, const double *keep_partials, size_t keep_len) {
  // where the arena cursor restarts: behind a region the caller still needs (its partials wait for the NEXT final
  // stage) -- set BEFORE the deferred host work runs, which may issue reductions of its own
  size_t cursor0 = 0;
  if (keep_partials && keep_partials >= c->partials_base && keep_partials < c->partials_base + c->partials_cap)
    cursor0 = (size_t)(keep_partials - c->partials_base) + keep_len;
  if (c->batch_pend.empty()) {
    // nothing queued: deferred host work (if any slipped in) still runs
    std::vector<std::function<void()>> after;
    after.swap(c->batch_after);
    for (auto &f : after) f();
    return PO_OK;
  }
  std::vector<Ctx::PendingRed> pend;
  pend.swap(c->batch_pend);
  std::vector<std::function<void()>> after;
  after.swap(c->batch_after);
  const int total = c->batch_cursor;
  c->batch_cursor = 0;
  c->partials_cursor = cursor0;
  // the final stages of everything queued: ONE launch (same per-slot arithmetic as the single-reduction kernel)
  PO_TRY(launch_reduce_final_multi(c, pend.data(), (int)pend.size()));
  bool pure_sum = true;
  for (const Ctx::PendingRed &p : pend) pure_sum = pure_sum && p.nmin == 0 && p.nmax == 0;
  const double *parts = nullptr;
  int nparts = 1;
  PO_TRY(exchange_reduced(c, total, pure_sum, &parts, &nparts));
  c->n_batched += (long)pend.size() - 1;
  for (const Ctx::PendingRed &p : pend) combine_segment(parts, nparts, total, p.off, p.nsum, p.nmin, p.nmax, p.host_out);
  // (the host has the results: every first-stage kernel that wrote into an overflow chunk has finished)
  if (!c->partials_overflow.empty()) free_overflow(c, keep_partials);
  for (auto &f : after) f();
  return PO_OK;
}

int reduce_finish(Ctx *c, int nblocks, int nsum, int nmin, int nmax, double *host_out, bool now) {
  const int nslots = nsum + nmin + nmax;
  if (nslots > kMaxRed) {
    set_error("reduction of %d slots exceeds kMaxRed=%d", nslots, kMaxRed);
    return PO_ERR_ARG;
  }
  const bool queued = c->batch_depth > 0 || !c->batch_pend.empty();
  // (the slot counts are the same on every rank: this flush is issued by all ranks or by none)
  const double *part = c->d_partials;
  if (queued && c->batch_cursor + nslots > kMaxRed) {
    // The partials of THIS reduction are still waiting in their region.  The flush resets the arena cursor and runs
    // the deferred host work, which may itself issue reductions (ensure_partials moves d_partials / partials_last):
    // region and length are taken before and handed to the flush, which restarts the cursor BEHIND the region before
    // that host work runs; the check below only restates it.
    const size_t len = c->partials_last;
    PO_TRY(batch_flush(c, part, len));
    if (part >= c->partials_base && part < c->partials_base + c->partials_cap) {
      const size_t end = (size_t)(part - c->partials_base) + len;
      if (c->partials_cursor < end) c->partials_cursor = end;
    }
  }
  if (c->batch_depth > 0 || !c->batch_pend.empty()) {
    c->batch_pend.push_back(Ctx::PendingRed{c->batch_cursor, nsum, nmin, nmax, host_out, part, nblocks});
    c->batch_cursor += nslots;
    if (now || c->batch_depth == 0) return batch_flush(c);
    return PO_OK;
  }
  PO_TRY(launch_reduce_final(c, nblocks, nslots, nsum, nmin, 0));
  const double *parts = nullptr;
  int nparts = 1;
  PO_TRY(exchange_reduced(c, nslots, nmin == 0 && nmax == 0, &parts, &nparts));
  combine_segment(parts, nparts, nslots, 0, nsum, nmin, nmax, host_out);
  return PO_OK;
}

}  // namespace po
