// paropt_amd -- hand-written gfx950 (CDNA4, wave64) kernels of the interior-point hot path.
//
// Design rules (DESIGN.md "Kernels"):
//   * every n-sized kernel is HBM-bound fp64 streaming: 16 B per lane per load (double2),
//     256-thread workgroups (4 wavefronts of 64), persistent grid-stride grids capped at
//     8 workgroups per CU so the 256 CUs / 8 XCDs are filled many times over;
//   * reductions are two-stage and deterministic: per-wave butterfly shuffles -> LDS across
//     the 4 waves -> one partial per workgroup per slot ([slot][block] layout) -> a second
//     tiny kernel that sums each slot over blocks in a fixed order;
//   * panel kernels keep the per-vector accumulators in VGPRs (compile-time panel blocks) and
//     read the multiplying vector once per block of up to 32 panel columns;
//   * the weighted Gram matrix is the only MFMA user (v_mfma_f64_16x16x4_f64, LDS-staged).
#include "core.hpp"

#include <math.h>
#include <stdlib.h>

namespace po {

// ---------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------
#define PO_PAIR_LOOP(q, n)                                                                 \
  const int64_t _npairs = ((n) + 1) >> 1;                                                  \
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < _npairs;            \
       q += (int64_t)gridDim.x * blockDim.x)

// Every device vector is allocated with an even number of (zero-initialised) elements plus slack
// (qn.cpp::vec_new), so the last pair of an odd-length vector can be loaded and stored as a full
// 16-byte access without a branch; the pad element is kept at exactly 0.0 by st2.
// Every n-sized operand is read once per kernel, gigabytes apart: non-temporal loads (element-wise kernels +4-13 %
// in tools/ab_libs.sh; -DPO_LD2_PLAIN builds the plain form for an A/B).
typedef double f64x2l __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double2 ld2(const double *__restrict__ p, int64_t q, int64_t n) {
#ifndef PO_LD2_PLAIN
  const f64x2l v = __builtin_nontemporal_load(reinterpret_cast<const f64x2l *>(p + 2 * q));
  return make_double2(v.x, v.y);
#else
  return *reinterpret_cast<const double2 *>(p + 2 * q);
#endif
}
#ifndef PO_ST2_PLAIN
#define PO_ST2_NT 1
#endif
typedef double f64x2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void st2(double *__restrict__ p, int64_t q, int64_t n, double2 v) {
  if (2 * q + 1 >= n) v.y = 0.0;
#ifdef PO_ST2_NT
  // every n-sized output is consumed by a later kernel, gigabytes of other traffic away: streaming store
  __builtin_nontemporal_store((f64x2s){v.x, v.y}, reinterpret_cast<f64x2s *>(p + 2 * q));
#else
  *reinterpret_cast<double2 *>(p + 2 * q) = v;
#endif
}

// native vector types: register arrays of HIP's double2 class get demoted to scratch across barriers
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Panel columns are read exactly once per pass: non-temporal loads (+2-3 % in tools/tune_mdot.hip).
__device__ __forceinline__ f64x2 ld_stream(const double *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
}

enum { OP_SUM = 0, OP_MIN = 1, OP_MAX = 2 };

template <int OP>
__device__ __forceinline__ double combine(double a, double b) {
  if (OP == OP_SUM) return a + b;
  if (OP == OP_MIN) return fmin(a, b);
  return fmax(a, b);
}

template <int OP>
__device__ __forceinline__ double wave_reduce(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = combine<OP>(v, __shfl_xor(v, o, 64));
  return v;
}

// Reduce N per-thread values over the 256-thread workgroup and store one partial per slot at
// partials[(slot0 + j) * gridDim.x + blockIdx.x].  `sm` must hold 4*N doubles.
template <int N, int OP>
__device__ __forceinline__ void block_reduce_store(double (&a)[N], double *__restrict__ partials,
                                                   int slot0, double *sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double v = wave_reduce<OP>(a[j]);
    if (lane == 0) sm[wave * N + j] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int j = threadIdx.x;
    double v = combine<OP>(combine<OP>(sm[j], sm[N + j]), combine<OP>(sm[2 * N + j], sm[3 * N + j]));
    partials[(size_t)(slot0 + j) * gridDim.x + blockIdx.x] = v;
  }
  __syncthreads();
}

// Completion flag of a final reduction stage whose results go straight into pinned host memory (Ctx::h_flag): every
// wave fences its result store to system scope, the workgroup meets at a barrier, and the workgroup that draws the
// last ticket writes the sequence number the host is polling for.  flag == nullptr: nothing (results go to d_red).
// Called from exactly ONE program point of each kernel, by every thread of the workgroup (the barrier inside is a
// convergent operation: waves without a slot fall through to the same call instead of returning early).
__device__ __forceinline__ void red_raise_flag(volatile unsigned long long *flag, unsigned long long seq,
                                               unsigned *ticket) {
  if (!flag) return;
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(ticket, 1u);
    if (t == gridDim.x - 1) {
      *ticket = 0;
      __threadfence_system();
      *flag = seq;
    }
  }
}
// one slot of a final stage: sum / min / max of its first-stage partials over the blocks, lane-strided + butterfly
__device__ __forceinline__ void red_final_slot(const double *__restrict__ p, int nblocks, int slot, int nsum, int nmin,
                                               double *__restrict__ o, int lane) {
  if (slot < nsum) {
    double acc = 0.0;
    for (int b = lane; b < nblocks; b += 64) acc += p[b];
    acc = wave_reduce<OP_SUM>(acc);
    if (lane == 0) o[slot] = acc;
  } else if (slot < nsum + nmin) {
    double acc = INFINITY;
    for (int b = lane; b < nblocks; b += 64) acc = fmin(acc, p[b]);
    acc = wave_reduce<OP_MIN>(acc);
    if (lane == 0) o[slot] = acc;
  } else {
    double acc = -INFINITY;
    for (int b = lane; b < nblocks; b += 64) acc = fmax(acc, p[b]);
    acc = wave_reduce<OP_MAX>(acc);
    if (lane == 0) o[slot] = acc;
  }
}
__global__ void __launch_bounds__(kBlock)
    reduce_final_kernel(const double *__restrict__ partials, int nblocks, int nslots, int nsum,
                        int nmin, double *__restrict__ out, volatile unsigned long long *flag, unsigned long long seq,
                        unsigned *ticket) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int slot = blockIdx.x * 4 + wave;
  const bool active = slot < nslots;  // wave-uniform
  if (active) red_final_slot(partials + (size_t)slot * nblocks, nblocks, slot, nsum, nmin, out, lane);
  red_raise_flag(flag, seq, ticket);
}

// After a device-side collective (RCCL): copies its `count` results from device memory into the pinned host buffer
// and raises the completion flag behind them, so that the host polls instead of queueing a device-to-host copy and
// synchronising the stream -- the completion path single-rank runs already had (one workgroup).
__global__ void __launch_bounds__(kBlock) red_publish_kernel(const double *__restrict__ src, int count,
                                                             double *__restrict__ dst,
                                                             volatile unsigned long long *flag,
                                                             unsigned long long seq) {
  for (int i = threadIdx.x; i < count; i += kBlock) dst[i] = src[i];
  __threadfence_system();
  __syncthreads();
  if (threadIdx.x == 0) *flag = seq;
}
int launch_red_publish(Ctx *c, const double *src, int count) {
  c->red_seq++;
  hipLaunchKernelGGL(red_publish_kernel, dim3(1), dim3(kBlock), 0, c->stream, src, count, c->h_red_dev, c->h_flag_dev,
                     c->red_seq);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return PO_OK;
}

int launch_reduce_final(Ctx *c, int nblocks, int nslots, int nsum, int nmin, int dst_off) {
  const int grid = (nslots + 3) / 4;
  // no device-side collective (one rank, or the host-callback communicator): straight into the pinned host buffer
  double *dst = red_direct(c) ? c->h_red_dev : c->d_red;
  const bool flagged = red_direct(c) && c->h_flag_dev != nullptr;
  if (flagged) c->red_seq++;
  hipLaunchKernelGGL(reduce_final_kernel, dim3(grid), dim3(kBlock), 0, c->stream, c->d_partials,
                     nblocks, nslots, nsum, nmin, dst + dst_off, flagged ? c->h_flag_dev : nullptr, c->red_seq,
                     c->d_ticket);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return PO_OK;
}

// The final stages of several queued reductions in ONE launch (BatchScope, round 4): wave `g` of the grid owns global
// slot g, finds the reduction it belongs to in the by-value table and sums that slot over the reduction's own blocks
// exactly as reduce_final_kernel does (same lane stride, same butterfly: same bits).
constexpr int kMaxSeg = 16;
struct RedSegTable {
  const double *part[kMaxSeg];
  int nblocks[kMaxSeg], nslots[kMaxSeg], nsum[kMaxSeg], nmin[kMaxSeg], dst[kMaxSeg];
  int count, total;
};
__global__ void __launch_bounds__(kBlock) reduce_final_multi_kernel(RedSegTable T, double *__restrict__ out,
                                                                    volatile unsigned long long *flag,
                                                                    unsigned long long seq, unsigned *ticket) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int slot = blockIdx.x * 4 + wave;
  const bool active = slot < T.total;  // wave-uniform
  if (active) {
    int k = 0;
    while (k + 1 < T.count && slot >= T.nslots[k]) {
      slot -= T.nslots[k];
      k++;
    }
    red_final_slot(T.part[k] + (size_t)slot * T.nblocks[k], T.nblocks[k], slot, T.nsum[k], T.nmin[k], out + T.dst[k],
                   lane);
  }
  red_raise_flag(flag, seq, ticket);
}
int launch_reduce_final_multi(Ctx *c, const Ctx::PendingRed *pend, int count) {
  double *dst = red_direct(c) ? c->h_red_dev : c->d_red;
  for (int i0 = 0; i0 < count; i0 += kMaxSeg) {
    RedSegTable T;
    T.count = count - i0 < kMaxSeg ? count - i0 : kMaxSeg;
    T.total = 0;
    for (int k = 0; k < kMaxSeg; k++) {
      const bool live = k < T.count;
      const Ctx::PendingRed &p = pend[i0 + (live ? k : 0)];
      T.part[k] = live ? p.part : nullptr;
      T.nblocks[k] = live ? p.nblocks : 0;
      T.nslots[k] = live ? p.nsum + p.nmin + p.nmax : 0;
      T.nsum[k] = live ? p.nsum : 0;
      T.nmin[k] = live ? p.nmin : 0;
      T.dst[k] = live ? p.off : 0;
      T.total += T.nslots[k];
    }
    if (T.total <= 0) continue;
    const bool flagged = red_direct(c) && c->h_flag_dev != nullptr;
    if (flagged) c->red_seq++;
    hipLaunchKernelGGL(reduce_final_multi_kernel, dim3((T.total + 3) / 4), dim3(kBlock), 0, c->stream, T, dst,
                       flagged ? c->h_flag_dev : nullptr, c->red_seq, c->d_ticket);
    c->n_launches++;
    PO_HIP(hipGetLastError());
  }
  return PO_OK;
}

// Persistent grid for an n-element streaming kernel: `bpc` workgroups per CU (tools/tune_mdot.hip:
// 3-6 resident workgroups per CU stream at 6.2-6.5 TB/s; 8 per CU with ~6 resident leaves a
// 1.33-wave tail and drops to 5.5 TB/s), fewer when n is small.  The two classes of the iteration were
// swept in-process at config 3 (profiles/r03_ab_bpc.txt): the panel-streaming kernels (c+k+~10 concurrent
// streams per thread) are fastest at 2 per CU, the few-stream kernels at 4.
int grid_for(Ctx *c, int64_t n, int bpc) {
  if (bpc == kBpcPanel) bpc = dbg_switch(SW_BPC3, "PAROPT_AMD_BPC_PANEL", bpc);
  else if (bpc == kBpcStream) bpc = dbg_switch(SW_BPC4, "PAROPT_AMD_BPC_STREAM", bpc);
  const int64_t npairs = (n + 1) >> 1;
  int64_t blocks = (npairs + kBlock - 1) / kBlock;
  const int64_t cap = (int64_t)c->num_cu * bpc;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  return (int)blocks;
}
int grid_for(Ctx *c, int64_t n) { return grid_for(c, n, kBpcStream); }

#define PO_LAUNCH(kernel, grid, ...)                                                     \
  do {                                                                                    \
    const double _ht0 = host_trace_begin(c);                                              \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, c->stream, __VA_ARGS__);      \
    c->n_launches++;                                                                      \
    host_trace_end(c, _ht0);                                                              \
    PO_HIP(hipGetLastError());                                                            \
  } while (0)

// ---------------------------------------------------------------------------------------------
// counter hash (DESIGN.md "Synthetic data"): identical to oracle/paropt_oracle.py::u01
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ double u01(uint64_t base, uint64_t i) {
  return (double)(splitmix64(base + i) >> 11) * (1.0 / 9007199254740992.0);
}

// ---------------------------------------------------------------------------------------------
// BLAS-1 style kernels (ParOptBasicVec, src/ParOptVec.cpp:32-204)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock) fill_kernel(double *__restrict__ y, int64_t n, double a) {
  PO_PAIR_LOOP(q, n) { st2(y, q, n, make_double2(a, a)); }
}
__global__ void __launch_bounds__(kBlock)
    fill_hash_kernel(double *__restrict__ y, int64_t n, uint64_t base, int64_t offset, double scale,
                     double shift) {
  PO_PAIR_LOOP(q, n) {
    const uint64_t i = (uint64_t)(offset + 2 * q);
    st2(y, q, n, make_double2(shift + scale * u01(base, i), shift + scale * u01(base, i + 1)));
  }
}
__global__ void __launch_bounds__(kBlock)
    copy_kernel(double *__restrict__ y, const double *__restrict__ x, int64_t n) {
  PO_PAIR_LOOP(q, n) { st2(y, q, n, ld2(x, q, n)); }
}
__global__ void __launch_bounds__(kBlock) scale_kernel(double *__restrict__ y, int64_t n, double a) {
  PO_PAIR_LOOP(q, n) {
    double2 v = ld2(y, q, n);
    st2(y, q, n, make_double2(a * v.x, a * v.y));
  }
}
__global__ void __launch_bounds__(kBlock)
    axpy_kernel(double *__restrict__ y, double a, const double *__restrict__ x, int64_t n) {
  PO_PAIR_LOOP(q, n) {
    double2 v = ld2(y, q, n), w = ld2(x, q, n);
    st2(y, q, n, make_double2(v.x + a * w.x, v.y + a * w.y));
  }
}

// y_i = +1 where x_i >= 0, else -1 (the step direction of checkGradients, src/ParOptProblem.cpp:262-270)
__global__ void __launch_bounds__(kBlock)
    sign_kernel(double *__restrict__ y, const double *__restrict__ x, int64_t n) {
  PO_PAIR_LOOP(q, n) {
    const double2 v = ld2(x, q, n);
    st2(y, q, n, make_double2(v.x >= 0.0 ? 1.0 : -1.0, v.y >= 0.0 ? 1.0 : -1.0));
  }
}
int k_sign(Ctx *c, double *y, const double *x, int64_t n) {
  count_bytes(c, 2, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(sign_kernel, grid_for(c, n), y, x, n);
  return PO_OK;
}
int k_fill(Ctx *c, double *y, int64_t n, double a) {
  count_bytes(c, 1, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(fill_kernel, grid_for(c, n), y, n, a);
  return PO_OK;
}
int k_fill_hash(Ctx *c, double *y, int64_t n, uint64_t seed, uint64_t aid, int64_t offset,
                double scale, double shift) {
  count_bytes(c, 1, n);
  if (n <= 0) return PO_OK;
  const uint64_t base = seed * 0x9E3779B97F4A7C15ULL + aid * 0xD1B54A32D192ED03ULL;
  PO_LAUNCH(fill_hash_kernel, grid_for(c, n), y, n, base, offset, scale, shift);
  return PO_OK;
}
int k_copy(Ctx *c, double *y, const double *x, int64_t n) {
  count_bytes(c, 2, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(copy_kernel, grid_for(c, n), y, x, n);
  return PO_OK;
}
int k_scale(Ctx *c, double *y, int64_t n, double a) {
  count_bytes(c, 2, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(scale_kernel, grid_for(c, n), y, n, a);
  return PO_OK;
}
int k_axpy(Ctx *c, double *y, double a, const double *x, int64_t n) {
  count_bytes(c, 3, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(axpy_kernel, grid_for(c, n), y, a, x, n);
  return PO_OK;
}


// sum_j coef_j P_j at pair q, streamed in register batches of 8 / 4 / 2 / 1 columns: the loads of a
// batch are issued back to back before its FMAs (see mdot_kernel for why).
template <int B>
__device__ __forceinline__ void panel_batch(const PtrTable &P, const CoefTable &a, int j, int64_t q,
                                            f64x2 &acc) {
  f64x2 v[B];
#pragma unroll
  for (int u = 0; u < B; u++) v[u] = ld_stream(P.p[j + u] + 2 * q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < B; u++) {
    acc.x += a.a[j + u] * v[u].x;
    acc.y += a.a[j + u] * v[u].y;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ double2 panel_sum(const PtrTable &P, const CoefTable &a, int nv, int64_t q,
                                             int j = 0) {
  f64x2 acc = (f64x2){0.0, 0.0};
  for (; j + 8 <= nv; j += 8) panel_batch<8>(P, a, j, q, acc);
  if (j + 4 <= nv) {
    panel_batch<4>(P, a, j, q, acc);
    j += 4;
  }
  if (j + 2 <= nv) {
    panel_batch<2>(P, a, j, q, acc);
    j += 2;
  }
  if (j < nv) panel_batch<1>(P, a, j, q, acc);
  return make_double2(acc.x, acc.y);
}

// The pair (2q, 2q+1) of a grouped column (core.hpp: GroupCol), as k_group_scatter_set would have stored it, in two
// steps: gcol_request issues the two loads (clamped indices, no arithmetic on the loaded values: nothing waits for
// them), gcol_value turns them into the column's entries where they are consumed.  (With the arithmetic inside the
// request, a prefetch of the next tile waited for ALL its loads on the spot: solve2_dots lost 16 %.)
struct GRaw {
  f64x2 raw;
  int valid;  // bit 0 / 1: element 2q / 2q + 1 belongs to a group
};
__device__ __forceinline__ GRaw gcol_request(const GroupCol &gc, int64_t q, int64_t n) {
  GRaw r;
  const unsigned r0 = (unsigned)(2 * q);
  const unsigned g0 = r0 / gc.period, k0 = r0 - g0 * gc.period;
  unsigned g1 = g0, k1 = k0 + 1;
  if (k1 == gc.period) {
    g1 = g0 + 1;
    k1 = 0;
  }
  const bool v0 = k0 < gc.nw && (long long)g0 < gc.nwcon;
  const bool v1 = k1 < gc.nw && (long long)g1 < gc.nwcon && 2 * q + 1 < n;
  // (an empty map loads nothing: nwcon - 1 would wrap, and w may be a null vector then; wave-uniform test)
  const unsigned last = gc.nwcon > 0 ? (unsigned)(gc.nwcon - 1) : 0u;
  r.raw.x = gc.nwcon > 0 ? gc.w[g0 < last ? g0 : last] : 0.0;
  r.raw.y = gc.nwcon > 0 ? gc.w[g1 < last ? g1 : last] : 0.0;
  r.valid = (v0 ? 1 : 0) | (v1 ? 2 : 0);
  return r;
}
__device__ __forceinline__ f64x2 gcol_value(const GroupCol &gc, const GRaw &r) {
  f64x2 v;
  v.x = (r.valid & 1) ? __dadd_rn(0.0, __dmul_rn(gc.scale, r.raw.x)) : 0.0;
  v.y = (r.valid & 2) ? __dadd_rn(0.0, __dmul_rn(gc.scale, r.raw.y)) : 0.0;
  return v;
}

// two coefficient sets over the same panel in one pass (solve + refinement residual)
template <int B>
__device__ __forceinline__ void panel_batch2(const PtrTable &P, const CoefTable &a, const CoefTable &b2,
                                             int j, int64_t q, f64x2 &acc, f64x2 &acc2) {
  f64x2 v[B];
#pragma unroll
  for (int u = 0; u < B; u++) v[u] = ld_stream(P.p[j + u] + 2 * q);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < B; u++) {
    acc.x += a.a[j + u] * v[u].x;
    acc.y += a.a[j + u] * v[u].y;
    acc2.x += b2.a[j + u] * v[u].x;
    acc2.y += b2.a[j + u] * v[u].y;
  }
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void panel_sum2(const PtrTable &P, const CoefTable &a, const CoefTable &b2,
                                           int nv, int64_t q, double2 &s1, double2 &s2, int j = 0) {
  f64x2 acc = (f64x2){0.0, 0.0}, acc2 = (f64x2){0.0, 0.0};
  for (; j + 8 <= nv; j += 8) panel_batch2<8>(P, a, b2, j, q, acc, acc2);
  if (j + 4 <= nv) {
    panel_batch2<4>(P, a, b2, j, q, acc, acc2);
    j += 4;
  }
  if (j + 2 <= nv) {
    panel_batch2<2>(P, a, b2, j, q, acc, acc2);
    j += 2;
  }
  if (j < nv) panel_batch2<1>(P, a, b2, j, q, acc, acc2);
  s1 = make_double2(acc.x, acc.y);
  s2 = make_double2(acc2.x, acc2.y);
}

// three coefficient sets over the same panel in one pass (first solve, its refinement residual, refinement solve)
template <int B>
__device__ __forceinline__ void panel_batch3(const PtrTable &P, const CoefTable &a, const CoefTable &b2,
                                             const CoefTable &c3, int j, int64_t q, f64x2 &acc, f64x2 &acc2,
                                             f64x2 &acc3, const VirtCols &vc) {
  f64x2 v[B];
#pragma unroll
  for (int u = 0; u < B; u++) v[u] = ld_stream(P.p[j + u] + 2 * q);
  if (j < vc.count) {  // unformed L-SR1 columns: Z = Y - b0 S in registers
    f64x2 w[B];
#pragma unroll
    for (int u = 0; u < B; u++) w[u] = ld_stream(vc.s[(j + u) < vc.count ? (j + u) : 0] + 2 * q);
#pragma unroll
    for (int u = 0; u < B; u++) {
      if (j + u < vc.count) {
        v[u].x -= vc.b0 * w[u].x;
        v[u].y -= vc.b0 * w[u].y;
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < B; u++) {
    acc.x += a.a[j + u] * v[u].x;
    acc.y += a.a[j + u] * v[u].y;
    acc2.x += b2.a[j + u] * v[u].x;
    acc2.y += b2.a[j + u] * v[u].y;
    acc3.x += c3.a[j + u] * v[u].x;
    acc3.y += c3.a[j + u] * v[u].y;
  }
  __builtin_amdgcn_sched_barrier(0);
}
// columns [j, nv) added onto (s1, s2, s3)
__device__ __forceinline__ void panel_sum3(const PtrTable &P, const CoefTable &a, const CoefTable &b2,
                                           const CoefTable &c3, int nv, int64_t q, double2 &s1, double2 &s2,
                                           double2 &s3, int j, const VirtCols &vc) {
  f64x2 acc = (f64x2){s1.x, s1.y}, acc2 = (f64x2){s2.x, s2.y}, acc3 = (f64x2){s3.x, s3.y};
  for (; j + 8 <= nv; j += 8) panel_batch3<8>(P, a, b2, c3, j, q, acc, acc2, acc3, vc);
  if (j + 4 <= nv) {
    panel_batch3<4>(P, a, b2, c3, j, q, acc, acc2, acc3, vc);
    j += 4;
  }
  if (j + 2 <= nv) {
    panel_batch3<2>(P, a, b2, c3, j, q, acc, acc2, acc3, vc);
    j += 2;
  }
  if (j < nv) panel_batch3<1>(P, a, b2, c3, j, q, acc, acc2, acc3, vc);
  s1 = make_double2(acc.x, acc.y);
  s2 = make_double2(acc2.x, acc2.y);
  s3 = make_double2(acc3.x, acc3.y);
}

// y <- a*x + b*y + sum_j alpha_j V_j : runtime panel width, no per-column registers needed.
__global__ void __launch_bounds__(kBlock)
    panel_axpy_kernel(double *__restrict__ y, double a, const double *__restrict__ x, double b,
                      CoefTable alpha, PtrTable V, int nv, int64_t n, GroupCol ybase) {
  PO_PAIR_LOOP(q, n) {
    GRaw graw;
    if (ybase.w && b != 0.0) graw = gcol_request(ybase, q, n);
    double2 acc = make_double2(0.0, 0.0);
    if (a != 0.0) {
      double2 v = ld2(x, q, n);
      acc.x = a * v.x;
      acc.y = a * v.y;
    }
    if (b != 0.0) {
      double2 v;
      if (ybase.w) {
        const f64x2 gbv = gcol_value(ybase, graw);
        v = make_double2(gbv.x, gbv.y);
      } else {
        v = ld2(y, q, n);
      }
      acc.x += b * v.x;
      acc.y += b * v.y;
    }
    const double2 ps = panel_sum(V, alpha, nv, q);
    acc.x += ps.x;
    acc.y += ps.y;
    st2(y, q, n, acc);
  }
}

// two linear combinations over ONE pass of the same panel: y1 = a1 x1 + sum_j c1_j V_j, y2 = a2 x2 + sum_j c2_j V_j
// (the objective gradient gk + B s and the modelled constraint's gradient g0 + H M H^T s of the compact eigenvalue
// subproblem share the columns H: src/ParOptCompactEigenvalueApprox.cpp:626-643).  Same accumulation order per output
// as panel_axpy_kernel; a zero coefficient contributes an exact zero.
__global__ void __launch_bounds__(kBlock)
    panel_axpy2_kernel(double *__restrict__ y1, double a1, const double *__restrict__ x1, double *__restrict__ y2,
                       double a2, const double *__restrict__ x2, CoefTable c1, CoefTable c2, PtrTable V, int nv,
                       int64_t n) {
  PO_PAIR_LOOP(q, n) {
    const double2 v1 = ld2(x1, q, n), v2 = ld2(x2, q, n);
    double2 s1, s2;
    panel_sum2(V, c1, c2, nv, q, s1, s2);
    double2 r1 = make_double2(a1 * v1.x, a1 * v1.y), r2 = make_double2(a2 * v2.x, a2 * v2.y);
    r1.x += s1.x;
    r1.y += s1.y;
    r2.x += s2.x;
    r2.y += s2.y;
    st2(y1, q, n, r1);
    st2(y2, q, n, r2);
  }
}

static void fill_tables(const double *alpha, const double *const *V, int nv, CoefTable *ct,
                        PtrTable *pt) {
  for (int j = 0; j < kMaxPanel; j++) {
    ct->a[j] = (alpha && j < nv) ? alpha[j] : 0.0;
    pt->p[j] = (V && j < nv) ? V[j] : nullptr;
  }
}

// A grouped column (core.hpp: GroupCol) is index arithmetic on 32-bit row numbers with a division by its period: every
// launcher that takes one checks the description here, whoever produced it.
static int gcol_check(const GroupCol *g, int64_t n, const char *who) {
  if (!g || !g->w) return PO_OK;
  if (g->period == 0 || g->nw > g->period || g->nwcon < 0 || n >= 2147483647LL ||
      (g->nwcon > 0 && (int64_t)(g->nwcon - 1) * g->period + g->nw > n)) {
    set_error("%s: bad grouped column (period %u, nw %u, nwcon %lld, n %lld)", who, g->period, g->nw,
              (long long)g->nwcon, (long long)n);
    return PO_ERR_ARG;
  }
  return PO_OK;
}

int k_panel_axpy(Ctx *c, double *y, double a, const double *x, double b, const double *alpha,
                 const double *const *V, int nv, int64_t n, const GroupCol *ybase) {
  if (n <= 0) return PO_OK;
  PO_TRY(gcol_check(ybase, n, "k_panel_axpy"));
  // panels wider than one kernel's argument tables go in slabs: the first carries a*x + b*y, the others accumulate
  int j0 = 0;
  do {
    const int w = nv - j0 > kMaxPanel ? kMaxPanel : nv - j0;
    const double aa = j0 == 0 ? a : 0.0, bb = j0 == 0 ? b : 1.0;
    const bool gb = ybase && ybase->w && j0 == 0 && bb != 0.0;  // the b*y term of the first slab from the grouped column
    count_bytes(c, w + 1 + (aa != 0.0 ? 1 : 0) + ((bb != 0.0 && !gb) ? 1 : 0), n);
    if (gb) count_bytes(c, 1.0, ybase->nwcon);
    CoefTable ct;
    PtrTable pt;
    fill_tables(alpha ? alpha + j0 : nullptr, V ? V + j0 : nullptr, w, &ct, &pt);
    PO_LAUNCH(panel_axpy_kernel, grid_for(c, n, kBpcPanel), y, aa, x, bb, ct, pt, w, n, gb ? *ybase : GroupCol());
    j0 += w;
  } while (j0 < nv);
  return PO_OK;
}

int k_panel_axpy2(Ctx *c, double *y1, double a1, const double *x1, const double *c1, double *y2, double a2,
                  const double *x2, const double *c2, const double *const *V, int nv, int64_t n) {
  if (nv > kMaxPanel) {
    set_error("panel of %d vectors exceeds kMaxPanel=%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  count_bytes(c, nv + 4, n);
  if (n <= 0) return PO_OK;
  PtrTable pt;
  CoefTable ct1, ct2;
  fill_tables(c1, V, nv, &ct1, &pt);
  fill_tables(c2, V, nv, &ct2, &pt);
  PO_LAUNCH(panel_axpy2_kernel, grid_for(c, n, kBpcPanel), y1, a1, x1, y2, a2, x2, ct1, ct2, pt, nv, n);
  return PO_OK;
}

// dst_j <- a*X_j + b*Y_j for a whole panel in ONE launch (constraint-Jacobian copies of the built-in
// problems, L-SR1's Z_j = Y_j - b0 S_j rebuild): same bytes as nv separate launches, nv-1 fewer
// launches per call, which matters once the per-GPU shard is small (multi-GPU strong scaling).
struct PtrTableW {
  double *p[kMaxPanel];
};
__global__ void __launch_bounds__(kBlock)
    panel_lincomb_kernel(PtrTableW dst, double a, PtrTable X, double b, PtrTable Y, int nv, int has_y,
                         int64_t n) {
  PO_PAIR_LOOP(q, n) {
    for (int j = 0; j < nv; j++) {
      const f64x2 xv = ld_stream(X.p[j] + 2 * q);
      double2 r = make_double2(a * xv.x, a * xv.y);
      if (has_y) {
        const f64x2 yv = ld_stream(Y.p[j] + 2 * q);
        r.x += b * yv.x;
        r.y += b * yv.y;
      }
      st2(dst.p[j], q, n, r);
    }
  }
}
// the same with the columns spread over blockIdx.y: every workgroup streams ONE column (a contiguous read and a
// contiguous write stream per workgroup, like the plain copy kernel) instead of touching all nv columns per row pair
__global__ void __launch_bounds__(kBlock)
    panel_lincomb2d_kernel(PtrTableW dst, double a, PtrTable X, double b, PtrTable Y, int has_y, int64_t n) {
  const int j = blockIdx.y;
  const double *__restrict__ xp = X.p[j];
  const double *__restrict__ yp = Y.p[j];
  double *__restrict__ dp = dst.p[j];
  PO_PAIR_LOOP(q, n) {
    const f64x2 xv = ld_stream(xp + 2 * q);
    double2 r = make_double2(a * xv.x, a * xv.y);
    if (has_y) {
      const f64x2 yv = ld_stream(yp + 2 * q);
      r.x += b * yv.x;
      r.y += b * yv.y;
    }
    st2(dp, q, n, r);
  }
}

int k_panel_lincomb(Ctx *c, double *const *dst, double a, const double *const *X, double b,
                    const double *const *Y, int nv, int64_t n) {
  if (n <= 0 || nv <= 0) return PO_OK;
  count_bytes(c, (double)nv * (Y ? 3 : 2), n);
  for (int j0 = 0; j0 < nv; j0 += kMaxPanel) {  // independent columns: slabs of one kernel's table width
    const int w = nv - j0 > kMaxPanel ? kMaxPanel : nv - j0;
    PtrTableW d;
    PtrTable x, y;
    CoefTable ct;
    for (int j = 0; j < kMaxPanel; j++) d.p[j] = j < w ? dst[j0 + j] : nullptr;
    fill_tables(nullptr, X + j0, w, &ct, &x);
    fill_tables(nullptr, Y ? Y + j0 : nullptr, Y ? w : 0, &ct, &y);
    // A/B (profiles/r03_ab_lincomb.txt): all columns per row pair (0, default), one column per blockIdx.y (1) and one
    // launch per column (2) all copy at 5.1-6.2 TB/s with the same run-to-run spread; the best samples of each sit on
    // the copy ceiling measured in the same process (6.15 TB/s)
    const int form2d = dbg_switch(SW_LINCOMB_2D, "PAROPT_AMD_LINCOMB_2D", 0);
    if (form2d == 2) {  // one launch per column (A/B: the plain 1-D copy shape)
      for (int j = 0; j < w; j++) {
        PtrTableW d1;
        PtrTable x1, y1;
        for (int q = 0; q < kMaxPanel; q++) {
          d1.p[q] = d.p[j];
          x1.p[q] = x.p[j];
          y1.p[q] = y.p[j];
        }
        hipLaunchKernelGGL(panel_lincomb2d_kernel, dim3(grid_for(c, n), 1), dim3(kBlock), 0, c->stream, d1, a, x1, b, y1,
                           Y ? 1 : 0, n);
        c->n_launches++;
      }
      PO_HIP(hipGetLastError());
    } else if (form2d && w >= 4) {
      int gx = (c->num_cu * dbg_switch(SW_LINCOMB_BPC, "PAROPT_AMD_LINCOMB_BPC", 8) + w - 1) / w;  // ~8 workgroups per CU in all
      const int need = grid_for(c, n);
      if (gx > need) gx = need;
      if (gx < 1) gx = 1;
      hipLaunchKernelGGL(panel_lincomb2d_kernel, dim3(gx, w), dim3(kBlock), 0, c->stream, d, a, x, b, y, Y ? 1 : 0, n);
      c->n_launches++;
      PO_HIP(hipGetLastError());
    } else {
      PO_LAUNCH(panel_lincomb_kernel, grid_for(c, n), d, a, x, b, y, w, Y ? 1 : 0, n);
    }
  }
  return PO_OK;
}


// ---------------------------------------------------------------------------------------------
// Panels wider than one kernel's argument tables (kMaxPanel columns).  The reference has no limit on the number of
// dense constraints or on the quasi-Newton width (src/ParOptInteriorPoint.cpp:1935-1950, 2648-2654), so the panel
// kernels below accept any width: sum_j coef_j P_j over a column range is formed by slabbed panel_axpy passes into
// one of the context's two scratch vectors and enters the kernel as ONE column with coefficient 1.  Costs a pass
// over the range plus one n-sized write per range: the wide path is correct, not fast.
// ---------------------------------------------------------------------------------------------
static int collapse_range(Ctx *c, int which, const double *coef, const double *const *P, int j0, int j1,
                          int64_t n, const double **out) {
  if (c->wide_n < n || !c->wide_scratch[which]) {
    PO_HIP(hipStreamSynchronize(c->stream));
    for (int i = 0; i < 2; i++) {
      if (c->wide_n < n && c->wide_scratch[i]) {
        PO_HIP(hipFree(c->wide_scratch[i]));
        c->wide_scratch[i] = nullptr;
      }
    }
    if (c->wide_n < n) c->wide_n = n;
    const size_t bytes = sizeof(double) * (size_t)(((c->wide_n + 1) / 2) * 2 + 2);
    PO_HIP(hipMalloc((void **)&c->wide_scratch[which], bytes));
    PO_HIP(hipMemsetAsync(c->wide_scratch[which], 0, bytes, c->stream));
  }
  PO_TRY(k_panel_axpy(c, c->wide_scratch[which], 0.0, nullptr, 0.0, coef + j0, P + j0, j1 - j0, n));
  *out = c->wide_scratch[which];
  return PO_OK;
}
// [A-part | rest] -> at most two columns: P2/coef2 receive the collapsed panel, *nv2 / *nca2 its widths
static int collapse_panel(Ctx *c, const double *coef, const double *const *P, int nv, int nca, int64_t n,
                          const double *P2[2], double coef2[2], int *nv2, int *nca2) {
  int m = 0;
  *nca2 = 0;
  if (nca > 0) {
    PO_TRY(collapse_range(c, 0, coef, P, 0, nca, n, &P2[m]));
    coef2[m++] = 1.0;
    *nca2 = 1;
  }
  if (nv > nca) {
    PO_TRY(collapse_range(c, 1, coef, P, nca, nv, n, &P2[m]));
    coef2[m++] = 1.0;
  }
  *nv2 = m;
  return PO_OK;
}

// stream ceilings for the bench line (po_bench_stream): read-only x.y and copy y <- x, no host sync
template <int KIND>
__global__ void __launch_bounds__(kBlock)
    reduce1_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                   double *__restrict__ partials);
int k_stream_launch(Ctx *c, int kind, double *x, double *y, int64_t n) {
  if (n <= 0) return PO_OK;
  if (kind == 0) {
    const int grid = grid_for(c, n, 5);
    PO_TRY(ensure_partials(c, (size_t)grid));
    PO_LAUNCH(reduce1_kernel<RED_DOT>, grid, x, y, n, c->d_partials);
  } else {
    PO_LAUNCH(copy_kernel, grid_for(c, n), y, x, n);
  }
  return PO_OK;
}

// dot / sum of squares / asum / amax
template <int KIND>
__global__ void __launch_bounds__(kBlock)
    reduce1_kernel(const double *__restrict__ x, const double *__restrict__ y, int64_t n,
                   double *__restrict__ partials) {
  __shared__ double sm[4];
  double acc[1] = {0.0};
  PO_PAIR_LOOP(q, n) {
    double2 v = ld2(x, q, n);
    if (KIND == RED_DOT) {
      double2 w = ld2(y, q, n);
      acc[0] += v.x * w.x + v.y * w.y;
    } else if (KIND == RED_SUMSQ) {
      acc[0] += v.x * v.x + v.y * v.y;
    } else if (KIND == RED_ASUM) {
      acc[0] += fabs(v.x) + fabs(v.y);
    } else {
      acc[0] = fmax(acc[0], fmax(fabs(v.x), fabs(v.y)));
    }
  }
  if (KIND == RED_AMAX) {
    block_reduce_store<1, OP_MAX>(acc, partials, 0, sm);
  } else {
    block_reduce_store<1, OP_SUM>(acc, partials, 0, sm);
  }
}

int k_reduce1(Ctx *c, int kind, const double *x, const double *y, int64_t n, double *out) {
  count_bytes(c, kind == RED_DOT ? 2 : 1, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid));
  switch (kind) {
    case RED_DOT:
      PO_LAUNCH(reduce1_kernel<RED_DOT>, grid, x, y, n, c->d_partials);
      break;
    case RED_SUMSQ:
      PO_LAUNCH(reduce1_kernel<RED_SUMSQ>, grid, x, y, n, c->d_partials);
      break;
    case RED_ASUM:
      PO_LAUNCH(reduce1_kernel<RED_ASUM>, grid, x, y, n, c->d_partials);
      break;
    default:
      PO_LAUNCH(reduce1_kernel<RED_AMAX>, grid, x, y, n, c->d_partials);
      break;
  }
  if (kind == RED_AMAX) return reduce_finish(c, grid, 0, 0, 1, out);
  return reduce_finish(c, grid, 1, 0, 0, out);
}

// smallest and largest entry (the uniformity check of a grouped sparse Jacobian, Problem::csrValuesChanged)
__global__ void __launch_bounds__(kBlock)
    minmax_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4];
  double mn[1] = {INFINITY}, mx[1] = {-INFINITY};
  const int64_t npairs = n >> 1;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
    const double2 v = ld2(x, q, n);
    mn[0] = fmin(mn[0], fmin(v.x, v.y));
    mx[0] = fmax(mx[0], fmax(v.x, v.y));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    mn[0] = fmin(mn[0], x[n - 1]);
    mx[0] = fmax(mx[0], x[n - 1]);
  }
  block_reduce_store<1, OP_MIN>(mn, partials, 0, sm);
  block_reduce_store<1, OP_MAX>(mx, partials, 1, sm);
}
int k_minmax(Ctx *c, const double *x, int64_t n, double out[2]) {
  // (collective: a rank without entries takes part with {+inf, -inf})
  if (n < 0) n = 0;
  count_bytes(c, 1, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 2));
  PO_LAUNCH(minmax_kernel, grid, x, n, c->d_partials);
  return reduce_finish(c, grid, 0, 1, 1, out, true);
}

// ---------------------------------------------------------------------------------------------
// mdot: out_j = sum_i x_i V_j[i]   (ParOptBasicVec::mdot, src/ParOptVec.cpp:152-170)
// x is read once per block of NVB panel columns; the NVB accumulators live in VGPRs.
// Algorithmic traffic: 8*(nvecs+1)*n bytes.
// ---------------------------------------------------------------------------------------------
template <int NVB>
__global__ void __launch_bounds__(kBlock)
    mdot_kernel(const double *__restrict__ x, PtrTable V, int j0, int64_t n,
                double *__restrict__ partials) {
  // tools/tune_mdot.hip on MI355X (n = 50 M, 32 columns): this plain form -- hipcc keeps one or
  // two 16-byte loads in flight per lane at ~6 waves per SIMD -- with non-temporal loads and 4-6
  // workgroups per CU streams at 6.4-6.5 TB/s; explicit 8-deep register batches reach 6.1-6.3.
  __shared__ double sm[4 * NVB];
  double acc[NVB];
  const double *vp[NVB];
#pragma unroll
  for (int j = 0; j < NVB; j++) {
    acc[j] = 0.0;
    vp[j] = V.p[j0 + j];
  }
  PO_PAIR_LOOP(q, n) {
    const f64x2 xv = *reinterpret_cast<const f64x2 *>(x + 2 * q);
#pragma unroll
    for (int j = 0; j < NVB; j++) {
      const f64x2 w = ld_stream(vp[j] + 2 * q);
      acc[j] = fma(xv.x, w.x, fma(xv.y, w.y, acc[j]));
    }
  }
  block_reduce_store<NVB, OP_SUM>(acc, partials, j0, sm);
}

#define PO_MDOT_CASE(NVB)                                                          \
  case NVB:                                                                        \
    PO_LAUNCH(mdot_kernel<NVB>, grid, x, pt, j0, n, c->d_partials);                \
    break;

int k_mdot_launch(Ctx *c, const double *x, const double *const *V, int nv, int64_t n, int *nblocks) {
  count_bytes(c, nv + 1, n);
  if (nv > kMaxPanel) {
    set_error("mdot of %d vectors exceeds kMaxPanel=%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  const int grid = grid_for(c, n, 5);
  PO_TRY(ensure_partials(c, (size_t)grid * (nv + 32)));
  PtrTable pt;
  CoefTable ct;
  fill_tables(nullptr, V, nv, &ct, &pt);
  // panel blocks of 32 / 24 / 16 columns, then one exact block of 1..15: x is re-read once per
  // block (nv = 42 -> 32 + 10: 44 streams instead of the ideal 43)
  int j0 = 0;
  while (j0 < nv) {
    const int rem = nv - j0;
    const int w = rem >= 32 ? 32 : (rem >= 24 ? 24 : (rem >= 16 ? 16 : rem));
    switch (w) {
      PO_MDOT_CASE(1) PO_MDOT_CASE(2) PO_MDOT_CASE(3) PO_MDOT_CASE(4) PO_MDOT_CASE(5)
      PO_MDOT_CASE(6) PO_MDOT_CASE(7) PO_MDOT_CASE(8) PO_MDOT_CASE(9) PO_MDOT_CASE(10)
      PO_MDOT_CASE(11) PO_MDOT_CASE(12) PO_MDOT_CASE(13) PO_MDOT_CASE(14) PO_MDOT_CASE(15)
      PO_MDOT_CASE(16) PO_MDOT_CASE(24) PO_MDOT_CASE(32)
    }
    j0 += w;
  }
  *nblocks = grid;
  return PO_OK;
}

int k_mdot(Ctx *c, const double *x, const double *const *V, int nv, int64_t n, double *out) {
  if (nv <= 0) return PO_OK;
  if (nv > kMaxPanel) {  // wider than one launch's pointer table: slabs (x is re-read once per slab)
    for (int j0 = 0; j0 < nv; j0 += kMaxPanel)
      PO_TRY(k_mdot(c, x, V + j0, nv - j0 > kMaxPanel ? kMaxPanel : nv - j0, n, out + j0));
    return PO_OK;
  }
  int grid = 0;
  const bool timed = c->time_mdot_nv == nv;  // po_ctx_time_mdot: HIP events on the launch stream
  // the event pair is read once the stream has been synchronised: right away, or when the enclosing batch is flushed
  // (one timed launch per batch: a second one would reuse the events, so the batch is flushed first)
  if (timed && c->mdot_timing_pending) PO_TRY(batch_flush(c));
  if (timed) PO_HIP(hipEventRecord(c->ev_mdot0, c->stream));
  PO_TRY(k_mdot_launch(c, x, V, nv, n, &grid));
  if (timed) PO_HIP(hipEventRecord(c->ev_mdot1, c->stream));
  const bool defer = timed && c->batch_depth > 0;
  PO_TRY(reduce_finish(c, grid, nv, 0, 0, out, timed && !defer));
  if (timed) {
    auto harvest = [c] {
      // (the results reached the host through the completion flag, possibly without a stream synchronisation: the
      // event is waited for itself, so that no sample is dropped as "not ready")
      float ms = 0.0f;
      if (hipEventSynchronize(c->ev_mdot1) == hipSuccess &&
          hipEventElapsedTime(&ms, c->ev_mdot0, c->ev_mdot1) == hipSuccess) {
        c->mdot_ms += ms;
        c->mdot_count++;
      }
      c->mdot_timing_pending = false;
    };
    if (defer) c->mdot_timing_pending = true;
    after_reduce(c, harvest);
  }
  return PO_OK;
}

// (the weighted Gram W = P^T diag(d) P lives in wgram.hip)

// ---------------------------------------------------------------------------------------------
// interior-point element kernels
// ---------------------------------------------------------------------------------------------
struct BE {  // per-element bound data
  bool L, U;
  double xl, xu, zl, zu;
};
__device__ __forceinline__ BE bound_elem(double x, double lb, double ub, double zl, double zu,
                                         double maxb, int use_lower, int use_upper) {
  BE e;
  e.L = use_lower && (lb > -maxb);
  e.U = use_upper && (ub < maxb);
  e.xl = e.L ? (x - lb) : 1.0;
  e.xu = e.U ? (ub - x) : 1.0;
  e.zl = zl;
  e.zu = zu;
  return e;
}
// e0 / e1 from the five loaded pairs _x, _lb, _ub, _zl, _zu (double2) of row pair q
#define PO_MAKE_BOUNDS(b, q, n)                                                          \
  const bool _has2 = (2 * q + 1 < n);                                                    \
  const BE e0 = bound_elem(_x.x, _lb.x, _ub.x, _zl.x, _zu.x, (b).max_bound, (b).use_lower, \
                           (b).use_upper);                                               \
  BE e1 = bound_elem(_x.y, _lb.y, _ub.y, _zl.y, _zu.y, (b).max_bound, (b).use_lower,     \
                     (b).use_upper);                                                     \
  if (!_has2) {                                                                          \
    e1.L = false;                                                                        \
    e1.U = false;                                                                        \
    e1.xl = 1.0;                                                                         \
    e1.xu = 1.0;                                                                         \
  }
#define PO_LOAD_BOUNDS(b, q, n)                                                         \
  const double2 _x = ld2((b).x, q, n), _lb = ld2((b).lb, q, n), _ub = ld2((b).ub, q, n), \
                _zl = ld2((b).zl, q, n), _zu = ld2((b).zu, q, n);                        \
  PO_MAKE_BOUNDS(b, q, n)

// rx, complementarity, residual norms -----------------------------------------------------------
// sums: {comp product, active count, l1|rx|, l1|rzl|, l1|rzu|, l2 rx, l2 rzl, l2 rzu}; maxs: {rx, rzl, rzu}
__device__ __forceinline__ void res_bound_acc(const BE &e, double beta_mu, double *sums, double *maxs,
                                              int il1, int il2, int imax) {
  if (e.L) {
    sums[0] += e.zl * e.xl;
    sums[1] += 1.0;
    const double r = fabs(-(e.xl * e.zl - beta_mu));
    sums[il1] += r;
    sums[il2] += r * r;
    maxs[imax] = fmax(maxs[imax], r);
  }
  if (e.U) {
    sums[0] += e.zu * e.xu;
    sums[1] += 1.0;
    const double r = fabs(-(e.xu * e.zu - beta_mu));
    sums[il1 + 1] += r;
    sums[il2 + 1] += r * r;
    maxs[imax + 1] = fmax(maxs[imax + 1], r);
  }
}
// max |rzl|, |rzu| for a SECOND barrier parameter (the one the monotone strategy would switch to, known before the
// pass: a function of the current one alone): when the switch happens the pass over the bound data that only
// re-evaluates these two maxima (res_norms_kernel) and its host synchronisation are not needed.  Same expression as
// res_bound_acc: the same bits as that pass would produce (a maximum does not depend on the summation order).
__device__ __forceinline__ void res_bound_max2(const BE &e, double beta_mu2, double *m2) {
  if (e.L) m2[0] = fmax(m2[0], fabs(-(e.xl * e.zl - beta_mu2)));
  if (e.U) m2[1] = fmax(m2[1], fabs(-(e.xu * e.zu - beta_mu2)));
}
__global__ void __launch_bounds__(kBlock)
    kkt_res_kernel(Bounds b, const double *__restrict__ g, PtrTable A, CoefTable z, int nc,
                   double beta_mu, int64_t n, double *__restrict__ rx, double *__restrict__ yqn,
                   double beta_mu2, GroupCol gcol, double gcoef, double *__restrict__ partials) {
  __shared__ double sm[4 * 8];
  double sums[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double maxs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 gv = ld2(g, q, n);
    double2 yv = make_double2(0.0, 0.0);
    if (yqn) yv = ld2(yqn, q, n);
    double2 r;
    r.x = (b.use_lower ? _zl.x : 0.0);
    r.y = (b.use_lower ? _zl.y : 0.0);
    if (b.use_upper) {
      r.x += -1.0 * _zu.x;
      r.y += -1.0 * _zu.y;
    }
    r.x += -1.0 * gv.x;
    r.y += -1.0 * gv.y;
    // (the grouped column's entries are requested before the panel: its two dependent loads overlap the panel's)
    GRaw graw;
    if (gcol.w) graw = gcol_request(gcol, q, n);
    double2 ps = panel_sum(A, z, nc, q);
    if (gcol.w) {
      const f64x2 gcv = gcol_value(gcol, graw);
      ps.x += gcoef * gcv.x;
      ps.y += gcoef * gcv.y;
    }
    r.x += ps.x;
    r.y += ps.y;
    if (!_has2) r.y = 0.0;
    st2(rx, q, n, r);
    if (yqn) {
      // second bracket of the quasi-Newton gradient difference, y += [lo]zl - [up]zu - rx at the new point, in the
      // operation order of the panel_axpy pass it replaces (computeStepAndUpdate)
      const double cl = b.use_lower ? 1.0 : 0.0, cu = b.use_upper ? -1.0 : 0.0;
      yv.x = __dadd_rn(yv.x, __fma_rn(-1.0, r.x, __fma_rn(cu, _zu.x, __fma_rn(cl, _zl.x, 0.0))));
      yv.y = __dadd_rn(yv.y, __fma_rn(-1.0, r.y, __fma_rn(cu, _zu.y, __fma_rn(cl, _zl.y, 0.0))));
      st2(yqn, q, n, yv);
    }
    maxs[0] = fmax(maxs[0], fmax(fabs(r.x), fabs(r.y)));
    sums[2] += fabs(r.x) + fabs(r.y);
    sums[5] += r.x * r.x + r.y * r.y;
    res_bound_acc(e0, beta_mu, sums, maxs, 3, 6, 1);
    res_bound_acc(e1, beta_mu, sums, maxs, 3, 6, 1);
    if (beta_mu2 >= 0.0) {
      res_bound_max2(e0, beta_mu2, maxs + 3);
      res_bound_max2(e1, beta_mu2, maxs + 3);
    }
  }
  block_reduce_store<8, OP_SUM>(sums, partials, 0, sm);
  if (beta_mu2 >= 0.0) {
    block_reduce_store<5, OP_MAX>(maxs, partials, 8, sm);
  } else {
    double m3[3] = {maxs[0], maxs[1], maxs[2]};
    block_reduce_store<3, OP_MAX>(m3, partials, 8, sm);
  }
}

int k_kkt_res(Ctx *c, const Bounds &b, const double *g, const double *const *A, const double *z,
              int nc, double beta_mu, int64_t n, double *rx, double *out, double *yqn, double beta_mu2,
              const GroupCol *gcol, double gcoef) {
  PO_TRY(gcol_check(gcol, n, "k_kkt_res"));
  if (nc > kMaxPanel) {  // wide A^T z: one collapsed column
    const double *w = nullptr, one = 1.0;
    PO_TRY(collapse_range(c, 0, z, A, 0, nc, n, &w));
    return k_kkt_res(c, b, g, &w, &one, 1, beta_mu, n, rx, out, yqn, beta_mu2, gcol, gcoef);
  }
  count_bytes(c, 7 + nc + (yqn ? 2 : 0), n);
  const int grid = grid_for(c, n, kBpcPanel);
  PO_TRY(ensure_partials(c, (size_t)grid * 13));
  PtrTable pt;
  CoefTable ct;
  fill_tables(z, A, nc, &ct, &pt);
  if (gcol) count_bytes(c, 1.0, gcol->nwcon);
  PO_LAUNCH(kkt_res_kernel, grid, b, g, pt, ct, nc, beta_mu, n, rx, yqn, beta_mu2, gcol ? *gcol : GroupCol(), gcoef,
            c->d_partials);
  return reduce_finish(c, grid, 8, 0, beta_mu2 >= 0.0 ? 5 : 3, out);
}

// the mu-dependent part only: sums {comp product, count, -, l1 rzl, l1 rzu, -, l2 rzl, l2 rzu}
// (slots 2 and 5 stay zero so the layout matches kkt_res_kernel), maxs {-, rzl, rzu}
__global__ void __launch_bounds__(kBlock)
    res_norms_kernel(Bounds b, double beta_mu, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 8];
  double sums[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double maxs[3] = {0.0, 0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    res_bound_acc(e0, beta_mu, sums, maxs, 3, 6, 1);
    res_bound_acc(e1, beta_mu, sums, maxs, 3, 6, 1);
  }
  block_reduce_store<8, OP_SUM>(sums, partials, 0, sm);
  block_reduce_store<3, OP_MAX>(maxs, partials, 8, sm);
}

int k_res_norms(Ctx *c, const Bounds &b, double beta_mu, int64_t n, double out[11]) {
  count_bytes(c, 5, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 11));
  PO_LAUNCH(res_norms_kernel, grid, b, beta_mu, n, c->d_partials);
  return reduce_finish(c, grid, 8, 0, 3, out);
}

// Dinv ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dinv_elem(const BE &e, double diag) {
  double v = diag;
  if (e.L) v += e.zl / e.xl;
  if (e.U) v += e.zu / e.xu;
  return 1.0 / v;
}
__global__ void __launch_bounds__(kBlock)
    dinv_kernel(Bounds b, double diag, const double *__restrict__ hdiag, int64_t n,
                double *__restrict__ dinv) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    double2 h = make_double2(0.0, 0.0);
    if (hdiag) h = ld2(hdiag, q, n);  // use_diag_hessian: b0 -> h_i (:1840-1842)
    st2(dinv, q, n, make_double2(dinv_elem(e0, diag + h.x), dinv_elem(e1, diag + h.y)));
  }
}
int k_dinv(Ctx *c, const Bounds &b, double diag, int64_t n, double *dinv, const double *hdiag) {
  count_bytes(c, 6 + (hdiag ? 1 : 0), n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(dinv_kernel, grid_for(c, n), b, diag, hdiag, n, dinv);
  return PO_OK;
}

// t = Dinv * d1 -----------------------------------------------------------------------------------
// cl / cu: Mehrotra corrector terms (addMehrotraCorrectorResidual :1765-1788), zero otherwise
__device__ __forceinline__ double d1_elem(const BE &e, double rx, double beta_mu, double cl = 0.0,
                                          double cu = 0.0) {
  double d1 = rx;
  if (e.L) d1 += (-(e.xl * e.zl - beta_mu) - cl) / e.xl;
  if (e.U) d1 -= (-(e.xu * e.zu - beta_mu) + cu) / e.xu;
  return d1;
}
__global__ void __launch_bounds__(kBlock)
    d1_kernel(Bounds b, const double *__restrict__ rx, const double *__restrict__ dinv,
              double beta_mu, const double *__restrict__ cl, const double *__restrict__ cu, int64_t n,
              double *__restrict__ t) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 r = ld2(rx, q, n);
    const double2 dv = dinv ? ld2(dinv, q, n) : make_double2(1.0, 1.0);  // null: raw d1
    double2 c0 = make_double2(0.0, 0.0), c1 = c0;
    if (cl) {
      c0 = ld2(cl, q, n);
      c1 = ld2(cu, q, n);
    }
    st2(t, q, n,
        make_double2(dv.x * d1_elem(e0, r.x, beta_mu, c0.x, c1.x),
                     dv.y * d1_elem(e1, r.y, beta_mu, c0.y, c1.y)));
  }
}
int k_d1(Ctx *c, const Bounds &b, const double *rx, const double *dinv, double beta_mu, int64_t n,
         double *t, const double *cl, const double *cu) {
  count_bytes(c, 7 + (dinv ? 1 : 0) + (cl ? 2 : 0), n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(d1_kernel, grid_for(c, n), b, rx, dinv, beta_mu, cl, cu, n, t);
  return PO_OK;
}

// Dinv and t = Dinv * d1 in one pass over the bound data (setUpKKTDiagSystem :1864-1910 + the d1 build of the
// solve that follows :2091-2108, when its right-hand side is known at set-up time): bit-identical to
// dinv_kernel followed by d1_kernel
// raw != 0: t receives d1 itself (the sparse-constraint path applies its block solve to the raw right-hand side),
// as d1_kernel with dinv == nullptr
__global__ void __launch_bounds__(kBlock)
    dinv_d1_kernel(Bounds b, double diag, const double *__restrict__ hdiag, const double *__restrict__ rx,
                   double beta_mu, int64_t n, double *__restrict__ dinv, double *__restrict__ t, int raw) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    double2 h = make_double2(0.0, 0.0);
    if (hdiag) h = ld2(hdiag, q, n);
    const double2 r = ld2(rx, q, n);
    const double2 dv = make_double2(dinv_elem(e0, diag + h.x), dinv_elem(e1, diag + h.y));
    st2(dinv, q, n, dv);
    const double w0 = raw ? 1.0 : dv.x, w1 = raw ? 1.0 : dv.y;
    st2(t, q, n, make_double2(w0 * d1_elem(e0, r.x, beta_mu), w1 * d1_elem(e1, r.y, beta_mu)));
  }
}
int k_dinv_d1(Ctx *c, const Bounds &b, double diag, const double *hdiag, const double *rx, double beta_mu,
              int64_t n, double *dinv, double *t, int raw) {
  count_bytes(c, 8 + (hdiag ? 1 : 0), n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(dinv_d1_kernel, grid_for(c, n), b, diag, hdiag, rx, beta_mu, n, dinv, t, raw);
  return PO_OK;
}

// corrector products of the affine step: cl = [L] px*pzl, cu = [U] px*pzu
__global__ void __launch_bounds__(kBlock)
    corrector_kernel(Bounds b, const double *__restrict__ px, const double *__restrict__ pzl,
                     const double *__restrict__ pzu, int64_t n, double *__restrict__ cl,
                     double *__restrict__ cu) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    st2(cl, q, n, make_double2(e0.L ? p.x * l.x : 0.0, e1.L ? p.y * l.y : 0.0));
    st2(cu, q, n, make_double2(e0.U ? p.x * u.x : 0.0, e1.U ? p.y * u.y : 0.0));
  }
}
int k_corrector(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu,
                int64_t n, double *cl, double *cu) {
  count_bytes(c, 8, n);  // x, lb, ub (the bound predicates), px, pzl, pzu in; cl, cu out (zl / zu are not used)
  if (n <= 0) return PO_OK;
  PO_LAUNCH(corrector_kernel, grid_for(c, n), b, px, pzl, pzu, n, cl, cu);
  return PO_OK;
}

// Corrector right-hand side of the predictor-corrector strategy in ONE pass (round 6): the corrector products of the
// affine step (corrector_kernel), t = Dinv o d1 with them (d1_kernel) and the panel products P^T t (mdot_kernel) --
// three launches, two n-sized writes (cl, cu) and their re-reads, and the re-read of t by the product pass.  Same
// expressions, same grid and per-thread order as those three kernels: the same bits in t and in the products.
// (cl, cu) are not stored: the corrector solve (solve2c_kernel) re-forms them from the affine step it overwrites.
template <int NVB>
__global__ void __launch_bounds__(kBlock)
    corr_d1_dots_kernel(Bounds b, const double *__restrict__ px, const double *__restrict__ pzl,
                        const double *__restrict__ pzu, const double *__restrict__ rx,
                        const double *__restrict__ dinv, double beta_mu, PtrTable V, int64_t n,
                        double *__restrict__ t, double *__restrict__ partials) {
  __shared__ double sm[4 * NVB];
  double acc[NVB];
  const double *vp[NVB];
#pragma unroll
  for (int j = 0; j < NVB; j++) {
    acc[j] = 0.0;
    vp[j] = V.p[j];
  }
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    const double2 r = ld2(rx, q, n), dv = ld2(dinv, q, n);
    // (explicitly rounded products: corrector_kernel stores them, and a product contracted into the subtraction of
    // d1_elem would be another number)
    const double c0x = e0.L ? __dmul_rn(p.x, l.x) : 0.0, c0y = e1.L ? __dmul_rn(p.y, l.y) : 0.0;
    const double c1x = e0.U ? __dmul_rn(p.x, u.x) : 0.0, c1y = e1.U ? __dmul_rn(p.y, u.y) : 0.0;
    double2 tv = make_double2(__dmul_rn(dv.x, d1_elem(e0, r.x, beta_mu, c0x, c1x)),
                              __dmul_rn(dv.y, d1_elem(e1, r.y, beta_mu, c0y, c1y)));
    if (!_has2) tv.y = 0.0;  // what st2 stores, and what the product pass would have read back
    st2(t, q, n, tv);
#pragma unroll
    for (int j = 0; j < NVB; j++) {
      const f64x2 w = ld_stream(vp[j] + 2 * q);
      acc[j] = fma(tv.x, w.x, fma(tv.y, w.y, acc[j]));
    }
  }
  block_reduce_store<NVB, OP_SUM>(acc, partials, 0, sm);
}

#define PO_CORR_CASE(NVB)                                                                              \
  case NVB:                                                                                            \
    PO_LAUNCH(corr_d1_dots_kernel<NVB>, grid, b, px, pzl, pzu, rx, dinv, beta_mu, pt, n, t, c->d_partials); \
    break;
int k_corr_d1_dots(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu, const double *rx,
                   const double *dinv, double beta_mu, const double *const *V, int nv, int64_t n, double *t,
                   double *out) {
  if (nv < 1 || nv > kCorrDotsMax) {
    set_error("k_corr_d1_dots: %d panel columns (1..%d)", nv, kCorrDotsMax);
    return PO_ERR_ARG;
  }
  count_bytes(c, nv + 11, n);
  const int grid = grid_for(c, n, 5);  // mdot's grid: the products come out with mdot's bits
  PO_TRY(ensure_partials(c, (size_t)grid * nv));
  PtrTable pt;
  CoefTable ct;
  fill_tables(nullptr, V, nv, &ct, &pt);
  switch (nv) {
    PO_CORR_CASE(1) PO_CORR_CASE(2) PO_CORR_CASE(3) PO_CORR_CASE(4) PO_CORR_CASE(5) PO_CORR_CASE(6) PO_CORR_CASE(7)
    PO_CORR_CASE(8) PO_CORR_CASE(9) PO_CORR_CASE(10) PO_CORR_CASE(11) PO_CORR_CASE(12) PO_CORR_CASE(13)
    PO_CORR_CASE(14) PO_CORR_CASE(15)
  }
  return reduce_finish(c, grid, nv, 0, 0, out);
}

// second half of the bordered solve ------------------------------------------------------------
struct Step3 {
  double px, pzl, pzu;
};
template <int REFINE>
__device__ __forceinline__ Step3 solve2_elem(const BE &e, double dx, double beta_mu, double px0,
                                             double pzl0, double pzu0, double cl = 0.0,
                                             double cu = 0.0) {
  Step3 s;
  const double rzl = e.L ? -(e.xl * e.zl - beta_mu) - cl : 0.0;
  const double rzu = e.U ? -(e.xu * e.zu - beta_mu) + cu : 0.0;
  if (REFINE) {
    const double rzl2 = e.L ? rzl - (e.xl * pzl0 + px0 * e.zl) : 0.0;
    const double rzu2 = e.U ? rzu - (e.xu * pzu0 - px0 * e.zu) : 0.0;
    s.px = px0 + dx;
    s.pzl = pzl0 + (e.L ? (rzl2 - e.zl * dx) / e.xl : 0.0);
    s.pzu = pzu0 + (e.U ? (rzu2 + e.zu * dx) / e.xu : 0.0);
  } else {
    s.px = dx;
    s.pzl = e.L ? (rzl - e.zl * dx) / e.xl : 0.0;
    s.pzu = e.U ? (rzu + e.zu * dx) / e.xu : 0.0;
  }
  return s;
}
// fraction-to-boundary minima (computeMaxStep :2958-2980, 3064-3090): the primal tests use the
// raw x-lb / ub-x and are not masked by the bound predicates, exactly as the reference.
__device__ __forceinline__ void max_step_elem(const Bounds &b, double x, double lb, double ub,
                                              double zl, double zu, const Step3 &s, double tau,
                                              double &mx, double &mz) {
  if (b.use_lower) {
    if (s.px < 0.0) mx = fmin(mx, -tau * (x - lb) / s.px);
    if (s.pzl < 0.0) mz = fmin(mz, -tau * zl / s.pzl);
  }
  if (b.use_upper) {
    if (s.px > 0.0) mx = fmin(mx, tau * (ub - x) / s.px);
    if (s.pzu < 0.0) mz = fmin(mz, -tau * zu / s.pzu);
  }
}

// refinement residual folded into the next right-hand side -----------------------------------------
__device__ __forceinline__ double res_step_elem(const BE &e, double rx, double acc, double diag,
                                                double px, double pzl, double pzu, double dinv,
                                                double beta_mu, int use_lower, int use_upper) {
  double r = rx - diag * px + acc;
  if (use_lower) r += pzl;
  if (use_upper) r -= pzu;
  double d1 = r;
  if (e.L) {
    const double rzl2 = -(e.xl * e.zl - beta_mu) - (e.xl * pzl + px * e.zl);
    d1 += rzl2 / e.xl;
  }
  if (e.U) {
    const double rzu2 = -(e.xu * e.zu - beta_mu) - (e.xu * pzu - px * e.zu);
    d1 -= rzu2 / e.xu;
  }
  return dinv * d1;
}
// REFINE: accumulate into (px, pzl, pzu).  FUSE (first pass only): also stream the second
// coefficient set and emit t' = Dinv*d1' of the refinement right-hand side in the same pass
// (what res_step_kernel computes), in place of t.
template <int REFINE, int FUSE>
__global__ void __launch_bounds__(kBlock)
    solve2_kernel(Bounds b, const double *t, const double *__restrict__ dinv, CoefTable alpha,
                  CoefTable coef2, PtrTable P, int nv, double beta_mu, double tau,
                  const double *__restrict__ rx, double diag, int64_t n, double *__restrict__ px,
                  double *__restrict__ pzl, double *__restrict__ pzu, double *tout,
                  double *__restrict__ va, int nca, const double *__restrict__ cl,
                  const double *__restrict__ cu, double *__restrict__ partials) {
  __shared__ double sm[4 * 2];
  double mins[2] = {1.0, 1.0};
  PO_PAIR_LOOP(q, n) {
    // the first nca columns are the constraint gradients: their partial sum A^T pz is kept in `va`
    // (accumulated over the refinement pass) for the quasi-Newton gradient difference
    const double2 accA = panel_sum(P, alpha, nca, q);
    if (va) {
      double2 w = accA;
      if (REFINE) {
        const double2 w0 = ld2(va, q, n);
        w.x += w0.x;
        w.y += w0.y;
      }
      st2(va, q, n, w);
    }
    double2 acc, acc2 = make_double2(0.0, 0.0);
    if (FUSE) {
      panel_sum2(P, alpha, coef2, nv, q, acc, acc2, nca);
      acc2.x += accA.x;  // coef2 == alpha on the constraint columns
      acc2.y += accA.y;
    } else {
      acc = panel_sum(P, alpha, nv, q, nca);
    }
    acc.x += accA.x;
    acc.y += accA.y;
    PO_LOAD_BOUNDS(b, q, n);
    const double2 tv = ld2(t, q, n), dv = ld2(dinv, q, n);
    const double dx0 = tv.x + dv.x * acc.x, dx1 = tv.y + dv.y * acc.y;
    double2 p0 = make_double2(0.0, 0.0), l0 = p0, u0 = p0;
    if (REFINE) {
      p0 = ld2(px, q, n);
      l0 = ld2(pzl, q, n);
      u0 = ld2(pzu, q, n);
    }
    double2 c0 = make_double2(0.0, 0.0), c1 = c0;
    if (!REFINE && !FUSE && cl) {
      c0 = ld2(cl, q, n);
      c1 = ld2(cu, q, n);
    }
    const Step3 s0 = solve2_elem<REFINE>(e0, dx0, beta_mu, p0.x, l0.x, u0.x, c0.x, c1.x);
    Step3 s1 = solve2_elem<REFINE>(e1, dx1, beta_mu, p0.y, l0.y, u0.y, c0.y, c1.y);
    if (!_has2) s1.px = s1.pzl = s1.pzu = 0.0;
    st2(px, q, n, make_double2(s0.px, s1.px));
    st2(pzl, q, n, make_double2(s0.pzl, s1.pzl));
    st2(pzu, q, n, make_double2(s0.pzu, s1.pzu));
    if (FUSE) {
      const double2 r = ld2(rx, q, n);
      st2(tout, q, n,
          make_double2(res_step_elem(e0, r.x, acc2.x, diag, s0.px, s0.pzl, s0.pzu, dv.x, beta_mu,
                                     b.use_lower, b.use_upper),
                       res_step_elem(e1, r.y, acc2.y, diag, s1.px, s1.pzl, s1.pzu, dv.y, beta_mu,
                                     b.use_lower, b.use_upper)));
    }
    max_step_elem(b, _x.x, _lb.x, _ub.x, _zl.x, _zu.x, s0, tau, mins[0], mins[1]);
    if (_has2) max_step_elem(b, _x.y, _lb.y, _ub.y, _zl.y, _zu.y, s1, tau, mins[0], mins[1]);
  }
  block_reduce_store<2, OP_MIN>(mins, partials, 0, sm);
}

// Refinement pass that RECOMPUTES the first step instead of reading it: the first pass (solve2_dots_kernel with
// store_step = 0) wrote only the refinement right-hand side t2, so here
//   px1 = t1 + Dinv*(P a1), (pzl1, pzu1) from px1          (the first solve, in registers)
//   px  = px1 + t2 + Dinv*(P a2), pzl, pzu accordingly      (solve2_elem<1>, as solve2_kernel<1,0>)
// in one sweep over P with both coefficient sets.  Four output streams and three input streams less per iteration
// than storing and re-reading the first step.
// RECT != 0: the refinement right-hand side t2 is recomputed as well (from the residual coefficients ar, rx and diag,
// as solve2_dots_kernel formed it when it took its panel products), so the first pass stores nothing at all.
// MERIT != 0: the pass also takes, for the FINAL step it has in registers, every sum scaleKKTStep's complementarity
// check (computeCompStep :2825-2923) and evalMeritInitDeriv (:3652-3714) need of it -- the separate pass over
// (x, lb, ub, zl, zu, px, pzl, pzu, g) disappears for one more input stream (g).  The complementarity at the scaled
// step is a polynomial in the two step lengths,
//   sum_L (zl + az pzl)(x - lb + ax px) + sum_U (zu + az pzu)(ub - x - ax px) = S00 + ax S10 + az S01 + ax az S11,
// so the pass needs no step length: S00 is the complementarity product of the iterate (known from the residual
// pass), S10 = sum_L zl px - sum_U zu px, S01 = sum_L pzl (x - lb) + sum_U pzu (ub - x), S11 = sum_L pzl px -
// sum_U pzu px.  sums {S10, S01, S11, ppos, pneg, g.px, px.px}, max {|px|}; the log-barrier sums of the iterate
// come from the accepted trial point of the previous line search (trial_kernel: same elements, same order).
template <int RECT, int MERIT>
__global__ void __launch_bounds__(kBlock)
    solve2r_kernel(Bounds b, const double *__restrict__ t1, const double *__restrict__ t2,
                   const double *__restrict__ dinv, CoefTable a1, CoefTable a2, CoefTable ar, PtrTable P, int nv,
                   double beta_mu, double tau, const double *__restrict__ rx, double diag, int64_t n,
                   double *__restrict__ px, double *__restrict__ pzl, double *__restrict__ pzu,
                   double *__restrict__ va, int nca, int ca0, VirtCols vc, const double *__restrict__ g,
                   double dinv_diag, GroupCol gcol, double gc1, double gc2, double *__restrict__ partials) {
  __shared__ double sm[4 * 8];
  double mins[2] = {1.0, 1.0};
  double ms[7] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double mx[1] = {0.0};
  PO_PAIR_LOOP(q, n) {
    GRaw graw;  // the grouped column's entries: requested first, used behind the panel sums
    if (!RECT && gcol.w) graw = gcol_request(gcol, q, n);
    const double2 zero2 = make_double2(0.0, 0.0);
    double2 a1A = zero2, a2A = zero2, acc1 = zero2, acc2 = zero2, arA = zero2, accr = zero2;
    if (RECT) {
      // the constraint columns [ca0, ca0 + nca) are summed apart (A^T pz is kept), the rest in panel order:
      // [0, ca0) (the quasi-Newton columns when they lead the panel, unformed ones included), then the tail
      panel_sum3(P, a1, a2, ar, ca0 + nca, q, a1A, a2A, arA, ca0, vc);
      if (ca0 > 0) panel_sum3(P, a1, a2, ar, ca0, q, acc1, acc2, accr, 0, vc);
      panel_sum3(P, a1, a2, ar, nv, q, acc1, acc2, accr, ca0 + nca, vc);
      accr.x += arA.x;
      accr.y += arA.y;
    } else {
      panel_sum2(P, a1, a2, nca, q, a1A, a2A);
      panel_sum2(P, a1, a2, nv, q, acc1, acc2, nca);
      if (gcol.w) {  // one more column behind the panel (core.hpp: GroupCol), coefficients gc1 / gc2
        const f64x2 gv = gcol_value(gcol, graw);
        acc1.x += gc1 * gv.x;
        acc1.y += gc1 * gv.y;
        acc2.x += gc2 * gv.x;
        acc2.y += gc2 * gv.y;
      }
    }
    acc1.x += a1A.x;
    acc1.y += a1A.y;
    acc2.x += a2A.x;
    acc2.y += a2A.y;
    if (va) st2(va, q, n, make_double2(a2A.x + a1A.x, a2A.y + a1A.y));  // A^T pz of both solves
    PO_LOAD_BOUNDS(b, q, n);
    double2 tv, dv;
    if (t1) {
      tv = ld2(t1, q, n);
      dv = ld2(dinv, q, n);
    } else {
      // t1 == nullptr (RECT form): Dinv and t = Dinv o d1 of the first solve are re-formed from the bound data and rx
      // this pass loads anyway, exactly as dinv_d1_kernel formed them (same expressions: same bits) -- two input
      // streams less
      // (the pad element of an odd n has no bounds: with a zero diagonal -- sequential linear method -- 1 / 0 would
      // enter the sums as inf * 0; the stored Dinv and t hold exactly 0 there)
      const double2 r0 = ld2(rx, q, n);
      dv = make_double2(dinv_elem(e0, dinv_diag), _has2 ? dinv_elem(e1, dinv_diag) : 0.0);
      tv = make_double2(dv.x * d1_elem(e0, r0.x, beta_mu), _has2 ? dv.y * d1_elem(e1, r0.y, beta_mu) : 0.0);
    }
    const Step3 f0 = solve2_elem<0>(e0, tv.x + dv.x * acc1.x, beta_mu, 0.0, 0.0, 0.0);
    Step3 f1 = solve2_elem<0>(e1, tv.y + dv.y * acc1.y, beta_mu, 0.0, 0.0, 0.0);
    if (!_has2) f1.px = f1.pzl = f1.pzu = 0.0;
    double2 tw;
    if (RECT) {
      const double2 r = ld2(rx, q, n);
      tw.x = dv.x * res_step_elem(e0, r.x, accr.x, diag, f0.px, f0.pzl, f0.pzu, 1.0, beta_mu, b.use_lower, b.use_upper);
      tw.y = _has2 ? dv.y * res_step_elem(e1, r.y, accr.y, diag, f1.px, f1.pzl, f1.pzu, 1.0, beta_mu, b.use_lower,
                                          b.use_upper)
                   : 0.0;
    } else {
      tw = ld2(t2, q, n);
    }
    const Step3 s0 = solve2_elem<1>(e0, tw.x + dv.x * acc2.x, beta_mu, f0.px, f0.pzl, f0.pzu);
    Step3 s1 = solve2_elem<1>(e1, tw.y + dv.y * acc2.y, beta_mu, f1.px, f1.pzl, f1.pzu);
    if (!_has2) s1.px = s1.pzl = s1.pzu = 0.0;
    st2(px, q, n, make_double2(s0.px, s1.px));
    if (pzl) {  // (nullptr: lean step, the multiplier update recomputes them: kkt_res_update_kernel)
      st2(pzl, q, n, make_double2(s0.pzl, s1.pzl));
      st2(pzu, q, n, make_double2(s0.pzu, s1.pzu));
    }
    max_step_elem(b, _x.x, _lb.x, _ub.x, _zl.x, _zu.x, s0, tau, mins[0], mins[1]);
    if (_has2) max_step_elem(b, _x.y, _lb.y, _ub.y, _zl.y, _zu.y, s1, tau, mins[0], mins[1]);
    if (MERIT) {
      const double2 gv = ld2(g, q, n);
      if (e0.L) {
        ms[0] += _zl.x * s0.px;
        ms[1] += s0.pzl * e0.xl;
        ms[2] += s0.pzl * s0.px;
        if (s0.px > 0.0) ms[3] += s0.px / e0.xl; else ms[4] += s0.px / e0.xl;
      }
      if (e1.L) {
        ms[0] += _zl.y * s1.px;
        ms[1] += s1.pzl * e1.xl;
        ms[2] += s1.pzl * s1.px;
        if (s1.px > 0.0) ms[3] += s1.px / e1.xl; else ms[4] += s1.px / e1.xl;
      }
      if (e0.U) {
        ms[0] -= _zu.x * s0.px;
        ms[1] += s0.pzu * e0.xu;
        ms[2] -= s0.pzu * s0.px;
        if (s0.px > 0.0) ms[4] -= s0.px / e0.xu; else ms[3] -= s0.px / e0.xu;
      }
      if (e1.U) {
        ms[0] -= _zu.y * s1.px;
        ms[1] += s1.pzu * e1.xu;
        ms[2] -= s1.pzu * s1.px;
        if (s1.px > 0.0) ms[4] -= s1.px / e1.xu; else ms[3] -= s1.px / e1.xu;
      }
      ms[5] += gv.x * s0.px + gv.y * s1.px;
      ms[6] += s0.px * s0.px + s1.px * s1.px;
      mx[0] = fmax(mx[0], fmax(fabs(s0.px), fabs(s1.px)));
    }
  }
  if (MERIT) {  // slots: 7 sums, 2 minima, 1 maximum
    block_reduce_store<7, OP_SUM>(ms, partials, 0, sm);
    block_reduce_store<2, OP_MIN>(mins, partials, 7, sm);
    block_reduce_store<1, OP_MAX>(mx, partials, 9, sm);
  } else {
    block_reduce_store<2, OP_MIN>(mins, partials, 0, sm);
  }
}

int k_solve2r(Ctx *c, const Bounds &b, const double *t1, const double *t2, const double *dinv, const double *a1,
              const double *a2, const double *const *P, int nv, double beta_mu, double tau, int64_t n, double *px,
              double *pzl, double *pzu, double *va, int nca, double out[2], const double *ar, const double *rx,
              double diag, int ca0, const double *const *vs, int nvirt, double b0v, const double *g,
              double *merit_out, double dinv_diag, const GroupCol *gcol, double gc1, double gc2) {
  count_bytes(c, nv + nvirt + 7 + (t1 ? 2 : 0) + (pzl ? 2 : 0) + (va ? 1 : 0) + (g ? 1 : 0), n);
  PO_TRY(gcol_check(gcol, n, "k_solve2r"));
  if (gcol && t2 == nullptr) {
    set_error("k_solve2r: a grouped column is only taken in the stored right-hand side form");
    return PO_ERR_ARG;
  }
  if (gcol) count_bytes(c, 1.0, gcol->nwcon);
  const GroupCol gcv = gcol ? *gcol : GroupCol();
  if (!t1 && (t2 != nullptr || !rx)) {
    set_error("k_solve2r: Dinv / t can only be re-formed in the recomputed right-hand side form");
    return PO_ERR_ARG;
  }
  if (nv > kMaxPanel) {
    set_error("panel of %d vectors exceeds kMaxPanel=%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  const int grid = grid_for(c, n, kBpcPanel);
  PO_TRY(ensure_partials(c, (size_t)grid * 10));
  PtrTable pt;
  CoefTable ct1, ct2, ctr;
  fill_tables(a1, P, nv, &ct1, &pt);
  fill_tables(a2, P, nv, &ct2, &pt);
  fill_tables(ar, P, nv, &ctr, &pt);
  VirtCols vc;
  vc.count = 0;
  vc.b0 = b0v;
  for (int j = 0; j < kMaxVirt; j++) vc.s[j] = nullptr;
  if (nvirt > 0) {
    if (nvirt > kMaxVirt || t2 != nullptr || ca0 < nvirt) {
      set_error("k_solve2r: %d unformed columns (at most %d, leading the panel, recomputed right-hand side only)", nvirt,
                kMaxVirt);
      return PO_ERR_ARG;
    }
    vc.count = nvirt;
    for (int j = 0; j < nvirt; j++) vc.s[j] = vs[j];
  }
  if (t2 != nullptr && ca0 != 0) {
    set_error("k_solve2r: the stored right-hand side form expects the constraint columns first");
    return PO_ERR_ARG;
  }
  if (t2 == nullptr) {
    if (!ar || !rx) {
      set_error("k_solve2r: neither the refinement right-hand side nor the data to recompute it");
      return PO_ERR_ARG;
    }
    if (g && merit_out) {
      PO_LAUNCH((solve2r_kernel<1, 1>), grid, b, t1, t2, dinv, ct1, ct2, ctr, pt, nv, beta_mu, tau, rx, diag, n, px,
                pzl, pzu, va, nca, ca0, vc, g, dinv_diag, gcv, gc1, gc2, c->d_partials);
      // {7 sums, 2 minima, 1 maximum} land in merit_out[0..10); the minima are copied to `out` by the caller's hook
      (void)out;
      return reduce_finish(c, grid, 7, 2, 1, merit_out);
    }
    PO_LAUNCH((solve2r_kernel<1, 0>), grid, b, t1, t2, dinv, ct1, ct2, ctr, pt, nv, beta_mu, tau, rx, diag, n, px, pzl,
              pzu, va, nca, ca0, vc, g, dinv_diag, gcv, gc1, gc2, c->d_partials);
  } else if (g && merit_out) {
    PO_LAUNCH((solve2r_kernel<0, 1>), grid, b, t1, t2, dinv, ct1, ct2, ctr, pt, nv, beta_mu, tau, rx, diag, n, px, pzl,
              pzu, va, nca, ca0, vc, g, dinv_diag, gcv, gc1, gc2, c->d_partials);
    (void)out;
    return reduce_finish(c, grid, 7, 2, 1, merit_out);
  } else {
    PO_LAUNCH((solve2r_kernel<0, 0>), grid, b, t1, t2, dinv, ct1, ct2, ctr, pt, nv, beta_mu, tau, rx, diag, n, px, pzl,
              pzu, va, nca, ca0, vc, g, dinv_diag, gcv, gc1, gc2, c->d_partials);
  }
  return reduce_finish(c, grid, 0, 2, 0, out);
}

int k_solve2(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
             const double *const *P, int nv, double beta_mu, int refine, double tau, int64_t n,
             double *px, double *pzl, double *pzu, double out[2], const double *coef2,
             const double *rx, double diag, double *tout, double *va, int nca, const double *cl,
             const double *cu) {
  count_bytes(c, nv + 10 + (refine ? 3 : 0) + (va ? (refine ? 2 : 1) : 0) + (coef2 ? 2 : 0) + (cl ? 2 : 0), n);
  if (nv > kMaxPanel) {
    if (coef2) {
      set_error("k_solve2: the fused refinement residual is not available for a panel of %d (> %d) columns", nv,
                kMaxPanel);
      return PO_ERR_ARG;
    }
    const double *P2[2] = {nullptr, nullptr};
    double a2[2] = {0.0, 0.0};
    int nv2 = 0, nca2 = 0;
    PO_TRY(collapse_panel(c, alpha, P, nv, va ? nca : 0, n, P2, a2, &nv2, &nca2));
    return k_solve2(c, b, t, dinv, a2, P2, nv2, beta_mu, refine, tau, n, px, pzl, pzu, out, nullptr, rx, diag, tout, va,
                    nca2, cl, cu);
  }
  const int grid = grid_for(c, n, kBpcPanel);
  PO_TRY(ensure_partials(c, (size_t)grid * 2));
  PtrTable pt;
  CoefTable ct, ct2;
  fill_tables(alpha, P, nv, &ct, &pt);
  fill_tables(coef2, P, nv, &ct2, &pt);
  if (refine) {
    PO_LAUNCH((solve2_kernel<1, 0>), grid, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, px,
              pzl, pzu, tout, va, nca, cl, cu, c->d_partials);
  } else if (coef2) {
    PO_LAUNCH((solve2_kernel<0, 1>), grid, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, px,
              pzl, pzu, tout, va, nca, cl, cu, c->d_partials);
  } else {
    PO_LAUNCH((solve2_kernel<0, 0>), grid, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, px,
              pzl, pzu, tout, va, nca, cl, cu, c->d_partials);
  }
  return reduce_finish(c, grid, 0, 2, 0, out);
}

// -------------------------------------------------------------------------------------------------
// First solve pass with the refinement residual AND its panel dots fused (one pass over P instead
// of three: solve2 + res_step + mdot).  Mapping: a workgroup owns tiles of 128 rows; wave w owns the
// panel columns j = w (mod 4), lane l the rows 2l, 2l+1 of the tile.  Per tile:
//   (1) every wave forms the partial row sums  sum_j alpha_j P_j  over its own columns straight from the
//       prefetch registers, posts them (6 KB of LDS) and parks its column slice of the tile in LDS;
//   (2) the prefetch registers are free again: the loads of the NEXT tile are issued now and stay in flight
//       through the rest of the tile;
//   (3) wave 0 runs the element epilogue (px, pzl, pzu, t' = Dinv*d1') on operands it prefetched one tile
//       ahead, broadcasts t' through 1 KB of LDS and prefetches its operands of the next tile;
//   (4) every wave accumulates  P_j^T t'  for its own columns from its LDS slice.
// Each wave reads back only the LDS columns it wrote itself, so two barriers per tile suffice.
// -------------------------------------------------------------------------------------------------
constexpr int kS2Tile = 128;
template <int NPASS, int OCC, int VIRT>
__global__ void __launch_bounds__(kBlock, OCC)
    solve2_dots_kernel(Bounds b, const double *t, const double *__restrict__ dinv, CoefTable alpha,
                       CoefTable coef2, PtrTable P, int nv, double beta_mu, double tau,
                       const double *__restrict__ rx, double diag, int64_t n, int64_t ntiles,
                       double *__restrict__ px, double *__restrict__ pzl, double *__restrict__ pzu,
                       double *tout, double *__restrict__ va, int nca, double *__restrict__ traw,
                       int store_step, int ca0, VirtCols vc, double dinv_diag, GroupCols2 gcs,
                       double *__restrict__ partials) {
  extern __shared__ double s2lds[];  // [4][NPASS][128] column slices, [4*64*6] partial sums, [128] t', [8]
  double *pt = s2lds;
  double *sacc = s2lds + 4 * NPASS * kS2Tile;
  double *stp = sacc + 4 * 64 * 6;
  double *sm = stp + kS2Tile;
  const int tid = threadIdx.x, lane = tid & 63;
  // wave-uniform on purpose: keeps the column pointers and coefficients in scalar registers
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  double *myslice = pt + (size_t)wave * NPASS * kS2Tile + 2 * lane;
  double dotacc[NPASS];
#pragma unroll
  for (int it = 0; it < NPASS; it++) dotacc[it] = 0.0;
  double mins[2] = {1.0, 1.0};
  const int64_t qlast = (n - 1) >> 1;
  f64x2 buf[NPASS];
  f64x2 sbuf[VIRT ? 3 : 1];  // S partners of this wave's unformed L-SR1 columns (they lead the panel: j < 12)
  // the grouped column this wave adds behind its own columns (if any), prefetched with them
  const int gmine = (gcs.count > 0 && wave == (nv & 3)) ? 0 : ((gcs.count > 1 && wave == ((nv + 1) & 3)) ? 1 : -1);
  GRaw gbuf;
  gbuf.raw = (f64x2){0.0, 0.0};
  gbuf.valid = 0;
  // The element epilogue is run by waves 0 and 1 with ONE element per lane (rows 64 w + lane of the tile): half the
  // dependent fp64 divisions per wave of the former one-wave, two-elements-per-lane form, whose chain bounded the pass.
  // Element arithmetic is per element, so which lane runs it changes no bit.  eb: x, lb, ub, zl, zu, t, dinv, rx of it.
  double eb[8];
  int64_t ie = 0;   // the element (global row) this lane finishes, clamped into range for the loads
  bool ein = false;  // ... and whether it exists (ie_raw < n)
  int64_t q = 0;
  bool in = false;
#define PO_S2_PREFETCH(TILE)                                                                 \
  {                                                                                          \
    q = (TILE) * 64 + lane;                                                                  \
    in = (2 * q < n);                                                                        \
    if (!in) q = qlast;                                                                      \
    _Pragma("unroll") for (int it = 0; it < NPASS; it++) {                                   \
      const int j = wave + 4 * it;                                                           \
      buf[it] = ld_stream(P.p[j < nv ? j : 0] + 2 * q);                                      \
    }                                                                                        \
    if (VIRT) {                                                                              \
      _Pragma("unroll") for (int it = 0; it < (VIRT ? 3 : 0); it++) {                        \
        const int j = wave + 4 * it;                                                         \
        sbuf[it] = ld_stream(vc.s[j < vc.count ? j : 0] + 2 * q);                            \
      }                                                                                      \
    }                                                                                        \
    if (gmine >= 0) gbuf = gcol_request(gcs.g[gmine], q, n);                                 \
  }
#define PO_S2_PREFETCH_E(TILE)                                                               \
  {                                                                                          \
    const int64_t _ir = (TILE) * kS2Tile + 64 * wave + lane;                                 \
    ein = _ir < n;                                                                           \
    ie = ein ? _ir : n - 1;                                                                  \
    eb[0] = b.x[ie];                                                                         \
    eb[1] = b.lb[ie];                                                                        \
    eb[2] = b.ub[ie];                                                                        \
    eb[3] = b.zl[ie];                                                                        \
    eb[4] = b.zu[ie];                                                                        \
    if (t) { /* (nullptr: t and Dinv re-formed from the bound data and rx, as dinv_d1_kernel formed them) */ \
      eb[5] = t[ie];                                                                         \
      eb[6] = dinv[ie];                                                                      \
    }                                                                                        \
    eb[7] = rx[ie];                                                                          \
  }
  if ((int64_t)blockIdx.x < ntiles) {
    PO_S2_PREFETCH((int64_t)blockIdx.x);
    if (wave < 2) PO_S2_PREFETCH_E((int64_t)blockIdx.x);
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t qc = q;  // row pair of this lane in the current tile
    const bool inc = in;
    f64x2 a1 = (f64x2){0.0, 0.0}, a2 = a1, aA = a1;
#pragma unroll
    for (int it = 0; it < NPASS; it++) {
      const int j = wave + 4 * it;  // coefficient tables are zero beyond nv
      f64x2 v = buf[it];
      if (VIRT && it < 3 && j < vc.count) {  // Z_j = Y_j - b0 S_j in registers
        v.x -= vc.b0 * sbuf[VIRT ? it : 0].x;
        v.y -= vc.b0 * sbuf[VIRT ? it : 0].y;
      }
      if (!inc) v = (f64x2){0.0, 0.0};
      const double ca = alpha.a[j], cb = coef2.a[j], cA = (j >= ca0 && j < ca0 + nca) ? ca : 0.0;
      a1 += ca * v;
      a2 += cb * v;
      aA += cA * v;
      *reinterpret_cast<f64x2 *>(myslice + it * kS2Tile) = v;
    }
    // grouped columns (core.hpp: GroupCol) at panel positions nv, nv + 1: the wave that would have held them as
    // stored columns adds them behind its own columns (same order of additions)
    if (gmine >= 0) {
      f64x2 v = gcol_value(gcs.g[gmine], gbuf);
      if (!inc) v = (f64x2){0.0, 0.0};
      a1 += gcs.ca[gmine] * v;
      a2 += gcs.cb[gmine] * v;
    }
    // [value][wave][lane]: consecutive lanes hit consecutive 8-byte slots (the [lane][value] layout of round 2's
    // first version cost 21 % LDS bank-conflict cycles, profiles/r02_pmc_counters.json)
    double *sa = sacc + wave * 64 + lane;
    sa[0 * 256] = a1.x;
    sa[1 * 256] = a1.y;
    sa[2 * 256] = a2.x;
    sa[3 * 256] = a2.y;
    sa[4 * 256] = aA.x;
    sa[5 * 256] = aA.y;
    __syncthreads();
    const bool more = tile + gridDim.x < ntiles;
    // waves 0 and 1 issue their share of the next tile after the epilogue (keeps the prefetch registers out of the
    // epilogue's register peak)
    if (more && wave >= 2) PO_S2_PREFETCH(tile + gridDim.x);
    if (wave < 2) {
      // this lane's element: row 64 wave + lane of the tile = component (lane & 1) of row pair 32 wave + lane / 2
      const int er = 64 * wave + lane, el = er >> 1, ec = er & 1;
      double acc = 0.0, acc2 = 0.0, accA = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const double *sp = sacc + w * 64 + el;
        acc += sp[(0 + ec) * 256];
        acc2 += sp[(2 + ec) * 256];
        accA += sp[(4 + ec) * 256];
      }
      double tp = 0.0;
      const int64_t i = tile * kS2Tile + er;  // (ie, clamped, is what the operands were loaded from)
      if (ein) {
        const double _x = eb[0], _lb = eb[1], _ub = eb[2], _zl = eb[3], _zu = eb[4];
        const BE e = bound_elem(_x, _lb, _ub, _zl, _zu, b.max_bound, b.use_lower, b.use_upper);
        const double r = eb[7];
        double tv, dv;
        if (t) {
          tv = eb[5];
          dv = eb[6];
        } else {
          dv = dinv_elem(e, dinv_diag);
          tv = dv * d1_elem(e, r, beta_mu);
        }
        if (va && store_step == 1) va[i] = accA;
        const Step3 s0 = solve2_elem<0>(e, tv + dv * acc, beta_mu, 0.0, 0.0, 0.0);
        // store_step == 0: the refinement pass recomputes this first step from t and alpha in registers
        // (solve2r_kernel) -- an HBM write costs about four reads on this part (tools/layout_probe.hip)
        // store_step == 2: px only -- the bound-multiplier steps are functions of it and of data every later pass
        // loads anyway (sparse-constraint path, whose refinement residual needs Aw px as a vector)
        if (store_step) {
          px[i] = s0.px;
          if (store_step == 1) {
            pzl[i] = s0.pzl;
            pzu[i] = s0.pzu;
          }
        }
        // raw d1' and t' = Dinv o d1' (the product res_step_elem would form itself)
        const double raw = res_step_elem(e, r, acc2, diag, s0.px, s0.pzl, s0.pzu, 1.0, beta_mu, b.use_lower,
                                         b.use_upper);
        tp = dv * raw;
        if (traw) {  // the caller applies its own block solve to the raw right-hand side (sparse constraints)
          traw[i] = raw;
        } else if (tout) {  // (nullptr: the refinement pass recomputes t' too -- this pass only takes its products)
          tout[i] = tp;
        }
        max_step_elem(b, _x, _lb, _ub, _zl, _zu, s0, tau, mins[0], mins[1]);
      } else if (i == n && (n & 1)) {
        // the pad element of an odd length stays 0.0 in every output (what the paired stores of the other passes keep)
        if (va && store_step == 1) va[i] = 0.0;
        if (store_step) {
          px[i] = 0.0;
          if (store_step == 1) {
            pzl[i] = 0.0;
            pzu[i] = 0.0;
          }
        }
        if (traw) {
          traw[i] = 0.0;
        } else if (tout) {
          tout[i] = 0.0;
        }
      }
      stp[er] = tp;
      if (more) {
        PO_S2_PREFETCH(tile + gridDim.x);
        PO_S2_PREFETCH_E(tile + gridDim.x);
      }
    }
    __syncthreads();
    const f64x2 tt = *reinterpret_cast<const f64x2 *>(stp + 2 * lane);
#pragma unroll
    for (int it = 0; it < NPASS; it++) {
      const f64x2 v = *reinterpret_cast<const f64x2 *>(myslice + it * kS2Tile);
      dotacc[it] = fma(v.x, tt.x, fma(v.y, tt.y, dotacc[it]));
    }
  }
#undef PO_S2_PREFETCH
#undef PO_S2_PREFETCH_E
  // dots: wave w holds columns w + 4*it; slots 0..nv-1 (sums), then the two minima
#pragma unroll
  for (int it = 0; it < NPASS; it++) {
    const double v = wave_reduce<OP_SUM>(dotacc[it]);
    const int j = wave + 4 * it;
    if (lane == 0 && j < nv) partials[(size_t)j * gridDim.x + blockIdx.x] = v;
  }
  __syncthreads();
  block_reduce_store<2, OP_MIN>(mins, partials, nv, sm);
}

// -------------------------------------------------------------------------------------------------
// TWO TILES PER STEP (round 5; narrow panels, NPASS <= 6, no unformed columns).  What bounds the form above on a narrow
// panel is the instruction stream of the element epilogue -- 13 IEEE divisions per element -- which two of the four
// wavefronts run while the other two wait (HISTORY R5.11).  Here a workgroup takes its next TWO tiles (the same tiles
// in the same order as above: blockIdx.x + k gridDim.x, k = 2 m and 2 m + 1) at once: waves 0-1 finish the elements of
// the first, waves 2-3 those of the second, at the same time.  The columns stay in registers (no LDS slices: the request
// for the next pair goes out behind the dots, and the other workgroups of the CU cover its latency -- R5.11 measured that
// they do), every wave takes the dots of its own columns with the first tile's t' and THEN with the second's: the order
// in which the form above adds them, so every sum keeps its bits.
// -------------------------------------------------------------------------------------------------
template <int NPASS, int OCC>
__global__ void __launch_bounds__(kBlock, OCC)
    solve2_dots2_kernel(Bounds b, const double *t, const double *__restrict__ dinv, CoefTable alpha,
                        CoefTable coef2, PtrTable P, int nv, double beta_mu, double tau,
                        const double *__restrict__ rx, double diag, int64_t n, int64_t ntiles,
                        double *__restrict__ px, double *__restrict__ pzl, double *__restrict__ pzu,
                        double *tout, double *__restrict__ va, int nca, double *__restrict__ traw,
                        int store_step, int ca0, double dinv_diag, GroupCols2 gcs,
                        double *__restrict__ partials) {
  extern __shared__ double s2lds[];  // [2][4*64*6] partial sums, [2][128] t', [8]
  double *sacc = s2lds;
  double *stp = s2lds + 2 * (4 * 64 * 6);
  double *sm = stp + 2 * kS2Tile;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ex = wave >> 1, eh = wave & 1;  // epilogue: rows 64 eh + lane of tile `ex` of the pair
  double dotacc[NPASS];
#pragma unroll
  for (int it = 0; it < NPASS; it++) dotacc[it] = 0.0;
  double mins[2] = {1.0, 1.0};
  const int64_t qlast = (n - 1) >> 1;
  f64x2 buf[2][NPASS];
  const int gmine = (gcs.count > 0 && wave == (nv & 3)) ? 0 : ((gcs.count > 1 && wave == ((nv + 1) & 3)) ? 1 : -1);
  GRaw gbuf[2];
  bool in[2] = {false, false};
  double eb[8];
  bool ein = false;
#pragma unroll
  for (int x = 0; x < 2; x++) {
    gbuf[x].raw = (f64x2){0.0, 0.0};
    gbuf[x].valid = 0;
  }
  // columns of both tiles of the pair that starts at FIRST, and this wave's element operands of its own tile
#define PO_S22_REQUEST(FIRST)                                                                \
  {                                                                                          \
    _Pragma("unroll") for (int x = 0; x < 2; x++) {                                          \
      const int64_t _tile = (FIRST) + x * (int64_t)gridDim.x;                                \
      int64_t _q = _tile * 64 + lane;                                                        \
      in[x] = _tile < ntiles && 2 * _q < n;                                                  \
      if (!in[x]) _q = qlast;                                                                \
      _Pragma("unroll") for (int it = 0; it < NPASS; it++) {                                 \
        const int j = wave + 4 * it;                                                         \
        buf[x][it] = ld_stream(P.p[j < nv ? j : 0] + 2 * _q);                                \
      }                                                                                      \
      if (gmine >= 0) gbuf[x] = gcol_request(gcs.g[gmine], _q, n);                           \
    }                                                                                        \
    const int64_t _mine = (FIRST) + ex * (int64_t)gridDim.x;                                 \
    const int64_t _ir = _mine * kS2Tile + 64 * eh + lane;                                    \
    ein = _mine < ntiles && _ir < n;                                                         \
    const int64_t _ie = ein ? _ir : n - 1;                                                   \
    eb[0] = b.x[_ie];                                                                        \
    eb[1] = b.lb[_ie];                                                                       \
    eb[2] = b.ub[_ie];                                                                       \
    eb[3] = b.zl[_ie];                                                                       \
    eb[4] = b.zu[_ie];                                                                       \
    if (t) {                                                                                 \
      eb[5] = t[_ie];                                                                        \
      eb[6] = dinv[_ie];                                                                     \
    }                                                                                        \
    eb[7] = rx[_ie];                                                                         \
  }
  if ((int64_t)blockIdx.x < ntiles) PO_S22_REQUEST((int64_t)blockIdx.x);
  for (int64_t first = blockIdx.x; first < ntiles; first += 2 * (int64_t)gridDim.x) {
    const bool has2 = first + gridDim.x < ntiles;  // (workgroup-uniform)
#pragma unroll
    for (int x = 0; x < 2; x++) {
      f64x2 a1 = (f64x2){0.0, 0.0}, a2 = a1, aA = a1;
#pragma unroll
      for (int it = 0; it < NPASS; it++) {
        const int j = wave + 4 * it;  // coefficient tables are zero beyond nv
        f64x2 v = buf[x][it];
        if (!in[x]) v = (f64x2){0.0, 0.0};
        buf[x][it] = v;  // (what the form above parks in LDS)
        const double ca = alpha.a[j], cb = coef2.a[j], cA = (j >= ca0 && j < ca0 + nca) ? ca : 0.0;
        a1 += ca * v;
        a2 += cb * v;
        aA += cA * v;
      }
      if (gmine >= 0) {
        f64x2 v = gcol_value(gcs.g[gmine], gbuf[x]);
        if (!in[x]) v = (f64x2){0.0, 0.0};
        a1 += gcs.ca[gmine] * v;
        a2 += gcs.cb[gmine] * v;
      }
      double *sa = sacc + x * (4 * 64 * 6) + wave * 64 + lane;
      sa[0 * 256] = a1.x;
      sa[1 * 256] = a1.y;
      sa[2 * 256] = a2.x;
      sa[3 * 256] = a2.y;
      sa[4 * 256] = aA.x;
      sa[5 * 256] = aA.y;
    }
    __syncthreads();
    if (ex == 0 || has2) {
      const int64_t tile = first + ex * (int64_t)gridDim.x;
      const int er = 64 * eh + lane, el = er >> 1, ec = er & 1;
      double acc = 0.0, acc2 = 0.0, accA = 0.0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const double *sp = sacc + ex * (4 * 64 * 6) + w * 64 + el;
        acc += sp[(0 + ec) * 256];
        acc2 += sp[(2 + ec) * 256];
        accA += sp[(4 + ec) * 256];
      }
      double tp = 0.0;
      const int64_t i = tile * kS2Tile + er;
      if (ein) {
        const double _x = eb[0], _lb = eb[1], _ub = eb[2], _zl = eb[3], _zu = eb[4];
        const BE e = bound_elem(_x, _lb, _ub, _zl, _zu, b.max_bound, b.use_lower, b.use_upper);
        const double r = eb[7];
        double tv, dv;
        if (t) {
          tv = eb[5];
          dv = eb[6];
        } else {
          dv = dinv_elem(e, dinv_diag);
          tv = dv * d1_elem(e, r, beta_mu);
        }
        if (va && store_step == 1) va[i] = accA;
        const Step3 s0 = solve2_elem<0>(e, tv + dv * acc, beta_mu, 0.0, 0.0, 0.0);
        if (store_step) {
          px[i] = s0.px;
          if (store_step == 1) {
            pzl[i] = s0.pzl;
            pzu[i] = s0.pzu;
          }
        }
        const double raw = res_step_elem(e, r, acc2, diag, s0.px, s0.pzl, s0.pzu, 1.0, beta_mu, b.use_lower,
                                         b.use_upper);
        tp = dv * raw;
        if (traw) {
          traw[i] = raw;
        } else if (tout) {
          tout[i] = tp;
        }
        max_step_elem(b, _x, _lb, _ub, _zl, _zu, s0, tau, mins[0], mins[1]);
      } else if (i == n && (n & 1)) {
        if (va && store_step == 1) va[i] = 0.0;
        if (store_step) {
          px[i] = 0.0;
          if (store_step == 1) {
            pzl[i] = 0.0;
            pzu[i] = 0.0;
          }
        }
        if (traw) {
          traw[i] = 0.0;
        } else if (tout) {
          tout[i] = 0.0;
        }
      }
      stp[ex * kS2Tile + er] = tp;
    }
    __syncthreads();
    {
      const f64x2 tt = *reinterpret_cast<const f64x2 *>(stp + 2 * lane);
#pragma unroll
      for (int it = 0; it < NPASS; it++) dotacc[it] = fma(buf[0][it].x, tt.x, fma(buf[0][it].y, tt.y, dotacc[it]));
    }
    if (has2) {
      const f64x2 tt = *reinterpret_cast<const f64x2 *>(stp + kS2Tile + 2 * lane);
#pragma unroll
      for (int it = 0; it < NPASS; it++) dotacc[it] = fma(buf[1][it].x, tt.x, fma(buf[1][it].y, tt.y, dotacc[it]));
    }
    if (first + 2 * (int64_t)gridDim.x < ntiles) PO_S22_REQUEST(first + 2 * (int64_t)gridDim.x);
  }
#undef PO_S22_REQUEST
#pragma unroll
  for (int it = 0; it < NPASS; it++) {
    const double v = wave_reduce<OP_SUM>(dotacc[it]);
    const int j = wave + 4 * it;
    if (lane == 0 && j < nv) partials[(size_t)j * gridDim.x + blockIdx.x] = v;
  }
  __syncthreads();
  block_reduce_store<2, OP_MIN>(mins, partials, nv, sm);
}

template <int NP, int OCC>
static int solve2_dots2_launch(Ctx *c, const Bounds &b, const double *t, const double *dinv, const CoefTable &ct,
                               const CoefTable &ct2, const PtrTable &pt, int nv, double beta_mu, double tau,
                               const double *rx, double diag, int64_t n, int64_t ntiles, double *px, double *pzl,
                               double *pzu, double *tout, double *va, int nca, double *traw, int store_step, int ca0,
                               double dinv_diag, int *grid_out, const GroupCols2 &gcs) {
  const size_t lds = sizeof(double) * (size_t)(2 * (4 * 64 * 6) + 2 * kS2Tile + 8);
  // the grid of the one-tile form (same tiles per workgroup, same partial sums)
  const size_t lds1 = sizeof(double) * (size_t)(4 * NP * kS2Tile + 4 * 64 * 6 + kS2Tile + 8);
  int per_cu = (int)((160 * 1024) / lds1);
  if (per_cu > OCC) per_cu = OCC;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)c->num_cu * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  PO_TRY(ensure_partials(c, (size_t)g * (nv + 2)));
  hipLaunchKernelGGL((solve2_dots2_kernel<NP, OCC>), dim3((int)g), dim3(kBlock), lds, c->stream, b, t, dinv, ct, ct2, pt,
                     nv, beta_mu, tau, rx, diag, n, ntiles, px, pzl, pzu, tout, va, nca, traw, store_step, ca0,
                     dinv_diag, gcs, c->d_partials);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  *grid_out = (int)g;
  return PO_OK;
}

template <int NP, int OCC, int VIRT>
static int solve2_dots_launch(Ctx *c, int grid_cap, const Bounds &b, const double *t, const double *dinv,
                              const CoefTable &ct, const CoefTable &ct2, const PtrTable &pt, int nv, double beta_mu,
                              double tau, const double *rx, double diag, int64_t n, int64_t ntiles, double *px,
                              double *pzl, double *pzu, double *tout, double *va, int nca, double *traw,
                              int store_step, int ca0, const VirtCols &vc, double dinv_diag, int *grid_out,
                              const GroupCols2 &gcs) {
  const size_t lds = sizeof(double) * (size_t)(4 * NP * kS2Tile + 4 * 64 * 6 + kS2Tile + 8);
  static bool attr_set = false;
  if (!attr_set) {
    PO_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(solve2_dots_kernel<NP, OCC, VIRT>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr_set = true;
  }
  int per_cu = (int)((160 * 1024) / lds);
  if (per_cu > OCC) per_cu = OCC;
  if (per_cu < 1) per_cu = 1;
  int64_t g = (int64_t)c->num_cu * per_cu;
  if (g > ntiles) g = ntiles;
  if (g < 1) g = 1;
  PO_TRY(ensure_partials(c, (size_t)g * (nv + 2)));
  hipLaunchKernelGGL((solve2_dots_kernel<NP, OCC, VIRT>), dim3((int)g), dim3(kBlock), lds, c->stream, b, t, dinv, ct, ct2, pt, nv,
                     beta_mu, tau, rx, diag, n, ntiles, px, pzl, pzu, tout, va, nca, traw, store_step, ca0, vc,
                     dinv_diag, gcs, c->d_partials);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  *grid_out = (int)g;
  return PO_OK;
}

// (OD / OA: workgroups per CU, default and alternative (PAROPT_AMD_S2D_OCC).  Until the element epilogue moved to two
// waves with one element per lane the stored-step form of narrow panels -- the sparse-constraint path -- took 2 by
// default (its epilogue kept the step and the raw right-hand side live beside the prefetched operands and spilled at
// 3); with 129-143 registers it takes 3 like the others: 177.8 -> 181.3 it/s at config 4 in one call.  Wider panels
// stay at 2: 3 spills there and costs 1 % at config 3, 7 % at config 2.)
#define PO_S2D_CASE(NP)                                                                                    \
  case NP: {                                                                                               \
    /* (round 5: the alternatives of the wide panels -- 3 workgroups per CU at 11 / 12 slots, 2 at 20 / 24 --       \
       spilled to scratch and are gone: every instantiation left is scratch-free, tests/test_kernel_resources.py) */ \
    constexpr int OD = NP <= 8 ? 3 : (NP <= 16 ? 2 : 1), OA = NP <= 8 ? 2 : OD;                            \
    constexpr int OV = NP <= 16 ? 2 : 1; /* with unformed columns: three more prefetch registers */        \
    /* (two tiles per step up to 6 slots per wave: 8 slots would need 168 + 13 registers at three workgroups per CU) */ \
    if (NP <= 6 && vc.count == 0 && two_tiles && occ_env != OA) {                                          \
      constexpr int NP2 = NP <= 6 ? NP : 6;                                                                \
      PO_TRY((solve2_dots2_launch<NP2, 3>(c, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, ntiles, px, pzl, \
                                          pzu, tout, va, nca, traw, store_step, ca0, dinv_diag, &grid, gcs))); \
    } else if (vc.count > 0)                                                                               \
      PO_TRY((solve2_dots_launch<NP, OV, 1>(c, 0, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, ntiles, \
                                            px, pzl, pzu, tout, va, nca, traw, store_step, ca0, vc, dinv_diag, &grid, gcs))); \
    else if (occ_env == OA)                                                                                \
      PO_TRY((solve2_dots_launch<NP, OA, 0>(c, 0, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, ntiles, \
                                            px, pzl, pzu, tout, va, nca, traw, store_step, ca0, vc, dinv_diag, &grid, gcs))); \
    else                                                                                                   \
      PO_TRY((solve2_dots_launch<NP, OD, 0>(c, 0, b, t, dinv, ct, ct2, pt, nv, beta_mu, tau, rx, diag, n, ntiles, \
                                            px, pzl, pzu, tout, va, nca, traw, store_step, ca0, vc, dinv_diag, &grid, gcs))); \
  } break;

// out = {dots[nv] = P^T t', max_x, max_z}
int k_solve2_dots(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
                  const double *coef2, const double *const *P, int nv, double beta_mu, double tau,
                  const double *rx, double diag, int64_t n, double *px, double *pzl, double *pzu,
                  double *tout, double *va, int nca, double *out, double *traw, int store_step, int ca0,
                  const double *const *vs, int nvirt, double b0v, double dinv_diag, const GroupCols2 *gcols) {
  count_bytes(c, nv + nvirt + 6 + (t ? 2 : 0) + (store_step == 1 ? 3 + (va ? 1 : 0) : (store_step == 2 ? 1 : 0)) + ((traw || tout) ? 1 : 0), n);
  const GroupCols2 gcs = gcols ? *gcols : GroupCols2();
  for (int e = 0; e < gcs.count; e++) PO_TRY(gcol_check(&gcs.g[e], n, "k_solve2_dots"));
  for (int e = 0; e < gcs.count; e++) count_bytes(c, 1.0, gcs.g[e].nwcon);
  if (nv > kMaxPanel || nv < 1) {
    set_error("panel of %d vectors outside 1..%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  VirtCols vc;
  vc.count = 0;
  vc.b0 = b0v;
  for (int j = 0; j < kMaxVirt; j++) vc.s[j] = nullptr;
  if (nvirt > 0) {
    if (nvirt > kMaxVirt || ca0 < nvirt) {
      set_error("k_solve2_dots: %d unformed columns (at most %d, leading the panel)", nvirt, kMaxVirt);
      return PO_ERR_ARG;
    }
    vc.count = nvirt;
    for (int j = 0; j < nvirt; j++) vc.s[j] = vs[j];
  }
  const int64_t ntiles = (((n + 1) >> 1) + 63) / 64;
  int grid = 0;
  static const int occ_env = getenv("PAROPT_AMD_S2D_OCC") ? atoi(getenv("PAROPT_AMD_S2D_OCC")) : 0;
  // PAROPT_AMD_S2D_TWO=1: narrow panels two tiles per step (solve2_dots2_kernel; same bits either way).  OFF by
  // default: measured in one call (profiles/r05_ab_solve2_dots_two_tiles.jsonl) the kernels are 9-15 % faster at
  // n = 5 M (config 5: 93.5 / 137.9 / 174.5 us against 109 / 150.9 / 194.9) and 5-7 % SLOWER at n = 20 M (config 4:
  // 1053 against 1006 us), and neither line moves (782 / 794 against 791 / 783 inner it/s, 178.0 against 178.0 it/s).
  const bool two_tiles = dbg_switch(SW_S2D_TWO, "PAROPT_AMD_S2D_TWO", 0) != 0;
  PtrTable pt;
  CoefTable ct, ct2;
  fill_tables(alpha, P, nv, &ct, &pt);
  fill_tables(coef2, P, nv, &ct2, &pt);
  const int need = (nv + 3) / 4;
  // (column slots per wave: a slot beyond the panel re-reads column 0 with a zero coefficient -- no HBM traffic, but a
  // load per tile all the same: 2 and 6 slots for the narrow panels of configs 5 and 4 instead of 4 and 8; same
  // workgroups per CU, so the same grid and the same bits)
  const int np = need <= 2 ? 2 : need <= 4 ? 4 : need <= 6 ? 6 : need <= 8 ? 8 : need <= 11 ? 11 : need <= 12 ? 12
                 : need <= 16 ? 16 : need <= 20 ? 20 : 24;
  switch (np) {
    PO_S2D_CASE(2) PO_S2D_CASE(4) PO_S2D_CASE(6) PO_S2D_CASE(8) PO_S2D_CASE(11) PO_S2D_CASE(12) PO_S2D_CASE(16)
    PO_S2D_CASE(20) PO_S2D_CASE(24)
  }
  return reduce_finish(c, grid, nv, 2, 0, out);
}

// stand-alone refinement residual (second and later refinement steps, or when the fused form of
// solve2_kernel is not applicable)
__global__ void __launch_bounds__(kBlock)
    res_step_kernel(Bounds b, const double *__restrict__ rx, const double *__restrict__ px,
                    const double *__restrict__ pzl, const double *__restrict__ pzu,
                    const double *__restrict__ dinv, CoefTable coef, PtrTable P, int nv, double diag,
                    double beta_mu, int64_t n, double *__restrict__ tp) {
  PO_PAIR_LOOP(q, n) {
    const double2 acc = panel_sum(P, coef, nv, q);
    PO_LOAD_BOUNDS(b, q, n);
    const double2 r = ld2(rx, q, n), p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    const double2 dv = dinv ? ld2(dinv, q, n) : make_double2(1.0, 1.0);  // null: raw d1'
    st2(tp, q, n,
        make_double2(res_step_elem(e0, r.x, acc.x, diag, p.x, l.x, u.x, dv.x, beta_mu, b.use_lower,
                                   b.use_upper),
                     res_step_elem(e1, r.y, acc.y, diag, p.y, l.y, u.y, dv.y, beta_mu, b.use_lower,
                                   b.use_upper)));
  }
}
// checkKKTStep (src/ParOptInteriorPoint.cpp:6212-6360): maxima of the three n-sized blocks of the residual of the
// linearised KKT system at the computed step: r'x = rx - diag px + sum coef_j P_j + [L]pzl - [U]pzu,
// r'zl = -((x-lb) zl - beta mu) - ((x-lb) pzl + px zl), r'zu likewise
__global__ void __launch_bounds__(kBlock)
    step_check_kernel(Bounds b, const double *__restrict__ rx, const double *__restrict__ px,
                      const double *__restrict__ pzl, const double *__restrict__ pzu, CoefTable coef, PtrTable P,
                      int nv, double diag, double beta_mu, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 3];
  double mx[3] = {0.0, 0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    const double2 acc = panel_sum(P, coef, nv, q);
    PO_LOAD_BOUNDS(b, q, n);
    const double2 r = ld2(rx, q, n), p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    const BE es[2] = {e0, e1};
    const double rr[2] = {r.x, r.y}, pp[2] = {p.x, p.y}, ll[2] = {l.x, l.y}, uu[2] = {u.x, u.y},
                 aa[2] = {acc.x, acc.y};
#pragma unroll
    for (int k = 0; k < 2; k++) {  // (a compile-time trip count keeps the element arrays in registers)
      if (k == 1 && !_has2) break;
      double v = rr[k] - diag * pp[k] + aa[k];
      if (b.use_lower) v += ll[k];
      if (b.use_upper) v -= uu[k];
      mx[0] = fmax(mx[0], fabs(v));
      if (es[k].L) mx[1] = fmax(mx[1], fabs(-(es[k].xl * es[k].zl - beta_mu) - (es[k].xl * ll[k] + pp[k] * es[k].zl)));
      if (es[k].U) mx[2] = fmax(mx[2], fabs(-(es[k].xu * es[k].zu - beta_mu) - (es[k].xu * uu[k] - pp[k] * es[k].zu)));
    }
  }
  block_reduce_store<3, OP_MAX>(mx, partials, 0, sm);
}
int k_step_check(Ctx *c, const Bounds &b, const double *rx, const double *px, const double *pzl, const double *pzu,
                 const double *coef, const double *const *P, int nv, double diag, double beta_mu, int64_t n,
                 double out[3]) {
  if (nv > kMaxPanel) {
    const double *w = nullptr, one = 1.0;
    PO_TRY(collapse_range(c, 0, coef, P, 0, nv, n, &w));
    return k_step_check(c, b, rx, px, pzl, pzu, &one, &w, 1, diag, beta_mu, n, out);
  }
  count_bytes(c, nv + 9, n);
  const int grid = grid_for(c, n, kBpcPanel);
  PO_TRY(ensure_partials(c, (size_t)grid * 3));
  PtrTable pt;
  CoefTable ct;
  fill_tables(coef, P, nv, &ct, &pt);
  PO_LAUNCH(step_check_kernel, grid, b, rx, px, pzl, pzu, ct, pt, nv, diag, beta_mu, n, c->d_partials);
  return reduce_finish(c, grid, 0, 0, 3, out);
}

int k_res_step(Ctx *c, const Bounds &b, const double *rx, const double *px, const double *pzl,
               const double *pzu, const double *dinv, const double *coef, const double *const *P,
               int nv, double diag, double beta_mu, int64_t n, double *tprime) {
  if (nv > kMaxPanel) {
    const double *w = nullptr, one = 1.0;
    PO_TRY(collapse_range(c, 0, coef, P, 0, nv, n, &w));
    return k_res_step(c, b, rx, px, pzl, pzu, dinv, &one, &w, 1, diag, beta_mu, n, tprime);
  }
  count_bytes(c, nv + 10 + (dinv ? 1 : 0), n);
  if (n <= 0) return PO_OK;
  PtrTable pt;
  CoefTable ct;
  fill_tables(coef, P, nv, &ct, &pt);
  PO_LAUNCH(res_step_kernel, grid_for(c, n, kBpcPanel), b, rx, px, pzl, pzu, dinv, ct, pt, nv, diag, beta_mu, n,
            tprime);
  return PO_OK;
}

// complementarity at a trial step -------------------------------------------------------------------
__global__ void __launch_bounds__(kBlock)
    comp_step_kernel(Bounds b, const double *__restrict__ px, const double *__restrict__ pzl,
                     const double *__restrict__ pzu, double ax, double az, int64_t n,
                     double *__restrict__ partials) {
  __shared__ double sm[4 * 2];
  double sums[2] = {0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    const double xn0 = _x.x + ax * p.x, xn1 = _x.y + ax * p.y;
    if (e0.L) {
      sums[0] += (_zl.x + az * l.x) * (xn0 - _lb.x);
      sums[1] += 1.0;
    }
    if (e1.L) {
      sums[0] += (_zl.y + az * l.y) * (xn1 - _lb.y);
      sums[1] += 1.0;
    }
    if (e0.U) {
      sums[0] += (_zu.x + az * u.x) * (_ub.x - xn0);
      sums[1] += 1.0;
    }
    if (e1.U) {
      sums[0] += (_zu.y + az * u.y) * (_ub.y - xn1);
      sums[1] += 1.0;
    }
  }
  block_reduce_store<2, OP_SUM>(sums, partials, 0, sm);
}
int k_comp_step(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu,
                double ax, double az, int64_t n, double out[2]) {
  count_bytes(c, 8, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 2));
  PO_LAUNCH(comp_step_kernel, grid, b, px, pzl, pzu, ax, az, n, c->d_partials);
  return reduce_finish(c, grid, 2, 0, 0, out);
}

// merit function pieces ---------------------------------------------------------------------------
__device__ __forceinline__ void barrier_elem(const BE &e, double &pos, double &neg) {
  if (e.L) {
    const double v = log(e.xl);
    if (e.xl > 1.0) pos += v; else neg += v;
  }
  if (e.U) {
    const double v = log(e.xu);
    if (e.xu > 1.0) pos += v; else neg += v;
  }
}
__global__ void __launch_bounds__(kBlock)
    merit0_kernel(Bounds b, const double *__restrict__ px, double sx, const double *__restrict__ g,
                  int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 6];
  double s[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    double2 p = ld2(px, q, n);
    const double2 gv = ld2(g, q, n);
    p.x *= sx;
    p.y *= sx;
    barrier_elem(e0, s[0], s[1]);
    barrier_elem(e1, s[0], s[1]);
    // directional derivative of the log terms, split by sign (:3692-3714)
    if (e0.L) { if (p.x > 0.0) s[2] += p.x / e0.xl; else s[3] += p.x / e0.xl; }
    if (e1.L) { if (p.y > 0.0) s[2] += p.y / e1.xl; else s[3] += p.y / e1.xl; }
    if (e0.U) { if (p.x > 0.0) s[3] -= p.x / e0.xu; else s[2] -= p.x / e0.xu; }
    if (e1.U) { if (p.y > 0.0) s[3] -= p.y / e1.xu; else s[2] -= p.y / e1.xu; }
    s[4] += gv.x * p.x + gv.y * p.y;
    s[5] += p.x * p.x + p.y * p.y;
  }
  block_reduce_store<6, OP_SUM>(s, partials, 0, sm);
}
int k_merit0(Ctx *c, const Bounds &b, const double *px, double sx, const double *g, int64_t n,
             double out[6]) {
  count_bytes(c, 7, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 6));
  PO_LAUNCH(merit0_kernel, grid, b, px, sx, g, n, c->d_partials);
  return reduce_finish(c, grid, 6, 0, 0, out);
}

// comp_step + merit0 + max|px| in ONE pass over (x, lb, ub, zl, zu, px, pzl, pzu, g): the three
// reductions scaleKKTStep / evalMeritInitDeriv / the line search's step norm need of the same step.
// The merit pieces are taken for the UNSCALED step (the scaling sx > 0 is applied on the host:
// the sign split of px is invariant, ppos/pneg/g.px are linear and px.px quadratic in sx).
// sums {comp product, count, pos log, neg log, ppos, pneg, g.px, px.px}; max {|px|}
__global__ void __launch_bounds__(kBlock)
    comp_merit_kernel(Bounds b, const double *__restrict__ px, const double *__restrict__ pzl,
                      const double *__restrict__ pzu, double ax, double az, const double *__restrict__ g,
                      int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 8];
  double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double mx[1] = {0.0};
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 p = ld2(px, q, n), l = ld2(pzl, q, n), u = ld2(pzu, q, n), gv = ld2(g, q, n);
    const double xn0 = _x.x + ax * p.x, xn1 = _x.y + ax * p.y;
    if (e0.L) {
      s[0] += (_zl.x + az * l.x) * (xn0 - _lb.x);
      s[1] += 1.0;
    }
    if (e1.L) {
      s[0] += (_zl.y + az * l.y) * (xn1 - _lb.y);
      s[1] += 1.0;
    }
    if (e0.U) {
      s[0] += (_zu.x + az * u.x) * (_ub.x - xn0);
      s[1] += 1.0;
    }
    if (e1.U) {
      s[0] += (_zu.y + az * u.y) * (_ub.y - xn1);
      s[1] += 1.0;
    }
    barrier_elem(e0, s[2], s[3]);
    barrier_elem(e1, s[2], s[3]);
    if (e0.L) { if (p.x > 0.0) s[4] += p.x / e0.xl; else s[5] += p.x / e0.xl; }
    if (e1.L) { if (p.y > 0.0) s[4] += p.y / e1.xl; else s[5] += p.y / e1.xl; }
    if (e0.U) { if (p.x > 0.0) s[5] -= p.x / e0.xu; else s[4] -= p.x / e0.xu; }
    if (e1.U) { if (p.y > 0.0) s[5] -= p.y / e1.xu; else s[4] -= p.y / e1.xu; }
    s[6] += gv.x * p.x + gv.y * p.y;
    s[7] += p.x * p.x + p.y * p.y;
    mx[0] = fmax(mx[0], fmax(fabs(p.x), fabs(p.y)));
  }
  block_reduce_store<8, OP_SUM>(s, partials, 0, sm);
  block_reduce_store<1, OP_MAX>(mx, partials, 8, sm);
}
int k_comp_merit(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu, double ax,
                 double az, const double *g, int64_t n, double out[9]) {
  count_bytes(c, 9, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 9));
  PO_LAUNCH(comp_merit_kernel, grid, b, px, pzl, pzu, ax, az, g, n, c->d_partials);
  return reduce_finish(c, grid, 8, 0, 1, out);
}

// Corrector solve of the predictor-corrector strategy (round 6): solve2_kernel<0, 0> with the corrector terms re-formed
// from the affine step it overwrites (cl = [L] px pzl, cu = [U] px pzu: corrector_kernel's expressions), plus every
// sum scaleKKTStep's complementarity check and evalMeritInitDeriv need of the step it has in registers -- the separate
// pass comp_merit_kernel and its host round trip disappear.  The complementarity at the scaled step is the polynomial
// S00 + ax S10 + az S01 + ax az S11 of solve2r_kernel; the merit pieces and the barrier sums of the iterate are taken
// element by element as comp_merit_kernel takes them, on comp_merit_kernel's grid.
// slots: sums {S10, S01, S11, ppos, pneg, g.px, px.px, pos log, neg log}, minima {max_x, max_z}, maximum {|px|}
__global__ void __launch_bounds__(kBlock)
    solve2c_kernel(Bounds b, const double *__restrict__ t, const double *__restrict__ dinv, CoefTable alpha, PtrTable P,
                   int nv, double beta_mu, double tau, int64_t n, double *__restrict__ px, double *__restrict__ pzl,
                   double *__restrict__ pzu, double *__restrict__ va, int nca, const double *__restrict__ g,
                   int want_logs, double *__restrict__ partials) {
  __shared__ double sm[4 * 9];
  double mins[2] = {1.0, 1.0};
  double ms[9] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double mx[1] = {0.0};
  PO_PAIR_LOOP(q, n) {
    const double2 accA = panel_sum(P, alpha, nca, q);
    if (va) st2(va, q, n, accA);
    double2 acc = panel_sum(P, alpha, nv, q, nca);
    acc.x += accA.x;
    acc.y += accA.y;
    PO_LOAD_BOUNDS(b, q, n);
    const double2 tv = ld2(t, q, n), dv = ld2(dinv, q, n);
    const double dx0 = tv.x + dv.x * acc.x, dx1 = tv.y + dv.y * acc.y;
    const double2 pa = ld2(px, q, n), la = ld2(pzl, q, n), ua = ld2(pzu, q, n);  // the affine step
    const double c0x = e0.L ? __dmul_rn(pa.x, la.x) : 0.0, c0y = e1.L ? __dmul_rn(pa.y, la.y) : 0.0;  // (rounded: see
    const double c1x = e0.U ? __dmul_rn(pa.x, ua.x) : 0.0, c1y = e1.U ? __dmul_rn(pa.y, ua.y) : 0.0;  // corr_d1_dots_kernel)
    const Step3 s0 = solve2_elem<0>(e0, dx0, beta_mu, 0.0, 0.0, 0.0, c0x, c1x);
    Step3 s1 = solve2_elem<0>(e1, dx1, beta_mu, 0.0, 0.0, 0.0, c0y, c1y);
    if (!_has2) s1.px = s1.pzl = s1.pzu = 0.0;
    st2(px, q, n, make_double2(s0.px, s1.px));
    st2(pzl, q, n, make_double2(s0.pzl, s1.pzl));
    st2(pzu, q, n, make_double2(s0.pzu, s1.pzu));
    max_step_elem(b, _x.x, _lb.x, _ub.x, _zl.x, _zu.x, s0, tau, mins[0], mins[1]);
    if (_has2) max_step_elem(b, _x.y, _lb.y, _ub.y, _zl.y, _zu.y, s1, tau, mins[0], mins[1]);
    const double2 gv = ld2(g, q, n);
    if (e0.L) {
      ms[0] += _zl.x * s0.px;
      ms[1] += s0.pzl * e0.xl;
      ms[2] += s0.pzl * s0.px;
    }
    if (e1.L) {
      ms[0] += _zl.y * s1.px;
      ms[1] += s1.pzl * e1.xl;
      ms[2] += s1.pzl * s1.px;
    }
    if (e0.U) {
      ms[0] -= _zu.x * s0.px;
      ms[1] += s0.pzu * e0.xu;
      ms[2] -= s0.pzu * s0.px;
    }
    if (e1.U) {
      ms[0] -= _zu.y * s1.px;
      ms[1] += s1.pzu * e1.xu;
      ms[2] -= s1.pzu * s1.px;
    }
    if (want_logs) {
      barrier_elem(e0, ms[7], ms[8]);
      barrier_elem(e1, ms[7], ms[8]);
    }
    if (e0.L) { if (s0.px > 0.0) ms[3] += s0.px / e0.xl; else ms[4] += s0.px / e0.xl; }
    if (e1.L) { if (s1.px > 0.0) ms[3] += s1.px / e1.xl; else ms[4] += s1.px / e1.xl; }
    if (e0.U) { if (s0.px > 0.0) ms[4] -= s0.px / e0.xu; else ms[3] -= s0.px / e0.xu; }
    if (e1.U) { if (s1.px > 0.0) ms[4] -= s1.px / e1.xu; else ms[3] -= s1.px / e1.xu; }
    ms[5] += gv.x * s0.px + gv.y * s1.px;
    ms[6] += s0.px * s0.px + s1.px * s1.px;
    mx[0] = fmax(mx[0], fmax(fabs(s0.px), fabs(s1.px)));
  }
  block_reduce_store<9, OP_SUM>(ms, partials, 0, sm);
  block_reduce_store<2, OP_MIN>(mins, partials, 9, sm);
  block_reduce_store<1, OP_MAX>(mx, partials, 11, sm);
}
int k_solve2c(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
              const double *const *P, int nv, double beta_mu, double tau, int64_t n, double *px, double *pzl,
              double *pzu, double *va, int nca, const double *g, int want_logs, double out[12]) {
  if (nv > kMaxPanel) {
    set_error("k_solve2c: panel of %d vectors exceeds kMaxPanel=%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  count_bytes(c, nv + 14 + (va ? 1 : 0), n);
  const int grid = grid_for(c, n);  // comp_merit_kernel's grid
  PO_TRY(ensure_partials(c, (size_t)grid * 12));
  PtrTable pt;
  CoefTable ct;
  fill_tables(alpha, P, nv, &ct, &pt);
  PO_LAUNCH(solve2c_kernel, grid, b, t, dinv, ct, pt, nv, beta_mu, tau, n, px, pzl, pzu, va, nca, g, want_logs,
            c->d_partials);
  return reduce_finish(c, grid, 9, 2, 1, out);
}

__device__ __forceinline__ double clamp_elem(double v, bool has_l, double lb, bool has_u, double ub,
                                             double eps) {
  if (has_l && v <= lb + eps) v = lb + eps;
  if (has_u && v + eps >= ub) v = ub - eps;
  return v;
}
__global__ void __launch_bounds__(kBlock)
    trial_kernel(Bounds b, const double *__restrict__ px, double a, double eps, int64_t n,
                 double *__restrict__ xt, double *__restrict__ sout, double *__restrict__ partials) {
  __shared__ double sm[4 * 2];
  double s[2] = {0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    const double2 x = ld2(b.x, q, n), lb = ld2(b.lb, q, n), ub = ld2(b.ub, q, n), p = ld2(px, q, n);
    const bool has2 = (2 * q + 1 < n);
    // computeStep :3146-3191: the clamps use the bound vectors themselves (no predicate)
    double2 v;
    v.x = clamp_elem(x.x + a * p.x, true, lb.x, true, ub.x, eps);
    v.y = clamp_elem(x.y + a * p.y, true, lb.y, true, ub.y, eps);
    st2(xt, q, n, v);
    // the quasi-Newton step s = a px of this trial (the unclamped increment, as computeStepAndUpdate forms it)
    if (sout) st2(sout, q, n, make_double2(__dadd_rn(__dmul_rn(a, p.x), 0.0), __dadd_rn(__dmul_rn(a, p.y), 0.0)));
    BE e0 = bound_elem(v.x, lb.x, ub.x, 0.0, 0.0, b.max_bound, b.use_lower, b.use_upper);
    barrier_elem(e0, s[0], s[1]);
    if (has2) {
      BE e1 = bound_elem(v.y, lb.y, ub.y, 0.0, 0.0, b.max_bound, b.use_lower, b.use_upper);
      barrier_elem(e1, s[0], s[1]);
    }
  }
  block_reduce_store<2, OP_SUM>(s, partials, 0, sm);
}
int k_trial(Ctx *c, const Bounds &b, const double *px, double a, double eps, int64_t n, double *xt,
            double out[2], double *sout) {
  count_bytes(c, 5 + (sout ? 1 : 0), n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 2));
  PO_LAUNCH(trial_kernel, grid, b, px, a, eps, n, xt, sout, c->d_partials);
  return reduce_finish(c, grid, 2, 0, 0, out);
}

__global__ void __launch_bounds__(kBlock)
    update_mult_kernel(double *__restrict__ zl, const double *__restrict__ pzl,
                       double *__restrict__ zu, const double *__restrict__ pzu, double a, double eps,
                       int use_lower, int use_upper, int64_t n) {
  PO_PAIR_LOOP(q, n) {
    if (use_lower) {
      const double2 z = ld2(zl, q, n), p = ld2(pzl, q, n);
      st2(zl, q, n,
          make_double2(clamp_elem(z.x + a * p.x, true, 0.0, false, 0.0, eps),
                       clamp_elem(z.y + a * p.y, true, 0.0, false, 0.0, eps)));
    }
    if (use_upper) {
      const double2 z = ld2(zu, q, n), p = ld2(pzu, q, n);
      st2(zu, q, n,
          make_double2(clamp_elem(z.x + a * p.x, true, 0.0, false, 0.0, eps),
                       clamp_elem(z.y + a * p.y, true, 0.0, false, 0.0, eps)));
    }
  }
}
int k_update_mult(Ctx *c, double *zl, const double *pzl, double *zu, const double *pzu, double a,
                  double eps, int use_lower, int use_upper, int64_t n) {
  count_bytes(c, 3 * (use_lower + use_upper), n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(update_mult_kernel, grid_for(c, n), zl, pzl, zu, pzu, a, eps, use_lower, use_upper, n);
  return PO_OK;
}

// zl, zu update fused with the first half of the quasi-Newton gradient difference
//   y_qn = -g + A^T z+  =  rx - [lo] zl_old + [up] zu_old + az * (A^T pz)
// (rx = [lo] zl - [up] zu - g + A^T z is the KKT residual of this iteration; va = A^T pz was
// accumulated by the solves), replacing a pass over all constraint gradients (:4200-4206).
__global__ void __launch_bounds__(kBlock)
    update_mult_yqn_kernel(double *__restrict__ zl, const double *__restrict__ pzl,
                           double *__restrict__ zu, const double *__restrict__ pzu, double a,
                           double eps, int use_lower, int use_upper, const double *__restrict__ rx,
                           const double *__restrict__ va, double az, int64_t n,
                           double *__restrict__ yqn, double *__restrict__ acz) {
  PO_PAIR_LOOP(q, n) {
    const double2 r = ld2(rx, q, n), w = ld2(va, q, n);
    // (explicitly rounded operations: kkt_res_update_kernel forms the same sums and must give the same bits)
    double2 y = make_double2(__fma_rn(az, w.x, r.x), __fma_rn(az, w.y, r.y));
    if (acz) {  // A^T z of a problem with a constant Jacobian follows the multiplier step: += az * A^T pz
      const double2 c0 = ld2(acz, q, n);
      st2(acz, q, n, make_double2(__fma_rn(az, w.x, c0.x), __fma_rn(az, w.y, c0.y)));
    }
    if (use_lower) {
      const double2 z = ld2(zl, q, n), p = ld2(pzl, q, n);
      y.x = __dsub_rn(y.x, z.x);
      y.y = __dsub_rn(y.y, z.y);
      st2(zl, q, n,
          make_double2(clamp_elem(__fma_rn(a, p.x, z.x), true, 0.0, false, 0.0, eps),
                       clamp_elem(__fma_rn(a, p.y, z.y), true, 0.0, false, 0.0, eps)));
    }
    if (use_upper) {
      const double2 z = ld2(zu, q, n), p = ld2(pzu, q, n);
      y.x = __dadd_rn(y.x, z.x);
      y.y = __dadd_rn(y.y, z.y);
      st2(zu, q, n,
          make_double2(clamp_elem(__fma_rn(a, p.x, z.x), true, 0.0, false, 0.0, eps),
                       clamp_elem(__fma_rn(a, p.y, z.y), true, 0.0, false, 0.0, eps)));
    }
    st2(yqn, q, n, y);
  }
}
int k_update_mult_yqn(Ctx *c, double *zl, const double *pzl, double *zu, const double *pzu, double a,
                      double eps, int use_lower, int use_upper, const double *rx, const double *va,
                      double az, int64_t n, double *yqn, double *acz) {
  count_bytes(c, 3 + 3 * (use_lower + use_upper) + (acz ? 2 : 0), n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(update_mult_yqn_kernel, grid_for(c, n), zl, pzl, zu, pzu, a, eps, use_lower, use_upper, rx,
            va, az, n, yqn, acz);
  return PO_OK;
}

// lean step: (pzl, pzu) from the design step by the first-solve formula, as kkt_res_update_kernel forms them in
// registers -- for the consumers that want them as vectors (checkKKTStep)
__global__ void __launch_bounds__(kBlock)
    form_pz_kernel(Bounds b, const double *__restrict__ px, double beta_mu, int64_t n, double *__restrict__ pzl,
                   double *__restrict__ pzu) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 pv = ld2(px, q, n);
    const Step3 t0 = solve2_elem<0>(e0, pv.x, beta_mu, 0.0, 0.0, 0.0);
    const Step3 t1 = solve2_elem<0>(e1, pv.y, beta_mu, 0.0, 0.0, 0.0);
    st2(pzl, q, n, make_double2(t0.pzl, t1.pzl));
    st2(pzu, q, n, make_double2(t0.pzu, t1.pzu));
  }
}
int k_form_pz(Ctx *c, const Bounds &b, const double *px, double beta_mu, int64_t n, double *pzl, double *pzu) {
  if (n <= 0) return PO_OK;
  count_bytes(c, 8, n);
  PO_LAUNCH(form_pz_kernel, grid_for(c, n), b, px, beta_mu, n, pzl, pzu);
  return PO_OK;
}

// update_mult_yqn_kernel and kkt_res_kernel (with its y_qn completion) in ONE pass, run after the gradient of the new
// point is known: the bound multipliers take their step here (nothing between the two original launches reads them:
// the problem callbacks see x and the dense multipliers only), the first bracket of y_qn is formed from the OLD
// (rx, zl, zu) and the second from the NEW ones, and rx / the norms are those of the new point.  One pass over the
// bound data instead of two, one output stream (y_qn) and four input streams less.  Same arithmetic, same order.
//   acz != nullptr: A^T z is that vector (+ az_acz * va first: the recurrence of the linear-constraint mode);
//   otherwise A^T z = sum_j z_j A_j over the nc panel columns.
__global__ void __launch_bounds__(kBlock)
    kkt_res_update_kernel(Bounds b, const double *__restrict__ g, PtrTable A, CoefTable z, int nc, double beta_mu,
                          int64_t n, double *__restrict__ rx, double *__restrict__ yqn, double *__restrict__ zl,
                          const double *__restrict__ pzl, double *__restrict__ zu, const double *__restrict__ pzu,
                          double a, double eps, const double *__restrict__ va, double az, double *__restrict__ acz,
                          double az_acz, const double *__restrict__ pxs, const double *__restrict__ xold,
                          double beta_mu_step, double beta_mu2, GroupCol gcol, double gcoef,
                          double *__restrict__ dinv_out, double *__restrict__ t_out, double spec_diag,
                          double spec_beta_mu, double *__restrict__ partials) {
  __shared__ double sm[4 * 8];
  double sums[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  double maxs[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  PO_PAIR_LOOP(q, n) {
    GRaw graw;  // the grouped column's entries: requested first, used by the panel sum below
    if (gcol.w) graw = gcol_request(gcol, q, n);
    const double2 _x = ld2(b.x, q, n), _lb = ld2(b.lb, q, n), _ub = ld2(b.ub, q, n);
    // pxs != nullptr ("lean step"): the refinement pass did not store the bound-multiplier steps; they are formed
    // here from the design step px, the OLD point (xold) and the old multipliers by the first-solve formula
    // pzl = [L] (rzl - zl px) / (x - lb), pzu = [U] (rzu + zu px) / (ub - x) -- which the refined values equal up to the
    // round-off the refinement itself corrects (two output streams of the refinement pass saved)
    double2 lean_pl = make_double2(0.0, 0.0), lean_pu = lean_pl;
    if (pxs) {
      const double2 pv = ld2(pxs, q, n), xo = ld2(xold, q, n);
      const double2 zlo = ld2(zl, q, n), zuo = ld2(zu, q, n);
      const BE o0 = bound_elem(xo.x, _lb.x, _ub.x, zlo.x, zuo.x, b.max_bound, b.use_lower, b.use_upper);
      BE o1 = bound_elem(xo.y, _lb.y, _ub.y, zlo.y, zuo.y, b.max_bound, b.use_lower, b.use_upper);
      if (2 * q + 1 >= n) {
        o1.L = false;
        o1.U = false;
        o1.xl = 1.0;
        o1.xu = 1.0;
      }
      const Step3 t0 = solve2_elem<0>(o0, pv.x, beta_mu_step, 0.0, 0.0, 0.0);
      const Step3 t1 = solve2_elem<0>(o1, pv.y, beta_mu_step, 0.0, 0.0, 0.0);
      lean_pl = make_double2(t0.pzl, t1.pzl);
      lean_pu = make_double2(t0.pzu, t1.pzu);
    }
    // (yqn == nullptr: no quasi-Newton update follows -- the multiplier step and the residual of the new point only)
    const double2 zero2 = make_double2(0.0, 0.0);
    const double2 r0 = yqn ? ld2(rx, q, n) : zero2, w = va ? ld2(va, q, n) : zero2, gv = ld2(g, q, n);
    // first bracket, as update_mult_yqn_kernel
    double2 y = make_double2(__fma_rn(az, w.x, r0.x), __fma_rn(az, w.y, r0.y));
    double2 aczv = make_double2(0.0, 0.0);
    if (acz) {
      const double2 c0 = ld2(acz, q, n);
      aczv = az_acz != 0.0 ? make_double2(__fma_rn(az_acz, w.x, c0.x), __fma_rn(az_acz, w.y, c0.y)) : c0;
      if (az_acz != 0.0) st2(acz, q, n, aczv);
    }
    double2 _zl = make_double2(0.0, 0.0), _zu = _zl;
    if (b.use_lower) {
      const double2 zo = ld2(zl, q, n), p = pxs ? lean_pl : ld2(pzl, q, n);
      y.x = __dsub_rn(y.x, zo.x);
      y.y = __dsub_rn(y.y, zo.y);
      _zl = make_double2(clamp_elem(__fma_rn(a, p.x, zo.x), true, 0.0, false, 0.0, eps),
                         clamp_elem(__fma_rn(a, p.y, zo.y), true, 0.0, false, 0.0, eps));
      st2(zl, q, n, _zl);
      if (2 * q + 1 >= n) _zl.y = 0.0;  // what a later load of the padded pair would see
    } else {
      _zl = ld2(zl, q, n);
    }
    if (b.use_upper) {
      const double2 zo = ld2(zu, q, n), p = pxs ? lean_pu : ld2(pzu, q, n);
      y.x = __dadd_rn(y.x, zo.x);
      y.y = __dadd_rn(y.y, zo.y);
      _zu = make_double2(clamp_elem(__fma_rn(a, p.x, zo.x), true, 0.0, false, 0.0, eps),
                         clamp_elem(__fma_rn(a, p.y, zo.y), true, 0.0, false, 0.0, eps));
      st2(zu, q, n, _zu);
      if (2 * q + 1 >= n) _zu.y = 0.0;
    } else {
      _zu = ld2(zu, q, n);
    }
    PO_MAKE_BOUNDS(b, q, n);
    // residual of the new point, as kkt_res_kernel
    double2 r;
    r.x = (b.use_lower ? _zl.x : 0.0);
    r.y = (b.use_lower ? _zl.y : 0.0);
    if (b.use_upper) {
      r.x += -1.0 * _zu.x;
      r.y += -1.0 * _zu.y;
    }
    r.x += -1.0 * gv.x;
    r.y += -1.0 * gv.y;
    double2 ps;
    if (acz) {
      ps = make_double2(fma(1.0, aczv.x, 0.0), fma(1.0, aczv.y, 0.0));
    } else {
      ps = panel_sum(A, z, nc, q);
      if (gcol.w) {
        const f64x2 gcv = gcol_value(gcol, graw);
        ps.x += gcoef * gcv.x;
        ps.y += gcoef * gcv.y;
      }
    }
    r.x += ps.x;
    r.y += ps.y;
    if (!_has2) r.y = 0.0;
    st2(rx, q, n, r);
    if (dinv_out) {
      // Dinv and t = Dinv o d1 of the NEXT iteration's first solve, from the operands this pass holds anyway (round 6;
      // the expressions of dinv_d1_kernel: the same bits): when no quasi-Newton update follows, the diagonal of the
      // next KKT system is known here, and the pass over the bound data at the head of setUpKKTSystem disappears
      const double2 dv = make_double2(dinv_elem(e0, spec_diag), dinv_elem(e1, spec_diag));
      st2(dinv_out, q, n, dv);
      st2(t_out, q, n, make_double2(dv.x * d1_elem(e0, r.x, spec_beta_mu), dv.y * d1_elem(e1, r.y, spec_beta_mu)));
    }
    const double cl = b.use_lower ? 1.0 : 0.0, cu = b.use_upper ? -1.0 : 0.0;
    if (yqn) {
      y.x = __dadd_rn(y.x, __fma_rn(-1.0, r.x, __fma_rn(cu, _zu.x, __fma_rn(cl, _zl.x, 0.0))));
      y.y = __dadd_rn(y.y, __fma_rn(-1.0, r.y, __fma_rn(cu, _zu.y, __fma_rn(cl, _zl.y, 0.0))));
      st2(yqn, q, n, y);
    }
    maxs[0] = fmax(maxs[0], fmax(fabs(r.x), fabs(r.y)));
    sums[2] += fabs(r.x) + fabs(r.y);
    sums[5] += r.x * r.x + r.y * r.y;
    res_bound_acc(e0, beta_mu, sums, maxs, 3, 6, 1);
    res_bound_acc(e1, beta_mu, sums, maxs, 3, 6, 1);
    if (beta_mu2 >= 0.0) {
      res_bound_max2(e0, beta_mu2, maxs + 3);
      res_bound_max2(e1, beta_mu2, maxs + 3);
    }
  }
  block_reduce_store<8, OP_SUM>(sums, partials, 0, sm);
  if (beta_mu2 >= 0.0) {
    block_reduce_store<5, OP_MAX>(maxs, partials, 8, sm);
  } else {
    double m3[3] = {maxs[0], maxs[1], maxs[2]};
    block_reduce_store<3, OP_MAX>(m3, partials, 8, sm);
  }
}

int k_kkt_res_update(Ctx *c, const Bounds &b, const double *g, const double *const *A, const double *z, int nc,
                     double beta_mu, int64_t n, double *rx, double *out, double *yqn, double *zl,
                     const double *pzl, double *zu, const double *pzu, double a, double eps, const double *va,
                     double az, double *acz, double az_acz, const double *pxs, const double *xold,
                     double beta_mu_step, double beta_mu2, const GroupCol *gcol, double gcoef, double *dinv_out,
                     double *t_out, double spec_diag, double spec_beta_mu) {
  PO_TRY(gcol_check(gcol, n, "k_kkt_res_update"));
  if (acz && gcol) {
    set_error("kkt_res_update: a grouped column cannot follow the A^T z vector of the linear-constraint mode");
    return PO_ERR_ARG;
  }
  if (nc > kMaxPanel) {
    const double *w = nullptr, one = 1.0;
    PO_TRY(collapse_range(c, 0, z, A, 0, nc, n, &w));
    return k_kkt_res_update(c, b, g, &w, &one, 1, beta_mu, n, rx, out, yqn, zl, pzl, zu, pzu, a, eps, va, az, acz, az_acz,
                            pxs, xold, beta_mu_step, beta_mu2, gcol, gcoef, dinv_out, t_out, spec_diag, spec_beta_mu);
  }
  count_bytes(c, (yqn ? 14 : 11) + (acz ? (az_acz != 0.0 ? 2 : 1) : nc) + (dinv_out ? 2 : 0), n);
  const int grid = grid_for(c, n, kBpcPanel);
  PO_TRY(ensure_partials(c, (size_t)grid * 13));
  PtrTable pt;
  CoefTable ct;
  fill_tables(z, A, nc, &ct, &pt);
  if (gcol) count_bytes(c, 1.0, gcol->nwcon);
  PO_LAUNCH(kkt_res_update_kernel, grid, b, g, pt, ct, nc, beta_mu, n, rx, yqn, zl, pzl, zu, pzu, a, eps, va, az, acz,
            az_acz, pxs, xold, beta_mu_step, beta_mu2, gcol ? *gcol : GroupCol(), gcoef, dinv_out, t_out, spec_diag,
            spec_beta_mu, c->d_partials);
  return reduce_finish(c, grid, 8, 0, beta_mu2 >= 0.0 ? 5 : 3, out);
}

__global__ void __launch_bounds__(kBlock)
    affine_mult_kernel(Bounds b, double *__restrict__ zl, const double *__restrict__ pzl,
                       double *__restrict__ zu, const double *__restrict__ pzu, double amin,
                       int64_t n) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 l = ld2(pzl, q, n), u = ld2(pzu, q, n);
    double2 nl = _zl, nu = _zu;
    if (e0.L) nl.x = fmax(amin, fabs(_zl.x + l.x));
    if (e1.L) nl.y = fmax(amin, fabs(_zl.y + l.y));
    if (e0.U) nu.x = fmax(amin, fabs(_zu.x + u.x));
    if (e1.U) nu.y = fmax(amin, fabs(_zu.y + u.y));
    if (b.use_lower) st2(zl, q, n, nl);
    if (b.use_upper) st2(zu, q, n, nu);
  }
}
int k_affine_mult(Ctx *c, const Bounds &b, double *zl, const double *pzl, double *zu,
                  const double *pzu, double amin, int64_t n) {
  count_bytes(c, 9, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(affine_mult_kernel, grid_for(c, n), b, zl, pzl, zu, pzu, amin, n);
  return PO_OK;
}

__global__ void __launch_bounds__(kBlock)
    check_bounds_kernel(double *__restrict__ x, double *__restrict__ lb, double *__restrict__ ub,
                        double *__restrict__ zl, double *__restrict__ zu, double maxb,
                        double rel_bound, int both, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 3];
  double flags[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    double xv = x[i], l = lb[i], u = ub[i];
    if (both) {
      double delta = 1.0;
      if (l > -maxb && u < maxb) {
        if (l >= u) {
          flags[0] = 1.0;
          l = 0.5 * (l + u) - 0.5 * rel_bound;
          u = l + rel_bound;
          lb[i] = l;
          ub[i] = u;
        }
        delta = u - l;
      }
      if (l > -maxb && xv < l + rel_bound * delta) {
        flags[1] = 1.0;
        xv = l + rel_bound * delta;
      }
      if (u < maxb && xv > u - rel_bound * delta) {
        flags[2] = 1.0;
        xv = u - rel_bound * delta;
      }
      x[i] = xv;
    }
    if (l <= -maxb) zl[i] = 0.0;
    if (u >= maxb) zu[i] = 0.0;
  }
  block_reduce_store<3, OP_MAX>(flags, partials, 0, sm);
}
int k_check_bounds(Ctx *c, double *x, double *lb, double *ub, double *zl, double *zu,
                   double max_bound, double rel_bound, int both, int64_t n, int *flag) {
  count_bytes(c, 5, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 3));
  PO_LAUNCH(check_bounds_kernel, grid, x, lb, ub, zl, zu, max_bound, rel_bound, both, n,
            c->d_partials);
  double out[3];
  PO_TRY(reduce_finish(c, grid, 0, 0, 3, out, true));
  *flag = (out[0] > 0.0 ? 1 : 0) | (out[1] > 0.0 ? 2 : 0) | (out[2] > 0.0 ? 4 : 0);
  return PO_OK;
}

// test data for initAndCheckDesignAndBounds (po_problem_set_bounds_mode; oracle/ref_driver.cpp SepProblem), by
// GLOBAL index gi: bit 1: gi % 7 == 3 -> lb = ub = midpoint; bit 2: gi % 11 == 5 -> x = lb; bit 4: gi % 13 == 6 -> x = ub
__global__ void __launch_bounds__(kBlock)
    bounds_mode_kernel(double *__restrict__ x, double *__restrict__ lb, double *__restrict__ ub, int mode,
                       int64_t offset, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t gi = offset + i;
    if ((mode & 1) && gi % 7 == 3) lb[i] = ub[i] = 0.5 * (lb[i] + ub[i]);
    if ((mode & 2) && gi % 11 == 5) x[i] = lb[i];
    if ((mode & 4) && gi % 13 == 6) x[i] = ub[i];
  }
}
int k_bounds_mode(Ctx *c, double *x, double *lb, double *ub, int mode, int64_t offset, int64_t n) {
  if (n <= 0 || mode == 0) return PO_OK;
  PO_LAUNCH(bounds_mode_kernel, grid_for(c, n), x, lb, ub, mode, offset, n);
  return PO_OK;
}

// clamp bookkeeping (SURVEY 8a'): entries sitting exactly at the clamp values lb + eps, ub - eps, eps that
// computeStepVec / computeStepAndUpdate (src/ParOptInteriorPoint.cpp:3150-3190, 4177-4195) leave behind;
// counts are summed as exact doubles
__global__ void __launch_bounds__(kBlock)
    clamp_count_kernel(const double *__restrict__ x, const double *__restrict__ lb, const double *__restrict__ ub,
                       const double *__restrict__ zl, const double *__restrict__ zu, double eps, int64_t n,
                       double *__restrict__ partials) {
  __shared__ double sm[4 * 4];
  double cnt[4] = {0.0, 0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (x[i] == lb[i] + eps) cnt[0] += 1.0;
    if (x[i] == ub[i] - eps) cnt[1] += 1.0;
    if (zl && zl[i] == eps) cnt[2] += 1.0;
    if (zu && zu[i] == eps) cnt[3] += 1.0;
  }
  block_reduce_store<4, OP_SUM>(cnt, partials, 0, sm);
}
int k_clamp_count(Ctx *c, const double *x, const double *lb, const double *ub, const double *zl, const double *zu,
                  double eps, int64_t n, double out[4]) {
  count_bytes(c, 5, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 4));
  PO_LAUNCH(clamp_count_kernel, grid, x, lb, ub, zl, zu, eps, n, c->d_partials);
  return reduce_finish(c, grid, 4, 0, 0, out);
}

__global__ void __launch_bounds__(kBlock)
    zero_inactive_kernel(const double *__restrict__ lb, const double *__restrict__ ub,
                         double *__restrict__ zl, double *__restrict__ zu, double maxb, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    if (lb[i] <= -maxb) zl[i] = 0.0;
    if (ub[i] >= maxb) zu[i] = 0.0;
  }
}
int k_zero_inactive(Ctx *c, const double *lb, const double *ub, double *zl, double *zu,
                    double max_bound, int64_t n) {
  count_bytes(c, 4, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(zero_inactive_kernel, grid_for(c, n), lb, ub, zl, zu, max_bound, n);
  return PO_OK;
}

// ---------------------------------------------------------------------------------------------
// built-in problems (DESIGN.md "Workloads")
// ---------------------------------------------------------------------------------------------
template <int KIND>  // 0 quadratic, 1 convex
__global__ void __launch_bounds__(kBlock)
    sep_f_kernel(const double *__restrict__ qv, const double *__restrict__ bv,
                 const double *__restrict__ x, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4];
  double acc[1] = {0.0};
  PO_PAIR_LOOP(q, n) {
    const double2 xv = ld2(x, q, n), b = ld2(bv, q, n);
    const bool has2 = (2 * q + 1 < n);
    if (KIND == 0) {
      const double2 qq = ld2(qv, q, n);
      acc[0] += 0.5 * qq.x * xv.x * xv.x + b.x * xv.x;
      if (has2) acc[0] += 0.5 * qq.y * xv.y * xv.y + b.y * xv.y;
    } else {
      acc[0] += b.x * b.x / (1e-3 + xv.x);
      if (has2) acc[0] += b.y * b.y / (1e-3 + xv.y);
    }
  }
  block_reduce_store<1, OP_SUM>(acc, partials, 0, sm);
}
template <int KIND>
__global__ void __launch_bounds__(kBlock)
    sep_g_kernel(const double *__restrict__ qv, const double *__restrict__ bv,
                 const double *__restrict__ x, int64_t n, double *__restrict__ g) {
  PO_PAIR_LOOP(q, n) {
    const double2 xv = ld2(x, q, n), b = ld2(bv, q, n);
    double2 r;
    if (KIND == 0) {
      const double2 qq = ld2(qv, q, n);
      r.x = qq.x * xv.x + b.x;
      r.y = qq.y * xv.y + b.y;
    } else {
      const double d0 = 1e-3 + xv.x, d1 = 1e-3 + xv.y;
      r.x = -(b.x * b.x) / (d0 * d0);
      r.y = -(b.y * b.y) / (d1 * d1);
    }
    st2(g, q, n, r);
  }
}
int k_quadratic_f(Ctx *c, const double *q, const double *b, const double *x, int64_t n, double *f) {
  count_bytes(c, 3, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid));
  PO_LAUNCH(sep_f_kernel<0>, grid, q, b, x, n, c->d_partials);
  return reduce_finish(c, grid, 1, 0, 0, f);
}
int k_convex_f(Ctx *c, const double *b, const double *x, int64_t n, double *f) {
  count_bytes(c, 2, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid));
  PO_LAUNCH(sep_f_kernel<1>, grid, (const double *)nullptr, b, x, n, c->d_partials);
  return reduce_finish(c, grid, 1, 0, 0, f);
}
int k_quadratic_g(Ctx *c, const double *q, const double *b, const double *x, int64_t n, double *g) {
  count_bytes(c, 4, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(sep_g_kernel<0>, grid_for(c, n), q, b, x, n, g);
  return PO_OK;
}
int k_convex_g(Ctx *c, const double *b, const double *x, int64_t n, double *g) {
  count_bytes(c, 3, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(sep_g_kernel<1>, grid_for(c, n), (const double *)nullptr, b, x, n, g);
  return PO_OK;
}

// Hessian of the Lagrangian of the separable workloads: diag h_i, or h_i * px_i when px != null
template <int KIND>
__global__ void __launch_bounds__(kBlock)
    sep_h_kernel(const double *__restrict__ qv, const double *__restrict__ bv,
                 const double *__restrict__ x, const double *__restrict__ px, int64_t n,
                 double *__restrict__ h) {
  PO_PAIR_LOOP(q, n) {
    double2 r;
    if (KIND == 0) {
      r = ld2(qv, q, n);
    } else {
      const double2 xv = ld2(x, q, n), b = ld2(bv, q, n);
      const double d0 = 1e-3 + xv.x, d1 = 1e-3 + xv.y;
      r.x = 2.0 * b.x * b.x / (d0 * d0 * d0);
      r.y = 2.0 * b.y * b.y / (d1 * d1 * d1);
    }
    if (px) {
      const double2 p = ld2(px, q, n);
      r.x *= p.x;
      r.y *= p.y;
    }
    st2(h, q, n, r);
  }
}
int k_sep_hess(Ctx *c, int kind, const double *q, const double *b, const double *x, const double *px,
               int64_t n, double *h) {
  count_bytes(c, 4, n);
  if (n <= 0) return PO_OK;
  if (kind == 0) {
    PO_LAUNCH(sep_h_kernel<0>, grid_for(c, n), q, b, x, px, n, h);
  } else {
    PO_LAUNCH(sep_h_kernel<1>, grid_for(c, n), q, b, x, px, n, h);
  }
  return PO_OK;
}
// chained Rosenbrock + z0 * (0.25 - sum x^2): tridiagonal Hessian times px, or its diagonal
__global__ void __launch_bounds__(kBlock)
    rosen_h_kernel(const double *__restrict__ x, double z0, const double *__restrict__ px, int64_t n,
                   double *__restrict__ h) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double xi = x[i];
    double d = 2.0 * z0, lo = 0.0, up = 0.0;  // diagonal, coupling to i-1, coupling to i+1
    if (i + 1 < n) {
      d += 2.0 - 400.0 * (x[i + 1] - xi * xi) + 800.0 * xi * xi;
      up = -400.0 * xi;
    }
    if (i > 0) {
      d += 200.0;
      lo = -400.0 * x[i - 1];
    }
    if (px) {
      double v = d * px[i];
      if (i + 1 < n) v += up * px[i + 1];
      if (i > 0) v += lo * px[i - 1];
      h[i] = v;
    } else {
      h[i] = d;
    }
  }
}
int k_rosen_hess(Ctx *c, const double *x, double z0, const double *px, int64_t n, double *h) {
  count_bytes(c, 3, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(rosen_h_kernel, grid_for(c, n), x, z0, px, n, h);
  return PO_OK;
}

// ---- Newton-Krylov support: the alpha-scaled bordered solve of solveKKTDiagSystem :2441-2614 ---------
// t = Dinv * (bx + alpha * ([L] rzl/(x-lb) - [U] rzu/(ub-x))), rzl / rzu recomputed from beta_mu
__global__ void __launch_bounds__(kBlock)
    d1s_kernel(Bounds b, const double *__restrict__ bx, const double *__restrict__ dinv, double alpha,
               double beta_mu, int64_t n, double *__restrict__ t) {
  PO_PAIR_LOOP(q, n) {
    PO_LOAD_BOUNDS(b, q, n);
    const double2 r = ld2(bx, q, n), dv = dinv ? ld2(dinv, q, n) : make_double2(1.0, 1.0);  // null: raw d1
    double d0 = r.x, d1 = r.y;
    if (e0.L) d0 += alpha * (-(e0.xl * e0.zl - beta_mu)) / e0.xl;
    if (e0.U) d0 -= alpha * (-(e0.xu * e0.zu - beta_mu)) / e0.xu;
    if (e1.L) d1 += alpha * (-(e1.xl * e1.zl - beta_mu)) / e1.xl;
    if (e1.U) d1 -= alpha * (-(e1.xu * e1.zu - beta_mu)) / e1.xu;
    st2(t, q, n, make_double2(dv.x * d0, dv.y * d1));
  }
}
int k_d1s(Ctx *c, const Bounds &b, const double *bx, const double *dinv, double alpha, double beta_mu,
          int64_t n, double *t) {
  count_bytes(c, 8, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(d1s_kernel, grid_for(c, n), b, bx, dinv, alpha, beta_mu, n, t);
  return PO_OK;
}
// px = t + Dinv * sum coef_j P_j ; FULL: pzl = [L](alpha rzl - zl px)/(x-lb), pzu = [U](alpha rzu + zu px)/(ub-x)
// and the fraction-to-boundary minima {max_x, max_z} with fraction tau
template <int FULL>
__global__ void __launch_bounds__(kBlock)
    solve2s_kernel(Bounds b, const double *__restrict__ t, const double *__restrict__ dinv, CoefTable coef,
                   PtrTable P, int nv, double alpha, double beta_mu, double tau, int64_t n,
                   double *__restrict__ px, double *__restrict__ pzl, double *__restrict__ pzu,
                   double *__restrict__ partials) {
  __shared__ double sm[4 * 2];
  double mins[2] = {1.0, 1.0};
  PO_PAIR_LOOP(q, n) {
    const double2 acc = panel_sum(P, coef, nv, q);
    const double2 tv = ld2(t, q, n), dv = ld2(dinv, q, n);
    const double p0 = tv.x + dv.x * acc.x, p1 = tv.y + dv.y * acc.y;
    if (FULL) {
      PO_LOAD_BOUNDS(b, q, n);
      Step3 s0, s1;
      s0.px = p0;
      s1.px = p1;
      s0.pzl = e0.L ? (alpha * (-(e0.xl * e0.zl - beta_mu)) - e0.zl * p0) / e0.xl : 0.0;
      s0.pzu = e0.U ? (alpha * (-(e0.xu * e0.zu - beta_mu)) + e0.zu * p0) / e0.xu : 0.0;
      s1.pzl = e1.L ? (alpha * (-(e1.xl * e1.zl - beta_mu)) - e1.zl * p1) / e1.xl : 0.0;
      s1.pzu = e1.U ? (alpha * (-(e1.xu * e1.zu - beta_mu)) + e1.zu * p1) / e1.xu : 0.0;
      if (!_has2) s1.px = s1.pzl = s1.pzu = 0.0;
      st2(px, q, n, make_double2(s0.px, s1.px));
      st2(pzl, q, n, make_double2(s0.pzl, s1.pzl));
      st2(pzu, q, n, make_double2(s0.pzu, s1.pzu));
      max_step_elem(b, _x.x, _lb.x, _ub.x, _zl.x, _zu.x, s0, tau, mins[0], mins[1]);
      if (_has2) max_step_elem(b, _x.y, _lb.y, _ub.y, _zl.y, _zu.y, s1, tau, mins[0], mins[1]);
    } else {
      st2(px, q, n, make_double2(p0, p1));
    }
  }
  if (FULL) block_reduce_store<2, OP_MIN>(mins, partials, 0, sm);
}
int k_solve2s(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *coef,
              const double *const *P, int nv, double alpha, double beta_mu, int full, double tau, int64_t n,
              double *px, double *pzl, double *pzu, double out[2]) {
  if (nv > kMaxPanel) {
    const double *w = nullptr, one = 1.0;
    PO_TRY(collapse_range(c, 0, coef, P, 0, nv, n, &w));
    return k_solve2s(c, b, t, dinv, &one, &w, 1, alpha, beta_mu, full, tau, n, px, pzl, pzu, out);
  }
  count_bytes(c, nv + 10, n);
  if (nv > kMaxPanel) {
    set_error("panel of %d vectors exceeds kMaxPanel=%d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  const int grid = grid_for(c, n, kBpcPanel);
  PtrTable pt;
  CoefTable ct;
  fill_tables(coef, P, nv, &ct, &pt);
  if (full) {
    PO_TRY(ensure_partials(c, (size_t)grid * 2));
    PO_LAUNCH((solve2s_kernel<1>), grid, b, t, dinv, ct, pt, nv, alpha, beta_mu, tau, n, px, pzl, pzu,
              c->d_partials);
    return reduce_finish(c, grid, 0, 2, 0, out);
  }
  if (n <= 0) return PO_OK;
  PO_LAUNCH((solve2s_kernel<0>), grid, b, t, dinv, ct, pt, nv, alpha, beta_mu, tau, n, px, pzl, pzu,
            c->d_partials);
  return PO_OK;
}

// rank-local chained Rosenbrock (examples/rosenbrock/rosenbrock.cpp:49-107)
__global__ void __launch_bounds__(kBlock)
    rosen_f_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ partials) {
  __shared__ double sm[4 * 3];
  double s[3] = {0.0, 0.0, 0.0};
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double xi = x[i];
    if (i + 1 < n) {
      const double xn = x[i + 1];
      s[0] += (1.0 - xi) * (1.0 - xi) + 100.0 * (xn - xi * xi) * (xn - xi * xi);
    }
    s[1] -= xi * xi;
    if ((i & 1) == 0) s[2] += xi;
  }
  block_reduce_store<3, OP_SUM>(s, partials, 0, sm);
}
__global__ void __launch_bounds__(kBlock)
    rosen_g_kernel(const double *__restrict__ x, int64_t n, double *__restrict__ g,
                   double *__restrict__ a0, double *__restrict__ a1) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const double xi = x[i];
    double gi = 0.0;
    if (i + 1 < n) gi += -2.0 * (1.0 - xi) + 200.0 * (x[i + 1] - xi * xi) * (-2.0 * xi);
    if (i > 0) gi += 200.0 * (xi - x[i - 1] * x[i - 1]);
    g[i] = gi;
    a0[i] = -2.0 * xi;
    a1[i] = ((i & 1) == 0) ? 1.0 : 0.0;
  }
}
int k_rosen_f(Ctx *c, const double *x, int64_t n, double out[3]) {
  count_bytes(c, 1, n);
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 3));
  PO_LAUNCH(rosen_f_kernel, grid, x, n, c->d_partials);
  return reduce_finish(c, grid, 3, 0, 0, out);
}
int k_rosen_g(Ctx *c, const double *x, int64_t n, double *g, double *a0, double *a1) {
  count_bytes(c, 4, n);
  if (n <= 0) return PO_OK;
  PO_LAUNCH(rosen_g_kernel, grid_for(c, n), x, n, g, a0, a1);
  return PO_OK;
}

}  // namespace po
