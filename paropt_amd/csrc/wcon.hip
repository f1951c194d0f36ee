// Kernels of the weighting / sparse-constraint blocks (nwcon > 0, block-diagonal Aw D^-1 Aw^T with
// nwblock = 1): the w-sized pieces of ParOptInteriorPoint (zw, sw, tw, zsw, ztw) and of
// ParOptQuasiDefBlockMat (reference src/ParOptSparseMat.cpp:11-229), plus the structured Jacobian
// of the built-in workloads (one constraint per group of `nw` consecutive variables, the pattern of
// examples/rosenbrock/rosenbrock.cpp:131-184).  w is a small fraction of n (config 4: n/20), so
// these are plain one-element-per-thread kernels; the n-sized passes stay in kernels.hip.
#include <math.h>

#include "core.hpp"
#include "wcon.hpp"

namespace po {

#define PO_W_LOOP(i, n)                                                                   \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);               \
       i += (int64_t)gridDim.x * blockDim.x)

#define PO_WLAUNCH(kernel, grid, ...)                                                     \
  do {                                                                                    \
    const double _ht0 = host_trace_begin(c);                                              \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, c->stream, __VA_ARGS__);      \
    c->n_launches++;                                                                      \
    host_trace_end(c, _ht0);                                                              \
    PO_HIP(hipGetLastError());                                                            \
  } while (0)

static int wgrid(Ctx *c, int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > (int64_t)c->num_cu * 4) b = (int64_t)c->num_cu * 4;
  if (b < 1) b = 1;
  return (int)b;
}

template <int OP>
__device__ __forceinline__ double wcomb(double a, double b) {
  if (OP == 0) return a + b;
  if (OP == 1) return fmin(a, b);
  return fmax(a, b);
}
template <int N, int OP>
__device__ __forceinline__ void w_block_reduce(double (&a)[N], double *partials, int slot0, double *sm) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < N; j++) {
    double v = a[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = wcomb<OP>(v, __shfl_xor(v, o, 64));
    if (lane == 0) sm[wave * N + j] = v;
  }
  __syncthreads();
  if (threadIdx.x < N) {
    const int j = threadIdx.x;
    partials[(size_t)(slot0 + j) * gridDim.x + blockIdx.x] =
        wcomb<OP>(wcomb<OP>(sm[j], sm[N + j]), wcomb<OP>(sm[2 * N + j], sm[3 * N + j]));
  }
  __syncthreads();
}

// ---- structured Jacobian of the built-in problems --------------------------------------------------
// Tiled segmented sums.  A workgroup owns a tile of G whole groups = TV = (G-1)*period + nw consecutive
// variables (TV <= kGroupTile): lanes load the tile coalesced (lane <-> variable), products go to LDS,
// then one thread per (group, column) adds its nw LDS values and the results leave coalesced over the
// group index.  JB panel columns share one tile pass (one d load, JB independent loads in flight).
constexpr int kGroupTile = 512;  // variables per tile = 2 per thread
// sum of nw consecutive LDS values in index order (the loads of four entries are issued together, the adds keep
// the order of the plain loop: same bits)
__device__ __forceinline__ double row_sum(const double *row, int nw) {
  double sacc = 0.0;
  int k = 0;
  for (; k + 4 <= nw; k += 4) {
    const double a = row[k], b = row[k + 1], c = row[k + 2], e = row[k + 3];
    sacc += a;
    sacc += b;
    sacc += c;
    sacc += e;
  }
  for (; k < nw; k++) sacc += row[k];
  return sacc;
}
// LDS rows of the tiled kernels: group gi of a tile starts at gi * (period + pad) with pad = 1 when the period is
// even -- an odd row stride in doubles keeps the ds_read_b64 of 64 lanes that walk 64 different groups free of bank
// conflicts (stride 20 doubles = 40 banks was an 8-way conflict: round 4).  Element e of the tile (group e / period)
// therefore lives at e + pad * (e / period).
template <int JB>
__global__ void __launch_bounds__(kBlock)
    group_panel_tiled_kernel(GroupMap m, PtrTable P, int nv, const double *__restrict__ d, double alpha,
                             PtrTableW U, int G, int64_t ntiles) {
  __shared__ double sm[JB * kGroupTile];
  // the output pointers in LDS: indexing the kernel-argument table with a per-lane column is a VECTOR load from the
  // kernel-argument segment (host memory), one per batch of columns
  __shared__ double *utab[kMaxPanel];
  const int period = m.nw + m.skip;
  const int pad = (period & 1) ? 0 : 1, rstride = period + pad;
  const int tid = threadIdx.x;
  const int s0 = tid + pad * (tid / period), s1 = tid + kBlock + pad * ((tid + kBlock) / period);
  if (tid < kMaxPanel) utab[tid] = U.p[tid];
  __syncthreads();
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t g0 = tile * G;
    const int ng = (int)((m.nwcon - g0) < G ? (m.nwcon - g0) : G);
    const int64_t v0 = m.start + g0 * (int64_t)period;
    const int nvv = (ng - 1) * period + m.nw;
    const bool in0 = tid < nvv, in1 = tid + kBlock < nvv;
    // (loads are unconditional on clamped indices: every lane's requests of a batch are in flight together)
    const int64_t i0 = v0 + (in0 ? tid : 0), i1 = v0 + (in1 ? tid + kBlock : 0);
    const double d0 = d ? d[i0] : 1.0, d1 = d ? d[i1] : 1.0;
    double a0[JB], a1[JB];
#pragma unroll
    for (int u = 0; u < JB; u++) {
      const double *pj = P.p[u < nv ? u : 0];
      a0[u] = __builtin_nontemporal_load(pj + i0);
      a1[u] = __builtin_nontemporal_load(pj + i1);
    }
    for (int jb = 0; jb < nv; jb += JB) {
#pragma unroll
      for (int u = 0; u < JB; u++) {
        if (in0) sm[u * kGroupTile + s0] = d0 * a0[u];
        if (in1) sm[u * kGroupTile + s1] = d1 * a1[u];
      }
      // the next batch of columns of this tile is requested before the row sums of the current one
      if (jb + JB < nv) {
#pragma unroll
        for (int u = 0; u < JB; u++) {
          const double *pj = P.p[(jb + JB + u < nv) ? jb + JB + u : jb];
          a0[u] = __builtin_nontemporal_load(pj + i0);
          a1[u] = __builtin_nontemporal_load(pj + i1);
        }
      }
      __syncthreads();
      for (int pair = tid; pair < ng * JB; pair += kBlock) {
        const int u = pair / ng, gi = pair - u * ng;
        if (jb + u < nv) {  // (a GLOBAL store: the pointer from LDS is generic, and a flat store costs every wait its count)
          typedef __attribute__((address_space(1))) double gdouble;
          gdouble *up = (gdouble *)utab[jb + u];
          up[g0 + gi] = alpha * row_sum(sm + u * kGroupTile + gi * rstride, m.nw);
        }
      }
      __syncthreads();
    }
  }
}
// out_i = (init ? out_i : cst) + alpha * (group sum of v), same tiling with a single column
// recip != 0: out_i = 1 / (...) (the factor Cw = 1 / (Cdiag + Aw D^-1 Aw^T) of the scalar block form in one launch)
// (round 5: TWO tiles ahead.  One tile of 4 KB per workgroup and eight workgroups per CU keep 32 KB per CU in flight -- half
// of what the HBM latency asks for; the kernels ran at 3.5 / 2.2 TB/s (group_factor had no prefetch at all).  The requests
// are unconditional on clamped tile indices and the two register sets alternate instead of being copied, so that the
// compiler's wait counts stay exact: a tile is staged behind vmcnt(2), with the next tile's two loads still in flight.)
struct GroupTileRegs {
  double a0, a1;
};
__device__ __forceinline__ void group_tile_request(const GroupMap &m, const double *__restrict__ v, int G, int64_t ntiles,
                                                   int64_t tile, int tid, int period, GroupTileRegs &r) {
  const int64_t tl = tile < ntiles ? tile : ntiles - 1;
  const int64_t g0 = tl * G;
  const int ng = (int)((m.nwcon - g0) < G ? (m.nwcon - g0) : G);
  const int64_t v0 = m.start + g0 * (int64_t)period;
  const int nvv = (ng - 1) * period + m.nw;
  r.a0 = v[v0 + (tid < nvv ? tid : 0)];
  r.a1 = v[v0 + (tid + kBlock < nvv ? tid + kBlock : 0)];
}
// out_i = (init ? out_i : cst) + alpha * (group sum of v), same tiling with a single column
// recip != 0: out_i = 1 / (...) (the factor Cw = 1 / (Cdiag + Aw D^-1 Aw^T) of the scalar block form in one launch)
// FACTOR: Cw_i = 1 / (sw_i/zsw_i + tw_i/ztw_i + sum of d over group i): w_cdiag_kernel + the reciprocal form in one launch
// (same expressions, same order)
template <int FACTOR>
__global__ void __launch_bounds__(kBlock, 8)  // (the launch is 8 workgroups per CU: at 88 registers only 5 were resident)
    group_sum_tiled_kernel(GroupMap m, double *__restrict__ out, int init, double cst, double alpha,
                           const double *__restrict__ v, int G, int64_t ntiles, int recip, WVars wv) {
  __shared__ double sm[kGroupTile];
  const int period = m.nw + m.skip;
  const int pad = (period & 1) ? 0 : 1, rstride = period + pad;
  const int tid = threadIdx.x;
  const int s0 = tid + pad * (tid / period), s1 = tid + kBlock + pad * ((tid + kBlock) / period);
  const int64_t stride = gridDim.x;
  GroupTileRegs ra, rb;
  group_tile_request(m, v, G, ntiles, blockIdx.x, tid, period, ra);
  group_tile_request(m, v, G, ntiles, blockIdx.x + stride, tid, period, rb);
#define PO_GROUP_TILE(R, TILE)                                                                  \
  {                                                                                             \
    const int64_t _tile = (TILE);                                                               \
    const int64_t g0 = _tile * G;                                                               \
    const int ng = (int)((m.nwcon - g0) < G ? (m.nwcon - g0) : G);                              \
    const int nvv = (ng - 1) * period + m.nw;                                                   \
    if (tid < nvv) sm[s0] = R.a0;                                                               \
    if (tid + kBlock < nvv) sm[s1] = R.a1;                                                      \
    /* the w-sized operands of this tile's results are requested BEFORE the tile two ahead: loads complete in      \
       order, so behind it their wait would be a wait for that tile as well */                                    \
    const int64_t i = g0 + (tid < ng ? tid : 0);                                                \
    double o0 = cst, o1 = 0.0, o2 = 0.0, o3 = 0.0;                                              \
    if (FACTOR) {                                                                               \
      o0 = wv.sw[i];                                                                            \
      o1 = wv.zsw[i];                                                                           \
      o2 = wv.tw[i];                                                                            \
      o3 = wv.ztw[i];                                                                           \
    } else if (init) {                                                                          \
      o0 = out[i];                                                                              \
    }                                                                                           \
    group_tile_request(m, v, G, ntiles, _tile + 2 * stride, tid, period, R);                    \
    __syncthreads();                                                                            \
    if (tid < ng) {                                                                             \
      const double sacc = row_sum(sm + tid * rstride, m.nw);                                    \
      if (FACTOR) {                                                                             \
        const double cd = o0 / o1 + o2 / o3;                                                    \
        const double val = cd + 1.0 * sacc;                                                     \
        out[i] = 1.0 / val;                                                                     \
      } else {                                                                                  \
        const double val = o0 + alpha * sacc;                                                   \
        out[i] = recip ? 1.0 / val : val;                                                       \
      }                                                                                         \
    }                                                                                           \
    __syncthreads();                                                                            \
  }
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += 2 * stride) {
    PO_GROUP_TILE(ra, tile)
    if (tile + stride < ntiles) PO_GROUP_TILE(rb, tile + stride)
  }
#undef PO_GROUP_TILE
}
static bool group_tiling(const GroupMap &m, int *G, int64_t *ntiles);
int k_group_factor(Ctx *c, const GroupMap &m, const WVars &v, const double *d, double *cw) {
  if (m.nwcon <= 0) return PO_OK;
  int G = 0;
  int64_t ntiles = 0;
  if (!group_tiling(m, &G, &ntiles)) {
    PO_TRY(k_w_cdiag(c, v, m.nwcon, cw));
    return k_group_sum(c, m, cw, 1, 0.0, 1.0, d, 1);
  }
  count_bytes(c, 1.0, m.nwcon * (int64_t)m.nw);
  count_bytes(c, 5.0, m.nwcon);
  int64_t grid = ntiles < (int64_t)c->num_cu * 8 ? ntiles : (int64_t)c->num_cu * 8;
  PO_WLAUNCH(group_sum_tiled_kernel<1>, (int)grid, m, cw, 0, 0.0, 1.0, d, G, ntiles, 1, v);
  return PO_OK;
}

// G whole groups per tile with the padded row layout inside kGroupTile doubles: (G-1)*(period+pad) + nw <= kGroupTile
static bool group_tiling(const GroupMap &m, int *G, int64_t *ntiles) {
  const int64_t period = (int64_t)m.nw + m.skip;
  if (m.nw > kGroupTile || period <= 0 || period > kGroupTile) return false;
  const int64_t rstride = period + ((period & 1) ? 0 : 1);
  int64_t g = (kGroupTile - m.nw) / rstride + 1;
  if (g > kBlock) g = kBlock;
  if (g < 1) return false;
  *G = (int)g;
  *ntiles = (m.nwcon + g - 1) / g;
  return true;
}

// out_i = (init ? out_i : cst) + alpha * sum_{k<nw} v[start + i*(nw+skip) + k]
__global__ void __launch_bounds__(kBlock)
    group_sum_kernel(GroupMap m, double *__restrict__ out, int init, double cst, double alpha,
                     const double *__restrict__ v, int recip) {
  PO_W_LOOP(i, m.nwcon) {
    const int64_t j0 = m.start + i * (int64_t)(m.nw + m.skip);
    double s = 0.0;
    for (int k = 0; k < m.nw; k++) s += v[j0 + k];
    const double val = (init ? out[i] : cst) + alpha * s;
    out[i] = recip ? 1.0 / val : val;
  }
}
int k_group_sum(Ctx *c, const GroupMap &m, double *out, int init, double cst, double alpha,
                const double *v, int recip) {
  if (m.nwcon <= 0) return PO_OK;
  count_bytes(c, 1.0, m.nwcon * (int64_t)m.nw);  // the grouped part of v (the w-sized output is noise beside it)
  count_bytes(c, init ? 2.0 : 1.0, m.nwcon);
  int G = 0;
  int64_t ntiles = 0;
  if (group_tiling(m, &G, &ntiles)) {
    int64_t grid = ntiles < (int64_t)c->num_cu * 8 ? ntiles : (int64_t)c->num_cu * 8;
    PO_WLAUNCH(group_sum_tiled_kernel<0>, (int)grid, m, out, init, cst, alpha, v, G, ntiles, recip, WVars());
    return PO_OK;
  }
  PO_WLAUNCH(group_sum_kernel, wgrid(c, m.nwcon), m, out, init, cst, alpha, v, recip);
  return PO_OK;
}
// out[g] += alpha * w[i(g)] for every variable g that belongs to a group
__global__ void __launch_bounds__(kBlock)
    group_scatter_kernel(GroupMap m, double *__restrict__ out, double alpha, const double *__restrict__ w,
                         int64_t n) {
  const int64_t period = m.nw + m.skip;
  PO_W_LOOP(g, n) {
    const int64_t r = g - m.start;
    if (r >= 0) {
      const int64_t i = r / period;
      if (i < m.nwcon && (r - i * period) < m.nw) out[g] += alpha * w[i];
    }
  }
}
// Group index of the variables of an element PAIR (2q, 2q + 1): one 32-bit division per pair (the 64-bit division per
// element of the plain kernels costs more than the memory access it guards); valid for n < 2^31.
struct PairGroups {
  int64_t i0, i1;  // constraint of element 2q / 2q + 1, or -1
};
__device__ __forceinline__ PairGroups pair_groups(const GroupMap &m, uint32_t period, int64_t q) {
  PairGroups p;
  p.i0 = p.i1 = -1;
  const int64_t r0 = 2 * q - m.start;
  if (r0 >= 0) {
    const uint32_t i = (uint32_t)r0 / period, k = (uint32_t)r0 - i * period;
    if (k < (uint32_t)m.nw && (int64_t)i < m.nwcon) p.i0 = i;
    uint32_t i1 = i, k1 = k + 1;
    if (k1 == period) {
      i1 = i + 1;
      k1 = 0;
    }
    if (k1 < (uint32_t)m.nw && (int64_t)i1 < m.nwcon) p.i1 = i1;
  } else if (r0 == -1) {
    if (m.nw > 0 && m.nwcon > 0) p.i1 = 0;
  }
  return p;
}
__global__ void __launch_bounds__(kBlock)
    group_scatter2_kernel(GroupMap m, double *__restrict__ out, double alpha, const double *__restrict__ w, int64_t n,
                          int set) {
  const uint32_t period = (uint32_t)(m.nw + m.skip);
  const int64_t npairs = (n + 1) >> 1;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * blockDim.x) {
    const PairGroups p = pair_groups(m, period, q);
    double2 *dst = reinterpret_cast<double2 *>(out + 2 * q);
    double2 v = make_double2(0.0, 0.0);
    if (set) {  // 0 + alpha w = alpha w exactly: "zero, then add" in one pass that never reads `out`
      if (p.i0 >= 0) v.x = __dadd_rn(0.0, __dmul_rn(alpha, w[p.i0]));
      if (p.i1 >= 0) v.y = __dadd_rn(0.0, __dmul_rn(alpha, w[p.i1]));
    } else {
      v = *dst;
      if (p.i0 >= 0) v.x += alpha * w[p.i0];
      if (p.i1 >= 0) v.y += alpha * w[p.i1];
    }
    if (2 * q + 1 >= n) v.y = 0.0;  // the pad element of an odd length stays zero
    *dst = v;
  }
}
static int pair_grid(Ctx *c, int64_t n) {
  int64_t b = (((n + 1) >> 1) + kBlock - 1) / kBlock;
  if (b > (int64_t)c->num_cu * 4) b = (int64_t)c->num_cu * 4;
  if (b < 1) b = 1;
  return (int)b;
}
int k_group_scatter(Ctx *c, const GroupMap &m, double *out, double alpha, const double *w, int64_t n) {
  if (m.nwcon <= 0 || n <= 0) return PO_OK;
  count_bytes(c, 2.0, n);
  count_bytes(c, 1.0, m.nwcon);
  if (n < 2000000000LL) {
    PO_WLAUNCH(group_scatter2_kernel, pair_grid(c, n), m, out, alpha, w, n, 0);
    return PO_OK;
  }
  PO_WLAUNCH(group_scatter_kernel, wgrid(c, n), m, out, alpha, w, n);
  return PO_OK;
}
// out[g] = alpha * w[i(g)] for the variables of a group, 0 elsewhere: "fill with zero, then scatter" in one pass
// that never reads `out` (0 + alpha w = alpha w exactly)
__global__ void __launch_bounds__(kBlock)
    group_scatter_set_kernel(GroupMap m, double *__restrict__ out, double alpha, const double *__restrict__ w,
                             int64_t n) {
  const int64_t period = m.nw + m.skip;
  PO_W_LOOP(g, n) {
    const int64_t r = g - m.start;
    double v = 0.0;
    if (r >= 0) {
      const int64_t i = r / period;
      if (i < m.nwcon && (r - i * period) < m.nw) v = __dadd_rn(0.0, __dmul_rn(alpha, w[i]));
    }
    out[g] = v;
  }
}
int k_group_scatter_set(Ctx *c, const GroupMap &m, double *out, double alpha, const double *w, int64_t n) {
  if (n <= 0) return PO_OK;
  if (m.nwcon <= 0) return k_fill(c, out, n, 0.0);
  count_bytes(c, 1.0, n);
  count_bytes(c, 1.0, m.nwcon);
  if (n < 2000000000LL) {
    PO_WLAUNCH(group_scatter2_kernel, pair_grid(c, n), m, out, alpha, w, n, 1);
    return PO_OK;
  }
  PO_WLAUNCH(group_scatter_set_kernel, wgrid(c, n), m, out, alpha, w, n);
  return PO_OK;
}
// U_j[i] = alpha * sum_k d[g] * P_j[g] over the group of constraint i, for all panel columns at once
// (fallback for groups wider than a tile: one thread per constraint)
__global__ void __launch_bounds__(kBlock)
    group_panel_kernel(GroupMap m, PtrTable P, int nv, const double *__restrict__ d, double alpha,
                       PtrTableW U) {
  PO_W_LOOP(i, m.nwcon) {
    const int64_t j0 = m.start + i * (int64_t)(m.nw + m.skip);
    for (int j = 0; j < nv; j++) {
      double s = 0.0;
      for (int k = 0; k < m.nw; k++) s += d[j0 + k] * P.p[j][j0 + k];
      U.p[j][i] = alpha * s;
    }
  }
}
int k_group_panel(Ctx *c, const GroupMap &m, const double *const *P, int nv, const double *d,
                  double alpha, double *const *U) {
  if (m.nwcon <= 0 || nv <= 0) return PO_OK;
  if (nv > kMaxPanel) {  // independent columns: slabs of one kernel's pointer table
    for (int j0 = 0; j0 < nv; j0 += kMaxPanel)
      PO_TRY(k_group_panel(c, m, P + j0, nv - j0 > kMaxPanel ? kMaxPanel : nv - j0, d, alpha, U + j0));
    return PO_OK;
  }
  PtrTable pt;
  PtrTableW ut;
  for (int j = 0; j < kMaxPanel; j++) {
    pt.p[j] = j < nv ? P[j] : nullptr;
    ut.p[j] = j < nv ? U[j] : nullptr;
  }
  count_bytes(c, (double)nv + 1.0, m.nwcon * (int64_t)m.nw);
  count_bytes(c, (double)nv, m.nwcon);
  int G = 0;
  int64_t ntiles = 0;
  if (group_tiling(m, &G, &ntiles)) {
    // (workgroups per CU that are actually resident: 32 KB of LDS per workgroup in the 8-column form, 16 KB in the other)
    const int per_cu = nv > 4 ? 4 : 8;
    int64_t grid = ntiles < (int64_t)c->num_cu * per_cu ? ntiles : (int64_t)c->num_cu * per_cu;
    if (nv > 4) {
      PO_WLAUNCH(group_panel_tiled_kernel<8>, (int)grid, m, pt, nv, d, alpha, ut, G, ntiles);
    } else {
      PO_WLAUNCH(group_panel_tiled_kernel<4>, (int)grid, m, pt, nv, d, alpha, ut, G, ntiles);
    }
    return PO_OK;
  }
  PO_WLAUNCH(group_panel_kernel, wgrid(c, m.nwcon), m, pt, nv, d, alpha, ut);
  return PO_OK;
}

// second half of the structured K0^-1 apply: yx_g = d_g (bx_g + alpha * yw[i(g)]) inside a group,
// yx_g = d_g bx_g outside
__global__ void __launch_bounds__(kBlock)
    group_apply_kernel(GroupMap m, const double *__restrict__ d, const double *__restrict__ bx, double alpha,
                       const double *__restrict__ yw, int64_t n, double *__restrict__ yx) {
  const int64_t period = m.nw + m.skip;
  PO_W_LOOP(g, n) {
    double v = bx[g];
    const int64_t r = g - m.start;
    if (r >= 0) {
      const int64_t i = r / period;
      if (i < m.nwcon && (r - i * period) < m.nw) v += alpha * yw[i];
    }
    yx[g] = d[g] * v;
  }
}
int k_group_apply(Ctx *c, const GroupMap &m, const double *d, const double *bx, double alpha, const double *yw,
                  int64_t n, double *yx) {
  if (n <= 0) return PO_OK;
  count_bytes(c, 3.0, n);
  count_bytes(c, 1.0, m.nwcon);
  PO_WLAUNCH(group_apply_kernel, wgrid(c, n), m, d, bx, alpha, yw, n, yx);
  return PO_OK;
}
// The whole structured K0^-1 apply in ONE pass over (d, bx) (round 4): u_i = alpha * sum_{g in group i} d_g bx_g,
// yw_i = cw_i ((bw ? bw_i : 0) - u_i), yx_g = d_g (bx_g + alpha yw_i(g)) inside a group and d_g bx_g outside --
// group_panel_tiled_kernel on one column, w_apply_mid_kernel and group_apply_kernel with the same arithmetic in the
// same order, the tile kept in registers between the two halves.  A tile is G whole PERIODS (G * period <=
// kGroupTile), so consecutive tiles cover [start, start + nwcon * period) without gaps; what lies before `start`
// and after the last period is element-wise.
__global__ void __launch_bounds__(kBlock, 8)  // (launched 8 per CU; without the bound 106 registers = 4 resident)
    group_k0_tiled_kernel(GroupMap m, const double *__restrict__ d, const double *__restrict__ bx,
                          const double *__restrict__ cw, const double *__restrict__ bw, double alpha, int64_t n,
                          double *__restrict__ yx, double *__restrict__ yw, int G, int64_t ntiles) {
  __shared__ double sm[kGroupTile + kBlock];
  double *smw = sm + kGroupTile;
  const int period = m.nw + m.skip;
  const int tid = threadIdx.x;
  const int gi0 = tid / period, k0 = tid - gi0 * period;
  const int gi1 = (tid + kBlock) / period, k1 = tid + kBlock - gi1 * period;
  const int64_t cover_end = (m.start + m.nwcon * (int64_t)period) < n ? (m.start + m.nwcon * (int64_t)period) : n;
  // the loads of a tile (clamped: all four in flight whatever the tile's fill); the NEXT tile's are issued before
  // this tile's two barriers, so the tile loop is not one memory latency per tile
  double d0 = 0.0, d1 = 0.0, b0 = 0.0, b1 = 0.0;
  auto tile_load = [&](int64_t tile, double &e0, double &e1, double &c0, double &c1) {
    const int64_t g0 = tile * G;
    const int ng = (int)((m.nwcon - g0) < G ? (m.nwcon - g0) : G);
    const int64_t v0 = m.start + g0 * (int64_t)period;
    int64_t vend = v0 + (int64_t)ng * period;
    if (vend > n) vend = n;
    const int nvv = (int)(vend - v0);
    const int64_t i0 = v0 + (tid < nvv ? tid : 0), i1 = v0 + (tid + kBlock < nvv ? tid + kBlock : 0);
    e0 = __builtin_nontemporal_load(d + i0);
    e1 = __builtin_nontemporal_load(d + i1);
    c0 = __builtin_nontemporal_load(bx + i0);
    c1 = __builtin_nontemporal_load(bx + i1);
  };
  if ((int64_t)blockIdx.x < ntiles) tile_load(blockIdx.x, d0, d1, b0, b1);
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t g0 = tile * G;
    const int ng = (int)((m.nwcon - g0) < G ? (m.nwcon - g0) : G);
    const int64_t v0 = m.start + g0 * (int64_t)period;
    int64_t vend = v0 + (int64_t)ng * period;
    if (vend > n) vend = n;
    const int nvv = (int)(vend - v0);
    const bool in0 = tid < nvv, in1 = tid + kBlock < nvv;
    const double cd0 = d0, cd1 = d1, cb0 = b0, cb1 = b1;
    if (tile + gridDim.x < ntiles) tile_load(tile + gridDim.x, d0, d1, b0, b1);
    sm[tid] = in0 ? cd0 * cb0 : 0.0;
    sm[tid + kBlock] = in1 ? cd1 * cb1 : 0.0;
    __syncthreads();
    if (tid < ng) {
      const double u = alpha * row_sum(sm + tid * period, m.nw);
      const double y = cw[g0 + tid] * ((bw ? bw[g0 + tid] : 0.0) - u);
      yw[g0 + tid] = y;
      smw[tid] = y;
    }
    __syncthreads();
    if (in0) {
      double v = cb0;
      if (k0 < m.nw) v += alpha * smw[gi0];
      yx[v0 + tid] = cd0 * v;
    }
    if (in1) {
      double v = cb1;
      if (k1 < m.nw) v += alpha * smw[gi1];
      yx[v0 + tid + kBlock] = cd1 * v;
    }
  }
  // variables outside every group's period: yx = d bx
  const int64_t nout = m.start + (n - cover_end);
  for (int64_t i = (int64_t)blockIdx.x * kBlock + tid; i < nout; i += (int64_t)gridDim.x * kBlock) {
    const int64_t g = i < m.start ? i : cover_end + (i - m.start);
    yx[g] = d[g] * bx[g];
  }
}
int k_group_k0(Ctx *c, const GroupMap &m, const double *d, const double *bx, const double *cw, const double *bw,
               double alpha, int64_t n, double *yx, double *yw, bool *done) {
  *done = false;
  const int64_t period = (int64_t)m.nw + m.skip;
  if (m.nwcon <= 0 || n <= 0 || period <= 0 || period > kGroupTile || m.start + (m.nwcon - 1) * period + m.nw > n)
    return PO_OK;
  int64_t G = kGroupTile / period;
  if (G > kBlock) G = kBlock;
  const int64_t ntiles = (m.nwcon + G - 1) / G;
  count_bytes(c, 3.0, n);
  count_bytes(c, bw ? 3.0 : 2.0, m.nwcon);
  int64_t grid = ntiles < (int64_t)c->num_cu * 8 ? ntiles : (int64_t)c->num_cu * 8;
  PO_WLAUNCH(group_k0_tiled_kernel, (int)grid, m, d, bx, cw, bw, alpha, n, yx, yw, (int)G, ntiles);
  *done = true;
  return PO_OK;
}

// out = -cw o (sum_j alpha_j U_j) (the second K0^-1 apply of a bordered solve as a correction from the panel image,
// Problem::sparseCorrection) and, with acc != nullptr, acc += out in the same pass: panel_axpy + mul + axpy
__global__ void __launch_bounds__(kBlock)
    w_correction_kernel(PtrTable U, CoefTable a, int nv, const double *__restrict__ cw, int64_t w,
                        double *__restrict__ out, double *__restrict__ acc) {
  PO_W_LOOP(i, w) {
    double s = 0.0;
    int j = 0;
    for (; j + 4 <= nv; j += 4) {  // the loads of four columns together, the sum in column order
      const double u0 = U.p[j][i], u1 = U.p[j + 1][i], u2 = U.p[j + 2][i], u3 = U.p[j + 3][i];
      s = fma(a.a[j], u0, s);
      s = fma(a.a[j + 1], u1, s);
      s = fma(a.a[j + 2], u2, s);
      s = fma(a.a[j + 3], u3, s);
    }
    for (; j < nv; j++) s = fma(a.a[j], U.p[j][i], s);
    const double o = -1.0 * cw[i] * s;
    out[i] = o;
    if (acc) acc[i] = fma(1.0, o, acc[i]);
  }
}
int k_w_correction(Ctx *c, const double *const *U, int nv, const double *alpha, const double *cw, int64_t w,
                   double *out, double *acc) {
  if (w <= 0) return PO_OK;
  if (nv > kMaxPanel) return PO_ERR_ARG;  // (callers fall back to the separate launches)
  count_bytes(c, (double)nv + 2.0 + (acc ? 2.0 : 0.0), w);
  PtrTable pt;
  CoefTable ct;
  for (int j = 0; j < kMaxPanel; j++) {
    pt.p[j] = j < nv ? U[j] : nullptr;
    ct.a[j] = j < nv ? alpha[j] : 0.0;
  }
  PO_WLAUNCH(w_correction_kernel, wgrid(c, w), pt, ct, nv, cw, w, out, acc);
  return PO_OK;
}

// yw = cw * (bw - u)   (bw may be null)
__global__ void __launch_bounds__(kBlock)
    w_apply_mid_kernel(const double *__restrict__ cw, const double *__restrict__ bw, const double *__restrict__ u,
                       int64_t w, double *__restrict__ yw) {
  PO_W_LOOP(i, w) yw[i] = cw[i] * ((bw ? bw[i] : 0.0) - u[i]);
}
int k_w_apply_mid(Ctx *c, const double *cw, const double *bw, const double *u, int64_t w, double *yw) {
  if (w <= 0) return PO_OK;
  count_bytes(c, bw ? 4.0 : 3.0, w);
  PO_WLAUNCH(w_apply_mid_kernel, wgrid(c, w), cw, bw, u, w, yw);
  return PO_OK;
}

// ---- nwblock > 1: block-diagonal Cw with dense nwblock x nwblock blocks (ParOptQuasiDefBlockMat, reference
// src/ParOptSparseMat.cpp:41-229), LAPACK's packed upper storage per block: (i, j), i <= j, at i + j (j+1)/2.
// One thread per block (the blocks are a handful of entries); kMaxBlock bounds the unrolled local arrays.
constexpr int kMaxBlock = 16;
// blocks <- diag(cdiag) (:60-76), before the problem's addSparseInnerProduct adds Aw D^-1 Aw^T
__global__ void __launch_bounds__(kBlock)
    blk_init_kernel(const double *__restrict__ cdiag, int64_t nblocks, int B, double *__restrict__ blk) {
  const int incr = B * (B + 1) / 2;
  PO_W_LOOP(b, nblocks) {
    double *a = blk + b * incr;
    for (int t = 0; t < incr; t++) a[t] = 0.0;
    for (int j = 0; j < B; j++) a[j + j * (j + 1) / 2] = cdiag[b * B + j];
  }
}
int k_blk_init(Ctx *c, const double *cdiag, int64_t nblocks, int B, double *blk) {
  if (nblocks <= 0) return PO_OK;
  PO_WLAUNCH(blk_init_kernel, wgrid(c, nblocks), cdiag, nblocks, B, blk);
  return PO_OK;
}
// A = U^T U in place (dpptrf "U", :104-112)
__global__ void __launch_bounds__(kBlock) blk_factor_kernel(double *__restrict__ blk, int64_t nblocks, int B, int *flag) {
  const int incr = B * (B + 1) / 2;
  PO_W_LOOP(b, nblocks) {
    double *a = blk + b * incr;
    for (int j = 0; j < B; j++) {
      const int cj = j * (j + 1) / 2;
      for (int i = 0; i < j; i++) {
        const int ci = i * (i + 1) / 2;
        double v = a[i + cj];
        for (int k = 0; k < i; k++) v -= a[k + ci] * a[k + cj];
        a[i + cj] = v / a[i + ci];
      }
      double d = a[j + cj];
      for (int k = 0; k < j; k++) d -= a[k + cj] * a[k + cj];
      if (!(d > 0.0)) {
        // flag[0]: number of non-positive pivots of this factorization; flag[1]: the FIRST failing row (flag[1]
        // starts at INT_MAX); the pivot is replaced by 1 so that the solves stay finite -- dpptrf stops here and
        // the reference ignores its info (src/ParOptSparseMat.cpp:104-112, ParOptInteriorPoint.cpp:1930)
        atomicAdd(&flag[0], 1);
        atomicMin(&flag[1], (int)(b * B + j));
        d = 1.0;
      }
      a[j + cj] = sqrt(d);
    }
  }
}
int k_blk_factor(Ctx *c, double *blk, int64_t nblocks, int B, int *flag) {
  if (nblocks <= 0) return PO_OK;
  PO_WLAUNCH(blk_factor_kernel, wgrid(c, nblocks), blk, nblocks, B, flag);
  return PO_OK;
}
// mode 0: y <- (U^T U)^-1 y (dpptrs, :199-216); mode 1: y <- U^-T y; mode 2: y <- U^-1 y.  One thread per
// (block, right-hand side).
__global__ void __launch_bounds__(kBlock)
    blk_solve_kernel(const double *__restrict__ blk, int64_t nblocks, int B, PtrTableW Y, int nv, int mode) {
  const int incr = B * (B + 1) / 2;
  PO_W_LOOP(t, nblocks * nv) {
    const int r = (int)(t / nblocks);
    const int64_t b = t - (int64_t)r * nblocks;
    const double *a = blk + b * incr;
    double *y = Y.p[r] + b * B;
    double v[kMaxBlock];
    for (int j = 0; j < B; j++) v[j] = y[j];
    if (mode != 2) {  // U^T z = y, forward
      for (int j = 0; j < B; j++) {
        const int cj = j * (j + 1) / 2;
        double s = v[j];
        for (int i = 0; i < j; i++) s -= a[i + cj] * v[i];
        v[j] = s / a[j + cj];
      }
    }
    if (mode != 1) {  // U x = z, backward
      for (int j = B - 1; j >= 0; j--) {
        double s = v[j];
        for (int k = j + 1; k < B; k++) s -= a[j + k * (k + 1) / 2] * v[k];
        v[j] = s / a[j + j * (j + 1) / 2];
      }
    }
    for (int j = 0; j < B; j++) y[j] = v[j];
  }
}
int k_blk_solve(Ctx *c, const double *blk, int64_t nblocks, int B, double *const *Y, int nv, int mode) {
  if (nblocks <= 0 || nv <= 0) return PO_OK;
  if (B > kMaxBlock) {
    set_error("block solve: nwblock %d exceeds %d", B, kMaxBlock);
    return PO_ERR_ARG;
  }
  if (nv > kMaxPanel) {  // independent right-hand sides: slabs
    for (int j0 = 0; j0 < nv; j0 += kMaxPanel)
      PO_TRY(k_blk_solve(c, blk, nblocks, B, Y + j0, nv - j0 > kMaxPanel ? kMaxPanel : nv - j0, mode));
    return PO_OK;
  }
  PtrTableW t;
  for (int j = 0; j < kMaxPanel; j++) t.p[j] = j < nv ? Y[j] : Y[0];
  PO_WLAUNCH(blk_solve_kernel, wgrid(c, nblocks * nv), blk, nblocks, B, t, nv, mode);
  return PO_OK;
}

// ---- small element-wise helpers ---------------------------------------------------------------------
// y = a * x1 * x2   (x2 may be null -> y = a * x1)
__global__ void __launch_bounds__(kBlock)
    mul_kernel(double *y, double a, const double *x1, const double *x2, int64_t n) {  // y may alias
  PO_W_LOOP(i, n) y[i] = x2 ? a * x1[i] * x2[i] : a * x1[i];
}
int k_mul(Ctx *c, double *y, double a, const double *x1, const double *x2, int64_t n) {
  if (n <= 0) return PO_OK;
  count_bytes(c, 3.0, n);
  PO_WLAUNCH(mul_kernel, wgrid(c, n), y, a, x1, x2, n);
  return PO_OK;
}
// Cw = 1 / Cw
__global__ void __launch_bounds__(kBlock) recip_kernel(double *__restrict__ y, int64_t n) {
  PO_W_LOOP(i, n) y[i] = 1.0 / y[i];
}
int k_recip(Ctx *c, double *y, int64_t n) {
  if (n <= 0) return PO_OK;
  count_bytes(c, 2.0, n);
  PO_WLAUNCH(recip_kernel, wgrid(c, n), y, n);
  return PO_OK;
}

// ---- residual blocks of the sparse constraints (computeKKTRes :1362-1398) ---------------------------
// in: rzw holds cw(x).  out: rzw = -(cw - sw + tw), rsw = zsw - gsw - zw, rtw = ztw - gtw + zw,
// rzsw = mu - sw zsw, rztw = mu - tw ztw.
// sums {sw.zsw + tw.ztw, l1 rzw, l2^2 rzw, l1 rsw, l1 rtw, l1 rzsw, l1 rztw}; maxs {rzw, rsw, rtw, rzsw, rztw}
__global__ void __launch_bounds__(kBlock)
    w_res_kernel(WVars v, WVars r, const double *__restrict__ gsw, const double *__restrict__ gtw,
                 const double *cw, double mu, int64_t w, double *__restrict__ d2out, double *__restrict__ partials) {
  __shared__ double sm[4 * 7];
  double sums[7] = {0, 0, 0, 0, 0, 0, 0};
  double maxs[5] = {0, 0, 0, 0, 0};
  PO_W_LOOP(i, w) {
    const double sw = v.sw[i], tw = v.tw[i], zw = v.zw[i], zsw = v.zsw[i], ztw = v.ztw[i];
    const double a = -(cw[i] - sw + tw);  // cw may be r.zw itself (the constraint values left there by the caller)
    const double b = zsw - gsw[i] - zw;
    const double cc = ztw - gtw[i] + zw;
    const double d = mu - sw * zsw;
    const double e = mu - tw * ztw;
    r.zw[i] = a;
    r.sw[i] = b;
    r.tw[i] = cc;
    r.zsw[i] = d;
    r.ztw[i] = e;
    // d2 of the block solve that follows, from the blocks just formed (w_d2_kernel's expression: same bits)
    if (d2out) d2out[i] = a + (d + sw * b) / zsw - (e + tw * cc) / ztw;
    sums[0] += sw * zsw + tw * ztw;
    sums[1] += fabs(a);
    sums[2] += a * a;
    sums[3] += fabs(b);
    sums[4] += fabs(cc);
    sums[5] += fabs(d);
    sums[6] += fabs(e);
    maxs[0] = fmax(maxs[0], fabs(a));
    maxs[1] = fmax(maxs[1], fabs(b));
    maxs[2] = fmax(maxs[2], fabs(cc));
    maxs[3] = fmax(maxs[3], fabs(d));
    maxs[4] = fmax(maxs[4], fabs(e));
  }
  w_block_reduce<7, 0>(sums, partials, 0, sm);
  w_block_reduce<5, 2>(maxs, partials, 7, sm);
}
int k_w_res(Ctx *c, const WVars &v, const WVars &r, const double *gsw, const double *gtw, double mu,
            int64_t w, double out[12], const double *cw, double *d2out) {
  count_bytes(c, d2out ? 13.0 : 12.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid * 12));
  PO_WLAUNCH(w_res_kernel, grid, v, r, gsw, gtw, cw ? cw : r.zw, mu, w, d2out, c->d_partials);
  if (!out) return PO_OK;  // the residual blocks only
  return reduce_finish(c, grid, 7, 0, 5, out);
}

// dst blocks = alpha * src blocks (the alpha-scaled right-hand side of solveKKTDiagSystem :2441-2614)
__global__ void __launch_bounds__(kBlock) w_scale5_kernel(WVars dst, WVars src, double alpha, int64_t w) {
  PO_W_LOOP(i, w) {
    dst.zw[i] = alpha * src.zw[i];
    dst.sw[i] = alpha * src.sw[i];
    dst.tw[i] = alpha * src.tw[i];
    dst.zsw[i] = alpha * src.zsw[i];
    dst.ztw[i] = alpha * src.ztw[i];
  }
}
int k_w_scale5(Ctx *c, const WVars &dst, const WVars &src, double alpha, int64_t w) {
  count_bytes(c, 10.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_scale5_kernel, wgrid(c, w), dst, src, alpha, w);
  return PO_OK;
}
// sums of squares of the five residual blocks (the sparse part of |b| in computeKKTGMRESStep :5846-5852)
__global__ void __launch_bounds__(kBlock) w_sumsq5_kernel(WVars r, int64_t w, double *__restrict__ partials) {
  __shared__ double sm[4 * 5];
  double s[5] = {0, 0, 0, 0, 0};
  PO_W_LOOP(i, w) {
    s[0] += r.zw[i] * r.zw[i];
    s[1] += r.sw[i] * r.sw[i];
    s[2] += r.tw[i] * r.tw[i];
    s[3] += r.zsw[i] * r.zsw[i];
    s[4] += r.ztw[i] * r.ztw[i];
  }
  w_block_reduce<5, 0>(s, partials, 0, sm);
}
int k_w_sumsq5(Ctx *c, const WVars &r, int64_t w, double out[5]) {
  count_bytes(c, 5.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid * 5));
  PO_WLAUNCH(w_sumsq5_kernel, grid, r, w, c->d_partials);
  return reduce_finish(c, grid, 5, 0, 0, out);
}

// Cdiag = sw/zsw + tw/ztw (setUpKKTDiagSystem :1912-1927)
__global__ void __launch_bounds__(kBlock) w_cdiag_kernel(WVars v, int64_t w, double *__restrict__ cd) {
  PO_W_LOOP(i, w) cd[i] = v.sw[i] / v.zsw[i] + v.tw[i] / v.ztw[i];
}
int k_w_cdiag(Ctx *c, const WVars &v, int64_t w, double *cd) {
  count_bytes(c, 5.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_cdiag_kernel, wgrid(c, w), v, w, cd);
  return PO_OK;
}

// d2 = b.zw + (b.zsw + sw b.sw)/zsw - (b.ztw + tw b.tw)/ztw (:2111-2136)
__global__ void __launch_bounds__(kBlock)
    w_d2_kernel(WVars v, WVars b, int64_t w, double *__restrict__ d2) {
  PO_W_LOOP(i, w) {
    d2[i] = b.zw[i] + (b.zsw[i] + v.sw[i] * b.sw[i]) / v.zsw[i] - (b.ztw[i] + v.tw[i] * b.tw[i]) / v.ztw[i];
  }
}
int k_w_d2(Ctx *c, const WVars &v, const WVars &b, int64_t w, double *d2) {
  count_bytes(c, 10.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_d2_kernel, wgrid(c, w), v, b, w, d2);
  return PO_OK;
}

// step blocks (:2180-2208), y.zw = dzw given; refine accumulates into p.
// out mins {max_x over sw, tw ; max_z over zsw, ztw} with fraction tau (computeMaxStep :3017-3061)
// comp != 0: also {S10, S01, S11} of the step written here -- the complementarity of the sparse slacks at step lengths
// (ax, az) is sum (sw + ax psw)(zsw + az pzsw) + (tw + ax ptw)(ztw + az pztw) = S00 + ax S10 + az S01 + ax az S11 with
// S00 = sum sw zsw + tw ztw of the iterate (w_res_kernel's first sum): computeCompStep's sparse part (:2866-2889)
// without a pass of its own and without knowing the step lengths (as solve2r_kernel does for the bound terms)
__global__ void __launch_bounds__(kBlock)
    w_step_kernel(WVars v, WVars b, const double *__restrict__ dzw, int refine, double tau, WVars p,
                  int64_t w, int comp, double *__restrict__ partials) {
  __shared__ double sm[4 * 3];
  double mins[2] = {1.0, 1.0};
  double cs[3] = {0.0, 0.0, 0.0};
  PO_W_LOOP(i, w) {
    const double yzw = dzw[i];
    const double yzsw = yzw - b.sw[i];
    const double yztw = -b.tw[i] - yzw;
    const double ysw = (b.zsw[i] - v.sw[i] * yzsw) / v.zsw[i];
    const double ytw = (b.ztw[i] - v.tw[i] * yztw) / v.ztw[i];
    double q0 = yzw, q1 = ysw, q2 = ytw, q3 = yzsw, q4 = yztw;
    if (refine) {
      q0 += p.zw[i];
      q1 += p.sw[i];
      q2 += p.tw[i];
      q3 += p.zsw[i];
      q4 += p.ztw[i];
    }
    p.zw[i] = q0;
    p.sw[i] = q1;
    p.tw[i] = q2;
    p.zsw[i] = q3;
    p.ztw[i] = q4;
    if (q1 < 0.0) mins[0] = fmin(mins[0], -tau * v.sw[i] / q1);
    if (q2 < 0.0) mins[0] = fmin(mins[0], -tau * v.tw[i] / q2);
    if (q3 < 0.0) mins[1] = fmin(mins[1], -tau * v.zsw[i] / q3);
    if (q4 < 0.0) mins[1] = fmin(mins[1], -tau * v.ztw[i] / q4);
    if (comp) {
      cs[0] += q1 * v.zsw[i] + q2 * v.ztw[i];
      cs[1] += v.sw[i] * q3 + v.tw[i] * q4;
      cs[2] += q1 * q3 + q2 * q4;
    }
  }
  if (comp) {
    w_block_reduce<3, 0>(cs, partials, 0, sm);
    w_block_reduce<2, 1>(mins, partials, 3, sm);
  } else {
    w_block_reduce<2, 1>(mins, partials, 0, sm);
  }
}
// out = {min_x, min_z}, or with comp {S10, S01, S11, min_x, min_z}
int k_w_step(Ctx *c, const WVars &v, const WVars &b, const double *dzw, int refine, double tau,
             const WVars &p, int64_t w, double *out, int comp) {
  count_bytes(c, refine ? 20.0 : 15.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid * 5));
  PO_WLAUNCH(w_step_kernel, grid, v, b, dzw, refine, tau, p, w, comp, c->d_partials);
  return reduce_finish(c, grid, comp ? 3 : 0, 2, 0, out);
}

// addKKTResStep, w blocks (:1498-1527); r.zw already holds r.zw - Aw px.
__global__ void __launch_bounds__(kBlock)
    w_res_step_kernel(WVars v, WVars p, WVars r, int64_t w, double *__restrict__ d2out) {
  PO_W_LOOP(i, w) {
    const double a = r.zw[i] + (p.sw[i] - p.tw[i]);
    const double b = r.sw[i] + (p.zsw[i] - p.zw[i]);
    const double cc = r.tw[i] + (p.ztw[i] + p.zw[i]);
    const double d = r.zsw[i] - (p.sw[i] * v.zsw[i] + v.sw[i] * p.zsw[i]);
    const double e = r.ztw[i] - (p.tw[i] * v.ztw[i] + v.tw[i] * p.ztw[i]);
    r.zw[i] = a;
    r.sw[i] = b;
    r.tw[i] = cc;
    r.zsw[i] = d;
    r.ztw[i] = e;
    // d2 of the refinement's block solve from the updated blocks (w_d2_kernel's expression)
    if (d2out) d2out[i] = a + (d + v.sw[i] * b) / v.zsw[i] - (e + v.tw[i] * cc) / v.ztw[i];
  }
}
int k_w_res_step(Ctx *c, const WVars &v, const WVars &p, const WVars &r, int64_t w, double *d2out) {
  count_bytes(c, d2out ? 16.0 : 15.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_res_step_kernel, wgrid(c, w), v, p, r, w, d2out);
  return PO_OK;
}
// Mehrotra corrector (:1733-1750): r.zsw -= psw pzsw, r.ztw -= ptw pztw
__global__ void __launch_bounds__(kBlock) w_corrector_kernel(WVars p, WVars r, int64_t w) {
  PO_W_LOOP(i, w) {
    r.zsw[i] -= p.sw[i] * p.zsw[i];
    r.ztw[i] -= p.tw[i] * p.ztw[i];
  }
}
int k_w_corrector(Ctx *c, const WVars &p, const WVars &r, int64_t w) {
  count_bytes(c, 10.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_corrector_kernel, wgrid(c, w), p, r, w);
  return PO_OK;
}

// computeCompStep, w part (:2866-2889): sum (sw + ax psw)(zsw + az pzsw) + (tw + ax ptw)(ztw + az pztw)
__global__ void __launch_bounds__(kBlock)
    w_comp_step_kernel(WVars v, WVars p, double ax, double az, int64_t w, double *__restrict__ partials) {
  __shared__ double sm[4];
  double s[1] = {0.0};
  PO_W_LOOP(i, w) {
    s[0] += (v.sw[i] + ax * p.sw[i]) * (v.zsw[i] + az * p.zsw[i]) +
            (v.tw[i] + ax * p.tw[i]) * (v.ztw[i] + az * p.ztw[i]);
  }
  w_block_reduce<1, 0>(s, partials, 0, sm);
}
int k_w_comp_step(Ctx *c, const WVars &v, const WVars &p, double ax, double az, int64_t w, double *out) {
  count_bytes(c, 10.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid));
  PO_WLAUNCH(w_comp_step_kernel, grid, v, p, ax, az, w, c->d_partials);
  return reduce_finish(c, grid, 1, 0, 0, out);
}

// evalMeritInitDeriv, w part (:3735-3765, 3489-3503). cw = cw(x), awpx = Aw px (unscaled).
// out {pos log, neg log, pos presult, neg presult, gsw.sw, gtw.tw, gsw.psw, gtw.ptw, |rw1|^2, rw1.rw2}
__global__ void __launch_bounds__(kBlock)
    w_merit_kernel(WVars v, WVars p, double sx, const double *__restrict__ gsw,
                   const double *__restrict__ gtw, const double *__restrict__ cw,
                   const double *__restrict__ awpx, int64_t w, double *__restrict__ partials) {
  __shared__ double sm[4 * 10];
  double s[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  PO_W_LOOP(i, w) {
    const double sw = v.sw[i], tw = v.tw[i], psw = sx * p.sw[i], ptw = sx * p.tw[i];
    const double ls = log(sw), lt = log(tw);
    if (sw > 1.0) s[0] += ls; else s[1] += ls;
    if (tw > 1.0) s[0] += lt; else s[1] += lt;
    if (psw > 0.0) s[2] += psw / sw; else s[3] += psw / sw;
    if (ptw > 0.0) s[2] += ptw / tw; else s[3] += ptw / tw;
    s[4] += gsw[i] * sw;
    s[5] += gtw[i] * tw;
    s[6] += gsw[i] * psw;
    s[7] += gtw[i] * ptw;
    const double r1 = cw[i] - sw + tw;
    const double r2 = sx * awpx[i] - psw + ptw;
    s[8] += r1 * r1;
    s[9] += r1 * r2;
  }
  w_block_reduce<10, 0>(s, partials, 0, sm);
}
int k_w_merit(Ctx *c, const WVars &v, const WVars &p, double sx, const double *gsw, const double *gtw,
              const double *cw, const double *awpx, int64_t w, double out[10]) {
  count_bytes(c, 12.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid * 10));
  PO_WLAUNCH(w_merit_kernel, grid, v, p, sx, gsw, gtw, cw, awpx, w, c->d_partials);
  return reduce_finish(c, grid, 10, 0, 0, out);
}

// line-search trial, w part (:4003-4008 + evalMeritFunc :3572-3590, 3438-3462): rsw = max(sw + a psw, eps) ...
// out {pos log, neg log, |cw(xt) - rsw + rtw|^2, gsw.rsw, gtw.rtw}
__global__ void __launch_bounds__(kBlock)
    w_trial_kernel(WVars v, WVars p, double a, double eps, const double *__restrict__ gsw,
                   const double *__restrict__ gtw, const double *__restrict__ cwt, int64_t w,
                   double *__restrict__ partials) {
  __shared__ double sm[4 * 5];
  double s[5] = {0, 0, 0, 0, 0};
  PO_W_LOOP(i, w) {
    double rs = v.sw[i] + a * p.sw[i];
    if (rs <= eps) rs = eps;
    double rt = v.tw[i] + a * p.tw[i];
    if (rt <= eps) rt = eps;
    const double ls = log(rs), lt = log(rt);
    if (rs > 1.0) s[0] += ls; else s[1] += ls;
    if (rt > 1.0) s[0] += lt; else s[1] += lt;
    const double r1 = cwt[i] - rs + rt;
    s[2] += r1 * r1;
    s[3] += gsw[i] * rs;
    s[4] += gtw[i] * rt;
  }
  w_block_reduce<5, 0>(s, partials, 0, sm);
}
int k_w_trial(Ctx *c, const WVars &v, const WVars &p, double a, double eps, const double *gsw,
              const double *gtw, const double *cwt, int64_t w, double out[5]) {
  count_bytes(c, 14.0, w);  // w-sized block vectors touched
  const int grid = wgrid(c, w);
  PO_TRY(ensure_partials(c, (size_t)grid * 5));
  PO_WLAUNCH(w_trial_kernel, grid, v, p, a, eps, gsw, gtw, cwt, w, c->d_partials);
  return reduce_finish(c, grid, 5, 0, 0, out);
}

// computeStepAndUpdate, w part (:4177-4183): primal slacks with ax-scaled step, multipliers with az
__global__ void __launch_bounds__(kBlock)
    w_update_kernel(WVars v, WVars p, double ax, double az, double eps, int64_t w) {
  PO_W_LOOP(i, w) {
    double q = v.sw[i] + ax * p.sw[i];
    v.sw[i] = q <= eps ? eps : q;
    q = v.tw[i] + ax * p.tw[i];
    v.tw[i] = q <= eps ? eps : q;
    v.zw[i] = v.zw[i] + az * p.zw[i];
    q = v.zsw[i] + az * p.zsw[i];
    v.zsw[i] = q <= eps ? eps : q;
    q = v.ztw[i] + az * p.ztw[i];
    v.ztw[i] = q <= eps ? eps : q;
  }
}
int k_w_update(Ctx *c, const WVars &v, const WVars &p, double ax, double az, double eps, int64_t w) {
  count_bytes(c, 15.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_update_kernel, wgrid(c, w), v, p, ax, az, eps, w);
  return PO_OK;
}

// initAffineStepMultipliers, w part (:5601-5628)
__global__ void __launch_bounds__(kBlock) w_affine_kernel(WVars v, WVars p, double amin, int64_t w) {
  PO_W_LOOP(i, w) {
    v.zw[i] = v.zw[i] + p.zw[i];
    v.sw[i] = fmax(amin, fabs(v.sw[i] + p.sw[i]));
    v.tw[i] = fmax(amin, fabs(v.tw[i] + p.tw[i]));
    v.zsw[i] = fmax(amin, fabs(v.zsw[i] + p.zsw[i]));
    v.ztw[i] = fmax(amin, fabs(v.ztw[i] + p.ztw[i]));
  }
}
int k_w_affine(Ctx *c, const WVars &v, const WVars &p, double amin, int64_t w) {
  count_bytes(c, 10.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_affine_kernel, wgrid(c, w), v, p, amin, w);
  return PO_OK;
}

// least-squares multiplier clip (:5522-5533): zw = |zw| > 10 max(gsw, gtw) ? 0 : zw
__global__ void __launch_bounds__(kBlock)
    w_clip_kernel(double *__restrict__ zw, const double *__restrict__ src, const double *__restrict__ gsw,
                  const double *__restrict__ gtw, int64_t w) {
  PO_W_LOOP(i, w) {
    const double gam = 10.0 * fmax(gsw[i], gtw[i]);
    const double z = src[i];
    zw[i] = (z < -gam || z > gam) ? 0.0 : z;
  }
}
int k_w_clip(Ctx *c, double *zw, const double *src, const double *gsw, const double *gtw, int64_t w) {
  count_bytes(c, 4.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_clip_kernel, wgrid(c, w), zw, src, gsw, gtw, w);
  return PO_OK;
}

// penalty vectors (:361-374): gsw_i = i < nwinequality ? 0 : gamma ; gtw = gamma  (global index)
__global__ void __launch_bounds__(kBlock)
    w_gamma_kernel(double *__restrict__ gsw, double *__restrict__ gtw, double gamma, int64_t nwineq,
                   int64_t w) {
  PO_W_LOOP(i, w) {
    gsw[i] = i < nwineq ? 0.0 : gamma;
    gtw[i] = gamma;
  }
}
int k_w_gamma(Ctx *c, double *gsw, double *gtw, double gamma, int64_t nwineq, int64_t w) {
  count_bytes(c, 2.0, w);  // w-sized block vectors touched
  if (w <= 0) return PO_OK;
  PO_WLAUNCH(w_gamma_kernel, wgrid(c, w), gsw, gtw, gamma, nwineq, w);
  return PO_OK;
}

}  // namespace po
