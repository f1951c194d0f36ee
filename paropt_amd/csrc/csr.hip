// Device kernels of the general sparse-constraint path (csr.hpp): CSR / CSC products with the fixed
// pattern of ParOptSparseProblem (reference src/ParOptProblem.cpp:762-816), the Schur complement
// S = C + Aw D^-1 Aw^T of ParOptQuasiDefSparseMat::factor (src/ParOptSparseMat.cpp:303-356) assembled
// straight into the factor's value array, and a level-scheduled sparse Cholesky with its triangular
// solves.  Everything is gather-based and deterministic: each output element has exactly one writer
// and a fixed summation order, so repeated runs are bit-identical.
//
// Rooflines: the products and the assembly are HBM/L2-gather bound (8-byte gathers over a fixed
// pattern); the factorization and the solves are LATENCY bound by the number of dependency levels
// (one launch per level, one wavefront per row) - the nested-dissection ordering on the host keeps
// that number near log2(w) + the separator sizes.
#include <math.h>

#include "csr.hpp"
#include "wcon.hpp"

namespace po {

#define PO_C_LOOP(i, n)                                                                   \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);               \
       i += (int64_t)gridDim.x * blockDim.x)

#define PO_CLAUNCH(kernel, grid, block, ...)                                              \
  do {                                                                                    \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, c->stream, __VA_ARGS__);       \
    c->n_launches++;                                                                      \
    PO_HIP(hipGetLastError());                                                            \
  } while (0)

static int cgrid(Ctx *c, int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  if (b > (int64_t)c->num_cu * 8) b = (int64_t)c->num_cu * 8;
  if (b < 1) b = 1;
  return (int)b;
}

template <int G>
__device__ __forceinline__ double group_sum(double v) {
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// dst[i] = src[idx[i]]
__global__ void __launch_bounds__(kBlock)
    csr_gather_kernel(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ idx,
                      int64_t n) {
  PO_C_LOOP(i, n) dst[i] = src[idx[i]];
}
int k_csr_gather(Ctx *c, double *dst, const double *src, const int *idx, int64_t n) {
  if (n <= 0) return PO_OK;
  PO_CLAUNCH(csr_gather_kernel, cgrid(c, n), kBlock, dst, src, idx, n);
  return PO_OK;
}

// out[op(i)] = beta*b[i] + alpha * sum_q vals[q] * scale[c_q] * x[c_q]; G lanes share a row.
// b may alias out (no restrict on either).
template <int G>
__global__ void __launch_bounds__(kBlock)
    csr_spmv_kernel(const int *__restrict__ rowp, const int *__restrict__ cols, const double *__restrict__ vals,
                    int64_t w, double alpha, const double *__restrict__ x, const double *__restrict__ scale,
                    double beta, const double *b, double *out, const int *__restrict__ outperm) {
  const int sub = threadIdx.x % G;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / G);
  for (int64_t i = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G; i < w; i += stride) {
    double acc = 0.0;
    const int e = rowp[i + 1];
    for (int q = rowp[i] + sub; q < e; q += G) {
      const int k = cols[q];
      acc += scale ? vals[q] * (scale[k] * x[k]) : vals[q] * x[k];
    }
    acc = group_sum<G>(acc);
    if (sub == 0) {
      const double base = beta != 0.0 ? beta * b[i] : 0.0;
      out[outperm ? outperm[i] : i] = base + alpha * acc;
    }
  }
}
int k_csr_spmv(Ctx *c, int group, const int *rowp, const int *cols, const double *vals, int64_t w, double alpha,
               const double *x, const double *scale, double beta, const double *b, double *out,
               const int *outperm) {
  if (w <= 0) return PO_OK;
  const int grid = cgrid(c, w * group);
  if (group == 1) {
    PO_CLAUNCH(csr_spmv_kernel<1>, grid, kBlock, rowp, cols, vals, w, alpha, x, scale, beta, b, out, outperm);
  } else if (group == 4) {
    PO_CLAUNCH(csr_spmv_kernel<4>, grid, kBlock, rowp, cols, vals, w, alpha, x, scale, beta, b, out, outperm);
  } else if (group == 16) {
    PO_CLAUNCH(csr_spmv_kernel<16>, grid, kBlock, rowp, cols, vals, w, alpha, x, scale, beta, b, out, outperm);
  } else {
    PO_CLAUNCH(csr_spmv_kernel<64>, grid, kBlock, rowp, cols, vals, w, alpha, x, scale, beta, b, out, outperm);
  }
  return PO_OK;
}

// out[j] = dscale[j] * (bx[j] + alpha * sum_{p in column j} vals[srcT[p]] * y[rowsT[p]]); bx may alias out
template <int G>
__global__ void __launch_bounds__(kBlock)
    csr_spmvT_kernel(const int *__restrict__ colp, const int *__restrict__ rowsT, const int *__restrict__ srcT,
                     const double *__restrict__ vals, int64_t n, double alpha, const double *__restrict__ y,
                     const double *bx, const double *__restrict__ dscale, double *out) {
  const int sub = threadIdx.x % G;
  const int64_t stride = (int64_t)gridDim.x * (kBlock / G);
  for (int64_t j = (int64_t)blockIdx.x * (kBlock / G) + threadIdx.x / G; j < n; j += stride) {
    double acc = 0.0;
    const int e = colp[j + 1];
    for (int p = colp[j] + sub; p < e; p += G) acc += vals[srcT[p]] * y[rowsT[p]];
    acc = group_sum<G>(acc);
    if (sub == 0) {
      const double v = (bx ? bx[j] : 0.0) + alpha * acc;
      out[j] = dscale ? dscale[j] * v : v;
    }
  }
}
int k_csr_spmvT(Ctx *c, int group, const int *colp, const int *rowsT, const int *srcT, const double *vals,
                int64_t n, double alpha, const double *y, const double *bx, const double *dscale, double *out) {
  if (n <= 0) return PO_OK;
  const int grid = cgrid(c, n * group);
  if (group == 1) {
    PO_CLAUNCH(csr_spmvT_kernel<1>, grid, kBlock, colp, rowsT, srcT, vals, n, alpha, y, bx, dscale, out);
  } else if (group == 4) {
    PO_CLAUNCH(csr_spmvT_kernel<4>, grid, kBlock, colp, rowsT, srcT, vals, n, alpha, y, bx, dscale, out);
  } else if (group == 16) {
    PO_CLAUNCH(csr_spmvT_kernel<16>, grid, kBlock, colp, rowsT, srcT, vals, n, alpha, y, bx, dscale, out);
  } else {
    PO_CLAUNCH(csr_spmvT_kernel<64>, grid, kBlock, colp, rowsT, srcT, vals, n, alpha, y, bx, dscale, out);
  }
  return PO_OK;
}

// out[j] = scale * sum_{rows i touching column j} y[i]   (pattern only)
__global__ void __launch_bounds__(kBlock)
    csr_colsum_kernel(const int *__restrict__ colp, const int *__restrict__ rowsT, int64_t n, double scale,
                      const double *__restrict__ y, double *__restrict__ out) {
  PO_C_LOOP(j, n) {
    double acc = 0.0;
    for (int p = colp[j]; p < colp[j + 1]; p++) acc += y[rowsT[p]];
    out[j] = scale * acc;
  }
}
int k_csr_colsum(Ctx *c, const int *colp, const int *rowsT, int64_t n, double scale, const double *y,
                 double *out) {
  if (n <= 0) return PO_OK;
  PO_CLAUNCH(csr_colsum_kernel, cgrid(c, n), kBlock, colp, rowsT, n, scale, y, out);
  return PO_OK;
}

// out_i += alpha * sum_q vals[q]^2 cvec[c_q]   (the diagonal of Aw diag(cvec) Aw^T)
__global__ void __launch_bounds__(kBlock)
    csr_inner_kernel(const int *__restrict__ rowp, const int *__restrict__ cols, const double *__restrict__ vals,
                     int64_t w, double alpha, const double *__restrict__ cvec, double *__restrict__ out) {
  PO_C_LOOP(i, w) {
    double acc = 0.0;
    for (int q = rowp[i]; q < rowp[i + 1]; q++) acc += vals[q] * vals[q] * cvec[cols[q]];
    out[i] += alpha * acc;
  }
}
int k_csr_inner(Ctx *c, const int *rowp, const int *cols, const double *vals, int64_t w, double alpha,
                const double *cvec, double *out) {
  if (w <= 0) return PO_OK;
  PO_CLAUNCH(csr_inner_kernel, cgrid(c, w), kBlock, rowp, cols, vals, w, alpha, cvec, out);
  return PO_OK;
}

// U_r[op(i)] = sum_q vals[q] d[c_q] P_r[c_q], eight columns of the panel per sweep over the row
constexpr int kPanelJB = 8;
__global__ void __launch_bounds__(kBlock)
    csr_panel_kernel(const int *__restrict__ rowp, const int *__restrict__ cols, const double *__restrict__ vals,
                     int64_t w, const double *__restrict__ d, PtrTable P, int nv, PtrTableW U,
                     const int *__restrict__ outperm) {
  PO_C_LOOP(i, w) {
    const int b = rowp[i], e = rowp[i + 1];
    const int64_t o = outperm ? outperm[i] : i;
    for (int r0 = 0; r0 < nv; r0 += kPanelJB) {
      double acc[kPanelJB];
#pragma unroll
      for (int jj = 0; jj < kPanelJB; jj++) acc[jj] = 0.0;
      for (int q = b; q < e; q++) {
        const int k = cols[q];
        const double a = vals[q] * d[k];
#pragma unroll
        for (int jj = 0; jj < kPanelJB; jj++) {
          if (r0 + jj < nv) acc[jj] += a * P.p[r0 + jj][k];
        }
      }
#pragma unroll
      for (int jj = 0; jj < kPanelJB; jj++) {
        if (r0 + jj < nv) U.p[r0 + jj][o] = acc[jj];
      }
    }
  }
}
int k_csr_panel(Ctx *c, const int *rowp, const int *cols, const double *vals, int64_t w, const double *d,
                const double *const *P, int nv, double *const *U, const int *outperm) {
  if (w <= 0 || nv <= 0) return PO_OK;
  if (nv > kMaxPanel) {  // independent columns: slabs of one kernel's pointer table
    for (int j0 = 0; j0 < nv; j0 += kMaxPanel)
      PO_TRY(k_csr_panel(c, rowp, cols, vals, w, d, P + j0, nv - j0 > kMaxPanel ? kMaxPanel : nv - j0, U + j0, outperm));
    return PO_OK;
  }
  PtrTable pt;
  PtrTableW ut;
  for (int j = 0; j < kMaxPanel; j++) {
    pt.p[j] = j < nv ? P[j] : P[0];
    ut.p[j] = j < nv ? U[j] : U[0];
  }
  PO_CLAUNCH(csr_panel_kernel, cgrid(c, w), kBlock, rowp, cols, vals, w, d, pt, nv, ut, outperm);
  return PO_OK;
}

// One thread per structural entry (a, b) of lower(P S P^T): the two column-sorted rows of Aw are merged,
// Lvals[slot] = [a == b] cdiag[a] + sum_k Aw[a,k] dinv[k] Aw[b,k].  Fill entries were zeroed by the caller.
__global__ void __launch_bounds__(kBlock)
    csr_assemble_kernel(const int *__restrict__ rowp, const int *__restrict__ cols,
                        const double *__restrict__ vals, const double *__restrict__ dinv,
                        const double *__restrict__ cdiag, const int *__restrict__ ent_a,
                        const int *__restrict__ ent_b, const int *__restrict__ ent_slot, int64_t nent,
                        double *__restrict__ Lvals) {
  PO_C_LOOP(e, nent) {
    const int a = ent_a[e], b = ent_b[e];
    double acc = 0.0;
    if (a == b) {
      for (int q = rowp[a]; q < rowp[a + 1]; q++) acc += vals[q] * vals[q] * dinv[cols[q]];
      acc += cdiag[a];
    } else {
      int qa = rowp[a], qb = rowp[b];
      const int ea = rowp[a + 1], eb = rowp[b + 1];
      while (qa < ea && qb < eb) {
        const int ka = cols[qa], kb = cols[qb];
        if (ka == kb) {
          acc += vals[qa] * dinv[ka] * vals[qb];
          qa++;
          qb++;
        } else if (ka < kb) {
          qa++;
        } else {
          qb++;
        }
      }
    }
    Lvals[ent_slot[e]] = acc;
  }
}
int k_csr_assemble(Ctx *c, const int *rowp, const int *cols, const double *vals, const double *dinv,
                   const double *cdiag, const int *ent_a, const int *ent_b, const int *ent_slot, int64_t nent,
                   double *Lvals) {
  if (nent <= 0) return PO_OK;
  PO_CLAUNCH(csr_assemble_kernel, cgrid(c, nent), kBlock, rowp, cols, vals, dinv, cdiag, ent_a, ent_b, ent_slot,
             nent, Lvals);
  return PO_OK;
}

// built-in chain constraints (examples/rosenbrock/sparse_rosenbrock.cpp:75-118 is span 2, stride 1)
__global__ void __launch_bounds__(kBlock)
    chain_con_kernel(const double *__restrict__ x, int64_t w, int span, int stride, double *__restrict__ cw) {
  PO_C_LOOP(i, w) {
    double v = 1.0;
    for (int k = 0; k < span; k++) {
      const double xi = x[i * stride + k];
      v -= xi * xi;
    }
    cw[i] = v;
  }
}
int k_chain_con(Ctx *c, const double *x, int64_t w, int span, int stride, double *cw) {
  if (w <= 0) return PO_OK;
  PO_CLAUNCH(chain_con_kernel, cgrid(c, w), kBlock, x, w, span, stride, cw);
  return PO_OK;
}
__global__ void __launch_bounds__(kBlock)
    chain_jac_kernel(const double *__restrict__ x, int64_t w, int span, int stride, int reverse,
                     double *__restrict__ data) {
  PO_C_LOOP(e, w * span) {
    const int64_t i = e / span;
    const int s = (int)(e - i * span);
    const int k = reverse ? span - 1 - s : s;
    data[e] = -2.0 * x[i * stride + k];
  }
}
int k_chain_jac(Ctx *c, const double *x, int64_t w, int span, int stride, int reverse, double *data) {
  if (w <= 0) return PO_OK;
  PO_CLAUNCH(chain_jac_kernel, cgrid(c, w * span), kBlock, x, w, span, stride, reverse, data);
  return PO_OK;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// An LDS hash table column -> value of the part of row i computed so far (open addressing, linear probing,
// capacity a power of two >= twice the row length): a lookup is one or two LDS probes instead of a binary
// search through L2.  Layout in dynamic LDS: tcap ints of keys, then tcap doubles of values.
struct RowTable {
  int *key;
  double *val;
  int mask;
  __device__ __forceinline__ int slot(int k) const { return (int)(((unsigned)k * 2654435761u) >> 7) & mask; }
  __device__ __forceinline__ void insert(int k, double v) {  // one inserting thread at a time
    int h = slot(k);
    while (key[h] != -1) h = (h + 1) & mask;
    key[h] = k;
    val[h] = v;
  }
  __device__ __forceinline__ void insert_shared(int k, double v) {  // concurrent inserts of distinct keys
    int h = slot(k);
    while (atomicCAS(&key[h], -1, k) != -1) h = (h + 1) & mask;
    val[h] = v;
  }
  __device__ __forceinline__ double find(int k) const {  // 0 when absent
    int h = slot(k);
    int kk;
    while ((kk = key[h]) != -1) {
      if (kk == k) return val[h];
      h = (h + 1) & mask;
    }
    return 0.0;
  }
};
__device__ __forceinline__ RowTable row_table(int *lds, int tcap) {
  RowTable t;
  t.key = lds;
  t.val = reinterpret_cast<double *>(lds + tcap);
  t.mask = tcap - 1;
  return t;
}
constexpr int kMaxTable = 4096;  // 16 KB of keys + 32 KB of values
static int table_cap(int maxlen) {  // 0: the rows are too long, use the search kernels
  int cap = 64;
  while (cap < 2 * maxlen) cap <<= 1;
  return cap <= kMaxTable ? cap : 0;
}

// Row Cholesky of one dependency level: one wavefront per row i, entries left to right,
//   L_ij = (S_ij - sum_{k<j} L_ik L_jk) / L_jj,   L_ii = sqrt(S_ii - sum_k L_ik^2).
// Row j (j < i) was finished by an earlier launch; the lanes stride over row j and look each column up in the
// part of row i already computed: in the LDS table (TABLE) or by binary search on the sorted columns.
template <int TABLE>
__global__ void __launch_bounds__(64)
    chol_level_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols, double *Lvals,
                      const int *__restrict__ rows, int nrows, int *flag, int tcap) {
  extern __shared__ int lds[];
  const int i = rows[blockIdx.x];
  const int lane = threadIdx.x;
  const int p0 = Lrowp[i], pd = Lrowp[i + 1] - 1;
  RowTable tab = row_table(lds, TABLE ? tcap : 1);
  if (TABLE) {
    for (int h = lane; h < tcap; h += 64) tab.key[h] = -1;
    __syncthreads();
  }
  double dsum = 0.0;
  // 64 entries at a time: lane l fetches what entry pc + l needs from row j (its extent, its diagonal) and
  // the entry's own assembled value, so that none of those loads sits on the sequential chain
  for (int pc = p0; pc < pd; pc += 64) {
    int mj = 0, mr0 = 0, mrd = 0;
    double mdiag = 1.0, mval = 0.0;
    if (pc + lane < pd) {
      mj = Lcols[pc + lane];
      mr0 = Lrowp[mj];
      mrd = Lrowp[mj + 1] - 1;
      mdiag = Lvals[mrd];
      mval = Lvals[pc + lane];
    }
    const int cnt = pd - pc < 64 ? pd - pc : 64;
    int nk = -1;
    double nv = 0.0;
    if (TABLE && cnt > 0) {
      const int q = __shfl(mr0, 0, 64) + lane;
      if (q < __shfl(mrd, 0, 64)) {
        nk = Lcols[q];
        nv = Lvals[q];
      }
    }
    for (int e = 0; e < cnt; e++) {
      const int p = pc + e;
      double acc = 0.0;
      const int j = __shfl(mj, e, 64);
      const int r0 = __shfl(mr0, e, 64), rd = __shfl(mrd, e, 64);
      const double jdiag = __shfl(mdiag, e, 64), sval = __shfl(mval, e, 64);
      if (TABLE) {
        // the first 64 entries of row j were fetched while the previous entry was being finished; the loads of
        // the next entry start now, before this entry's arithmetic: they do not depend on it
        const int ck = nk;
        const double cv = nv;
        nk = -1;
        if (e + 1 < cnt) {
          const int q = __shfl(mr0, e + 1, 64) + lane;
          if (q < __shfl(mrd, e + 1, 64)) {
            nk = Lcols[q];
            nv = Lvals[q];
          }
        }
        if (ck >= 0) acc = cv * tab.find(ck);
        for (int q = r0 + 64 + lane; q < rd; q += 64) acc += Lvals[q] * tab.find(Lcols[q]);
      }
      for (int q = r0 + lane; !TABLE && q < rd; q += 64) {
        const int k = Lcols[q];
        {
          int lo = p0, hi = p;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Lcols[mid] < k) {
              lo = mid + 1;
            } else {
              hi = mid;
            }
          }
          if (lo < p && Lcols[lo] == k) acc += Lvals[q] * Lvals[lo];
        }
      }
      acc = wave_sum(acc);
      const double v = (sval - acc) / jdiag;
      dsum += v * v;
      if (lane == 0) {
        Lvals[p] = v;
        if (TABLE) tab.insert(j, v);
      }
      if (TABLE) {
        // one wavefront per workgroup: its LDS operations execute in program order, so the other lanes see the
        // insert without a barrier (which would also wait for the loads just started); the fence only keeps
        // the compiler from reordering
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      } else {
        __threadfence_block();
        __syncthreads();
      }
    }
  }
  double a = Lvals[pd] - dsum;
  if (!(a > 0.0)) {
    if (lane == 0) {
      flag[0] = 1;
      flag[1] = i;
    }
    a = 1.0;
  }
  if (lane == 0) Lvals[pd] = sqrt(a);
}
// The same recurrence with ONE THREAD per row, for levels whose rows are short (the leaves of the
// dissection tree: a wavefront per row would idle 60 of its 64 lanes and cost a workgroup launch per row).
__global__ void __launch_bounds__(kBlock)
    chol_level_thin_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols, double *Lvals,
                           const int *__restrict__ rows, int nrows, int *flag) {
  PO_C_LOOP(t, nrows) {
    const int i = rows[t];
    const int p0 = Lrowp[i], pd = Lrowp[i + 1] - 1;
    double dsum = 0.0;
    for (int p = p0; p < pd; p++) {
      const int j = Lcols[p];
      const int r0 = Lrowp[j], rd = Lrowp[j + 1] - 1;
      double acc = 0.0;
      int lo = p0;  // both rows are sorted: a merge instead of a search
      for (int q = r0; q < rd; q++) {
        const int k = Lcols[q];
        while (lo < p && Lcols[lo] < k) lo++;
        if (lo < p && Lcols[lo] == k) acc += Lvals[q] * Lvals[lo];
      }
      const double v = (Lvals[p] - acc) / Lvals[rd];
      Lvals[p] = v;
      dsum += v * v;
    }
    double a = Lvals[pd] - dsum;
    if (!(a > 0.0)) {
      flag[0] = 1;
      flag[1] = i;
      a = 1.0;
    }
    Lvals[pd] = sqrt(a);
  }
}
int k_chol_level(Ctx *c, const int *Lrowp, const int *Lcols, double *Lvals, const int *rows, int nrows,
                 int *flag, int thin, int maxlen) {
  if (nrows <= 0) return PO_OK;
  if (thin) {
    PO_CLAUNCH(chol_level_thin_kernel, cgrid(c, nrows), kBlock, Lrowp, Lcols, Lvals, rows, nrows, flag);
  } else {
    const int tcap = table_cap(maxlen);
    if (tcap > 0) {
      hipLaunchKernelGGL(chol_level_kernel<1>, dim3(nrows), dim3(64), (size_t)tcap * 12, c->stream, Lrowp, Lcols,
                         Lvals, rows, nrows, flag, tcap);
    } else {
      hipLaunchKernelGGL(chol_level_kernel<0>, dim3(nrows), dim3(64), 16, c->stream, Lrowp, Lcols, Lvals, rows,
                         nrows, flag, 1);
    }
    c->n_launches++;
    PO_HIP(hipGetLastError());
  }
  return PO_OK;
}

// ---- fronts: dense separator cliques (csr.hpp) ----------------------------------------------------------
// Row f0 + r of a front ends with the columns f0 .. f0 + r: T(r, q) = Lvals[Lrowp[f0 + r + 1] - 1 - r + q].
// Step 1 (all rows of all fronts of a level at once): the entries LEFT of the front, the recurrence of
// chol_level_kernel stopped at column f0.
template <int TABLE>
__global__ void __launch_bounds__(64)
    chol_front_rows_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols, double *Lvals, int row0,
                           const int *__restrict__ front_of, int tcap) {
  extern __shared__ int lds[];
  const int i = row0 + blockIdx.x;
  const int lane = threadIdx.x;
  const int f0 = front_of[i];
  const int p0 = Lrowp[i], pe = Lrowp[i + 1] - 1 - (i - f0);
  RowTable tab = row_table(lds, TABLE ? tcap : 1);
  if (TABLE) {
    for (int h = lane; h < tcap; h += 64) tab.key[h] = -1;
    __syncthreads();
  }
  for (int pc = p0; pc < pe; pc += 64) {  // metadata of 64 entries at a time, off the sequential chain
    int mj = 0, mr0 = 0, mrd = 0;
    double mdiag = 1.0, mval = 0.0;
    if (pc + lane < pe) {
      mj = Lcols[pc + lane];
      mr0 = Lrowp[mj];
      mrd = Lrowp[mj + 1] - 1;
      mdiag = Lvals[mrd];
      mval = Lvals[pc + lane];
    }
    const int cnt = pe - pc < 64 ? pe - pc : 64;
    int nk = -1;
    double nv = 0.0;
    if (TABLE && cnt > 0) {
      const int q = __shfl(mr0, 0, 64) + lane;
      if (q < __shfl(mrd, 0, 64)) {
        nk = Lcols[q];
        nv = Lvals[q];
      }
    }
    for (int e = 0; e < cnt; e++) {
      const int p = pc + e;
      double acc = 0.0;
      const int j = __shfl(mj, e, 64);
      const int r0 = __shfl(mr0, e, 64), rd = __shfl(mrd, e, 64);
      const double jdiag = __shfl(mdiag, e, 64), sval = __shfl(mval, e, 64);
      if (TABLE) {
        // the first 64 entries of row j were fetched while the previous entry was being finished; the loads of
        // the next entry start now, before this entry's arithmetic: they do not depend on it
        const int ck = nk;
        const double cv = nv;
        nk = -1;
        if (e + 1 < cnt) {
          const int q = __shfl(mr0, e + 1, 64) + lane;
          if (q < __shfl(mrd, e + 1, 64)) {
            nk = Lcols[q];
            nv = Lvals[q];
          }
        }
        if (ck >= 0) acc = cv * tab.find(ck);
        for (int q = r0 + 64 + lane; q < rd; q += 64) acc += Lvals[q] * tab.find(Lcols[q]);
      }
      for (int q = r0 + lane; !TABLE && q < rd; q += 64) {
        const int k = Lcols[q];
        {
          int lo = p0, hi = p;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (Lcols[mid] < k) {
              lo = mid + 1;
            } else {
              hi = mid;
            }
          }
          if (lo < p && Lcols[lo] == k) acc += Lvals[q] * Lvals[lo];
        }
      }
      acc = wave_sum(acc);
      const double v = (sval - acc) / jdiag;
      if (lane == 0) {
        Lvals[p] = v;
        if (TABLE) tab.insert(j, v);
      }
      if (TABLE) {
        // one wavefront per workgroup: its LDS operations execute in program order, so the other lanes see the
        // insert without a barrier (which would also wait for the loads just started); the fence only keeps
        // the compiler from reordering
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
      } else {
        __threadfence_block();
        __syncthreads();
      }
    }
  }
}
// The same step for levels whose front rows are LONG (the top separators: thousands of entries, each a dot with a
// row of hundreds): NW wavefronts share a row, every entry's dot is split over all of them and combined through
// LDS.  Two barriers per entry, but 1/NW of the arithmetic on the sequential chain.
template <int NW>
__global__ void __launch_bounds__(64 * NW)
    chol_front_rows_wide_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols, double *Lvals,
                                int row0, const int *__restrict__ front_of, int tcap) {
  extern __shared__ int lds[];
  __shared__ double red[NW];
  __shared__ double vbroadcast;
  const int i = row0 + blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int f0 = front_of[i];
  const int p0 = Lrowp[i], pe = Lrowp[i + 1] - 1 - (i - f0);
  RowTable tab = row_table(lds, tcap);
  for (int h = tid; h < tcap; h += 64 * NW) tab.key[h] = -1;
  __syncthreads();
  for (int pc = p0; pc < pe; pc += 64) {
    int mj = 0, mr0 = 0, mrd = 0;
    double mdiag = 1.0, mval = 0.0;
    if (pc + lane < pe) {  // every wavefront keeps its own copy of the chunk's metadata
      mj = Lcols[pc + lane];
      mr0 = Lrowp[mj];
      mrd = Lrowp[mj + 1] - 1;
      mdiag = Lvals[mrd];
      mval = Lvals[pc + lane];
    }
    const int cnt = pe - pc < 64 ? pe - pc : 64;
    for (int e = 0; e < cnt; e++) {
      const int j = __shfl(mj, e, 64);
      const int r0 = __shfl(mr0, e, 64), rd = __shfl(mrd, e, 64);
      const double jdiag = __shfl(mdiag, e, 64), sval = __shfl(mval, e, 64);
      double acc = 0.0;
      for (int q = r0 + tid; q < rd; q += 64 * NW) acc += Lvals[q] * tab.find(Lcols[q]);
      acc = wave_sum(acc);
      if (lane == 0) red[wave] = acc;
      __syncthreads();
      if (tid == 0) {
        double tot = 0.0;
#pragma unroll
        for (int k = 0; k < NW; k++) tot += red[k];
        const double v = (sval - tot) / jdiag;
        Lvals[pc + e] = v;
        tab.insert(j, v);
      }
      __syncthreads();
    }
  }
}
// Step 2: T(i, j) -= sum over the columns left of the front of L_ik L_jk, one wavefront per pair, four
// wavefronts share a row i (whose left part sits in the LDS table).
template <int TABLE>
__global__ void __launch_bounds__(kBlock)
    chol_front_syrk_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols, double *Lvals, int row0,
                           const int *__restrict__ front_of, int tcap) {
  extern __shared__ int lds[];
  const int i = row0 + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int f0 = front_of[i];
  const int p0 = Lrowp[i], pe = Lrowp[i + 1] - 1 - (i - f0);
  RowTable tab = row_table(lds, TABLE ? tcap : 1);
  if (TABLE) {
    for (int h = threadIdx.x; h < tcap; h += kBlock) tab.key[h] = -1;
    __syncthreads();
    for (int q = p0 + (int)threadIdx.x; q < pe; q += kBlock) tab.insert_shared(Lcols[q], Lvals[q]);
    __syncthreads();
  }
  for (int j = f0 + wave; j <= i; j += 4) {
    const int r0 = Lrowp[j], re = Lrowp[j + 1] - 1 - (j - f0);
    double acc = 0.0;
    for (int q = r0 + lane; q < re; q += 64) {
      const int k = Lcols[q];
      if (TABLE) {
        acc += Lvals[q] * tab.find(k);
      } else {
        int lo = p0, hi = pe;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (Lcols[mid] < k) {
            lo = mid + 1;
          } else {
            hi = mid;
          }
        }
        if (lo < pe && Lcols[lo] == k) acc += Lvals[q] * Lvals[lo];
      }
    }
    acc = wave_sum(acc);
    if (lane == 0) Lvals[pe + (j - f0)] -= acc;
  }
}
// Step 3: dense right-looking Cholesky of the triangle, one workgroup of 1024 per front.  Per column: the
// scaled column goes to LDS (and back to L), then the trailing triangle is updated from LDS - two barriers.
constexpr int kFrontThreads = 1024;
constexpr int kFrontLds = 2048;  // fronts up to this many rows keep their row offsets and the column in LDS
#define PO_T(r, q) Lvals[Lrowp[f0 + (r) + 1] - 1 - (r) + (q)]
__global__ void __launch_bounds__(kFrontThreads)
    chol_front_dense_kernel(const int *__restrict__ Lrowp, double *Lvals, const int *__restrict__ fstart,
                            const int *__restrict__ fsize, int *flag) {
  __shared__ int tb[kFrontLds];
  __shared__ double colk[kFrontLds];
  const int f0 = fstart[blockIdx.x], s = fsize[blockIdx.x];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  if (s <= kFrontLds) {
    for (int r = tid; r < s; r += kFrontThreads) tb[r] = Lrowp[f0 + r + 1] - 1 - r;
    __syncthreads();
    for (int k = 0; k < s; k++) {
      double a = Lvals[tb[k] + k];  // final since the previous trailing update
      if (!(a > 0.0)) {
        if (tid == 0) {
          flag[0] = 1;
          flag[1] = f0 + k;
        }
        a = 1.0;
      }
      const double d = sqrt(a);
      for (int r = k + tid; r < s; r += kFrontThreads) {
        const double v = r == k ? d : Lvals[tb[r] + k] / d;
        colk[r] = v;
        Lvals[tb[r] + k] = v;
      }
      __syncthreads();
      for (int r = k + 1 + ty; r < s; r += 32) {
        const int base = tb[r];
        const double lrk = colk[r];
        for (int q = k + 1 + tx; q <= r; q += 32) Lvals[base + q] -= lrk * colk[q];
      }
      __threadfence_block();
      __syncthreads();
    }
    return;
  }
  for (int k = 0; k < s; k++) {  // very large fronts: the same through global memory
    if (tid == 0) {
      double a = PO_T(k, k);
      if (!(a > 0.0)) {
        flag[0] = 1;
        flag[1] = f0 + k;
        a = 1.0;
      }
      PO_T(k, k) = sqrt(a);
    }
    __threadfence_block();
    __syncthreads();
    const double d = PO_T(k, k);
    for (int r = k + 1 + tid; r < s; r += kFrontThreads) PO_T(r, k) /= d;
    __threadfence_block();
    __syncthreads();
    for (int r = k + 1 + ty; r < s; r += 32) {
      const int base = Lrowp[f0 + r + 1] - 1 - r;
      const double lrk = Lvals[base + k];
      for (int q = k + 1 + tx; q <= r; q += 32) Lvals[base + q] -= lrk * PO_T(q, k);
    }
    __threadfence_block();
    __syncthreads();
  }
}
int k_chol_fronts(Ctx *c, const int *Lrowp, const int *Lcols, double *Lvals, int row0, int nrows,
                  const int *front_of, const int *fstart, const int *fsize, int nfronts, int maxdesc, int *flag) {
  if (nrows <= 0 || nfronts <= 0) return PO_OK;
  const int tcap = table_cap(maxdesc);
  if (tcap > 0 && maxdesc >= 256) {
    hipLaunchKernelGGL(chol_front_rows_wide_kernel<8>, dim3(nrows), dim3(512), (size_t)tcap * 12, c->stream, Lrowp,
                       Lcols, Lvals, row0, front_of, tcap);
  } else if (tcap > 0) {
    hipLaunchKernelGGL(chol_front_rows_kernel<1>, dim3(nrows), dim3(64), (size_t)tcap * 12, c->stream, Lrowp, Lcols,
                       Lvals, row0, front_of, tcap);
  } else {
    hipLaunchKernelGGL(chol_front_rows_kernel<0>, dim3(nrows), dim3(64), 16, c->stream, Lrowp, Lcols, Lvals, row0,
                       front_of, 1);
  }
  c->n_launches++;
  PO_HIP(hipGetLastError());
  if (tcap > 0) {
    hipLaunchKernelGGL(chol_front_syrk_kernel<1>, dim3(nrows), dim3(kBlock), (size_t)tcap * 12, c->stream, Lrowp,
                       Lcols, Lvals, row0, front_of, tcap);
  } else {
    hipLaunchKernelGGL(chol_front_syrk_kernel<0>, dim3(nrows), dim3(kBlock), 16, c->stream, Lrowp, Lcols, Lvals,
                       row0, front_of, 1);
  }
  c->n_launches++;
  PO_HIP(hipGetLastError());
  PO_CLAUNCH(chol_front_dense_kernel, nfronts, kFrontThreads, Lrowp, Lvals, fstart, fsize, flag);
  return PO_OK;
}

// Solves on a front.  Forward: first y_i -= sum over the columns left of the front (all rows at once), then the
// dense forward substitution by one workgroup.  Backward: first y_i -= sum over the rows BELOW the front in
// column i (the ancestors, through the column storage), then the dense backward substitution.
__global__ void __launch_bounds__(kBlock)
    trsv_front_fwd_rows_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols,
                               const double *__restrict__ Lvals, int row0, const int *__restrict__ front_of,
                               PtrTableW Y, int nv) {
  const int i = row0 + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = Lrowp[i], pe = Lrowp[i + 1] - 1 - (i - front_of[i]);
  for (int r = wave; r < nv; r += 4) {
    double *y = Y.p[r];
    double acc = 0.0;
    for (int q = p0 + lane; q < pe; q += 64) acc += Lvals[q] * y[Lcols[q]];
    acc = wave_sum(acc);
    if (lane == 0) y[i] -= acc;
  }
}
__global__ void __launch_bounds__(kBlock)
    trsv_front_bwd_cols_kernel(const int *__restrict__ Ltp, const int *__restrict__ Ltrows,
                               const int *__restrict__ Ltsrc, const double *__restrict__ Lvals, int row0,
                               const int *__restrict__ front_of, const int *__restrict__ front_end, PtrTableW Y,
                               int nv) {
  const int i = row0 + blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // the first entries of column i are the rows of the front below i
  const int q0 = Ltp[i] + (front_end[i] - 1 - i), q1 = Ltp[i + 1];
  for (int r = wave; r < nv; r += 4) {
    double *y = Y.p[r];
    double acc = 0.0;
    for (int q = q0 + lane; q < q1; q += 64) acc += Lvals[Ltsrc[q]] * y[Ltrows[q]];
    acc = wave_sum(acc);
    if (lane == 0) y[i] -= acc;
  }
}
__global__ void __launch_bounds__(kFrontThreads)
    trsv_front_fwd_dense_kernel(const int *__restrict__ Lrowp, const double *__restrict__ Lvals,
                                const int *__restrict__ fstart, const int *__restrict__ fsize, PtrTableW Y,
                                int nv) {
  const int f0 = fstart[blockIdx.x], s = fsize[blockIdx.x];
  const int tid = threadIdx.x;
  for (int k = 0; k < s; k++) {
    if (tid < nv) Y.p[tid][f0 + k] /= PO_T(k, k);
    __threadfence_block();
    __syncthreads();
    const int64_t items = (int64_t)(s - k - 1) * nv;
    for (int64_t t = tid; t < items; t += kFrontThreads) {
      const int r = (int)(t / (s - k - 1));
      const int i = k + 1 + (int)(t - (int64_t)r * (s - k - 1));
      double *y = Y.p[r];
      y[f0 + i] -= PO_T(i, k) * y[f0 + k];
    }
    __threadfence_block();
    __syncthreads();
  }
}
__global__ void __launch_bounds__(kFrontThreads)
    trsv_front_bwd_dense_kernel(const int *__restrict__ Lrowp, const double *__restrict__ Lvals,
                                const int *__restrict__ fstart, const int *__restrict__ fsize, PtrTableW Y,
                                int nv) {
  const int f0 = fstart[blockIdx.x], s = fsize[blockIdx.x];
  const int tid = threadIdx.x;
  for (int k = s - 1; k >= 0; k--) {
    if (tid < nv) Y.p[tid][f0 + k] /= PO_T(k, k);
    __threadfence_block();
    __syncthreads();
    const int64_t items = (int64_t)k * nv;
    for (int64_t t = tid; t < items; t += kFrontThreads) {
      const int r = (int)(t / k);
      const int i = (int)(t - (int64_t)r * k);
      double *y = Y.p[r];
      y[f0 + i] -= PO_T(k, i) * y[f0 + k];
    }
    __threadfence_block();
    __syncthreads();
  }
}
#undef PO_T
static int fill_table(PtrTableW &t, double *const *Y, int nv);
int k_trsv_fronts_fwd(Ctx *c, const int *Lrowp, const int *Lcols, const double *Lvals, int row0, int nrows,
                      const int *front_of, const int *fstart, const int *fsize, int nfronts, double *const *Y,
                      int nv) {
  if (nrows <= 0 || nfronts <= 0 || nv <= 0) return PO_OK;
  PtrTableW t;
  PO_TRY(fill_table(t, Y, nv));
  PO_CLAUNCH(trsv_front_fwd_rows_kernel, nrows, nv > 1 ? kBlock : 64, Lrowp, Lcols, Lvals, row0, front_of, t, nv);
  PO_CLAUNCH(trsv_front_fwd_dense_kernel, nfronts, kFrontThreads, Lrowp, Lvals, fstart, fsize, t, nv);
  return PO_OK;
}
int k_trsv_fronts_bwd(Ctx *c, const int *Lrowp, const int *Ltp, const int *Ltrows, const int *Ltsrc,
                      const double *Lvals, int row0, int nrows, const int *front_of, const int *front_end,
                      const int *fstart, const int *fsize, int nfronts, double *const *Y, int nv) {
  if (nrows <= 0 || nfronts <= 0 || nv <= 0) return PO_OK;
  PtrTableW t;
  PO_TRY(fill_table(t, Y, nv));
  PO_CLAUNCH(trsv_front_bwd_cols_kernel, nrows, nv > 1 ? kBlock : 64, Ltp, Ltrows, Ltsrc, Lvals, row0, front_of,
             front_end, t, nv);
  PO_CLAUNCH(trsv_front_bwd_dense_kernel, nfronts, kFrontThreads, Lrowp, Lvals, fstart, fsize, t, nv);
  return PO_OK;
}

// y_i = (y_i - sum_{j<i} L_ij y_j) / L_ii for the rows of one level; wavefront `wave` of the workgroup takes
// the right-hand sides wave, wave+4, ...
__global__ void __launch_bounds__(kBlock)
    trsv_fwd_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols,
                    const double *__restrict__ Lvals, const int *__restrict__ rows, int nrows, PtrTableW Y,
                    int nv) {
  const int i = rows[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = Lrowp[i], pd = Lrowp[i + 1] - 1;
  const double diag = Lvals[pd];
  for (int r = wave + 4 * blockIdx.y; r < nv; r += 4 * gridDim.y) {
    double *y = Y.p[r];
    double acc = 0.0;
    for (int q = p0 + lane; q < pd; q += 64) acc += Lvals[q] * y[Lcols[q]];
    acc = wave_sum(acc);
    if (lane == 0) y[i] = (y[i] - acc) / diag;
  }
}
// x_i = (y_i - sum_{j>i} L_ji x_j) / L_ii through the column storage of L
__global__ void __launch_bounds__(kBlock)
    trsv_bwd_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Ltp, const int *__restrict__ Ltrows,
                    const int *__restrict__ Ltsrc, const double *__restrict__ Lvals,
                    const int *__restrict__ rows, int nrows, PtrTableW Y, int nv) {
  const int i = rows[blockIdx.x];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p0 = Ltp[i], p1 = Ltp[i + 1];
  const double diag = Lvals[Lrowp[i + 1] - 1];
  for (int r = wave + 4 * blockIdx.y; r < nv; r += 4 * gridDim.y) {
    double *y = Y.p[r];
    double acc = 0.0;
    for (int q = p0 + lane; q < p1; q += 64) acc += Lvals[Ltsrc[q]] * y[Ltrows[q]];
    acc = wave_sum(acc);
    if (lane == 0) y[i] = (y[i] - acc) / diag;
  }
}
// thin variants: one thread per (row, right-hand side), consecutive threads on consecutive rows
__global__ void __launch_bounds__(kBlock)
    trsv_fwd_thin_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Lcols,
                         const double *__restrict__ Lvals, const int *__restrict__ rows, int nrows, PtrTableW Y,
                         int nv) {
  PO_C_LOOP(t, (int64_t)nrows * nv) {
    const int r = (int)(t / nrows);
    const int i = rows[t - (int64_t)r * nrows];
    double *y = Y.p[r];
    const int p0 = Lrowp[i], pd = Lrowp[i + 1] - 1;
    double acc = 0.0;
    for (int q = p0; q < pd; q++) acc += Lvals[q] * y[Lcols[q]];
    y[i] = (y[i] - acc) / Lvals[pd];
  }
}
__global__ void __launch_bounds__(kBlock)
    trsv_bwd_thin_kernel(const int *__restrict__ Lrowp, const int *__restrict__ Ltp,
                         const int *__restrict__ Ltrows, const int *__restrict__ Ltsrc,
                         const double *__restrict__ Lvals, const int *__restrict__ rows, int nrows, PtrTableW Y,
                         int nv) {
  PO_C_LOOP(t, (int64_t)nrows * nv) {
    const int r = (int)(t / nrows);
    const int i = rows[t - (int64_t)r * nrows];
    double *y = Y.p[r];
    double acc = 0.0;
    for (int q = Ltp[i]; q < Ltp[i + 1]; q++) acc += Lvals[Ltsrc[q]] * y[Ltrows[q]];
    y[i] = (y[i] - acc) / Lvals[Lrowp[i + 1] - 1];
  }
}
static int fill_table(PtrTableW &t, double *const *Y, int nv) {
  if (nv > kMaxPanel) {
    set_error("sparse triangular solve: %d right-hand sides exceed the panel limit %d", nv, kMaxPanel);
    return PO_ERR_ARG;
  }
  for (int j = 0; j < kMaxPanel; j++) t.p[j] = j < nv ? Y[j] : Y[0];
  return PO_OK;
}
int k_trsv_fwd_level(Ctx *c, const int *Lrowp, const int *Lcols, const double *Lvals, const int *rows, int nrows,
                     double *const *Y, int nv, int thin) {
  if (nrows <= 0 || nv <= 0) return PO_OK;
  PtrTableW t;
  PO_TRY(fill_table(t, Y, nv));
  if (thin) {
    PO_CLAUNCH(trsv_fwd_thin_kernel, cgrid(c, (int64_t)nrows * nv), kBlock, Lrowp, Lcols, Lvals, rows, nrows, t, nv);
    return PO_OK;
  }
  const int gy = nv > 4 ? (nrows < 1024 ? (nv + 3) / 4 : 1) : 1;
  hipLaunchKernelGGL(trsv_fwd_kernel, dim3(nrows, gy), dim3(nv > 1 ? kBlock : 64), 0, c->stream, Lrowp, Lcols,
                     Lvals, rows, nrows, t, nv);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return PO_OK;
}
int k_trsv_bwd_level(Ctx *c, const int *Lrowp, const int *Ltp, const int *Ltrows, const int *Ltsrc,
                     const double *Lvals, const int *rows, int nrows, double *const *Y, int nv, int thin) {
  if (nrows <= 0 || nv <= 0) return PO_OK;
  PtrTableW t;
  PO_TRY(fill_table(t, Y, nv));
  if (thin) {
    PO_CLAUNCH(trsv_bwd_thin_kernel, cgrid(c, (int64_t)nrows * nv), kBlock, Lrowp, Ltp, Ltrows, Ltsrc, Lvals, rows,
               nrows, t, nv);
    return PO_OK;
  }
  const int gy = nv > 4 ? (nrows < 1024 ? (nv + 3) / 4 : 1) : 1;
  hipLaunchKernelGGL(trsv_bwd_kernel, dim3(nrows, gy), dim3(nv > 1 ? kBlock : 64), 0, c->stream, Lrowp, Ltp,
                     Ltrows, Ltsrc, Lvals, rows, nrows, t, nv);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return PO_OK;
}

}  // namespace po
