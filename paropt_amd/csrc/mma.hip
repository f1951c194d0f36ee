// Element-wise kernels of the method of moving asymptotes (reference src/ParOptMMA.cpp:523-1010).
#include <math.h>

#include "core.hpp"
#include "mma.hpp"
#include "wcon.hpp"

namespace po {

#define PO_M_LOOP(i, n)                                                                   \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);               \
       i += (int64_t)gridDim.x * blockDim.x)
#define PO_MLAUNCH(kernel, grid, ...)                                                     \
  do {                                                                                    \
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBlock), 0, c->stream, __VA_ARGS__);      \
    c->n_launches++;                                                                      \
    PO_HIP(hipGetLastError());                                                            \
  } while (0)

// asymptote update :615-651
__global__ void __launch_bounds__(kBlock)
    mma_asymptotes_kernel(const double *__restrict__ x, const double *__restrict__ x1,
                          const double *__restrict__ x2, const double *__restrict__ lb,
                          const double *__restrict__ ub, MmaParams p, int first, int64_t n,
                          double *__restrict__ L, double *__restrict__ U) {
  PO_M_LOOP(j, n) {
    const double xj = x[j];
    const double lower = fmax(lb[j], xj - p.movlim), upper = fmin(ub[j], xj + p.movlim);
    if (first) {
      L[j] = xj - p.init_off * (upper - lower);
      U[j] = xj + p.init_off * (upper - lower);
    } else {
      const double indc = (xj - x1[j]) * (x1[j] - x2[j]);
      const double Lprev = L[j], Uprev = U[j];
      double intrvl = fmax(upper - lower, 0.01);
      intrvl = fmin(intrvl, 100.0);
      double l, u;
      if (indc < 0.0) {
        l = xj - p.contract * (x1[j] - Lprev);
        u = xj + p.contract * (Uprev - x1[j]);
      } else {
        l = xj - p.relax * (x1[j] - Lprev);
        u = xj + p.relax * (Uprev - x1[j]);
      }
      l = fmin(l, xj - p.min_off * intrvl);
      u = fmax(u, xj + p.min_off * intrvl);
      l = fmax(l, xj - p.max_off * intrvl);
      u = fmin(u, xj + p.max_off * intrvl);
      L[j] = l;
      U[j] = u;
    }
  }
}
int k_mma_asymptotes(Ctx *c, const double *x, const double *x1, const double *x2, const double *lb,
                     const double *ub, const MmaParams &p, int first, int64_t n, double *L, double *U) {
  if (n <= 0) return PO_OK;
  PO_MLAUNCH(mma_asymptotes_kernel, grid_for(c, n), x, x1, x2, lb, ub, p, first, n, L, U);
  return PO_OK;
}

// move limits and objective coefficients :669-690
__global__ void __launch_bounds__(kBlock)
    mma_coef_kernel(const double *__restrict__ x, const double *__restrict__ lb, const double *__restrict__ ub,
                    const double *__restrict__ L, const double *__restrict__ U, const double *__restrict__ g,
                    MmaParams p, int64_t n, double *__restrict__ alpha, double *__restrict__ beta,
                    double *__restrict__ p0, double *__restrict__ q0) {
  PO_M_LOOP(j, n) {
    const double xj = x[j], Lj = L[j], Uj = U[j];
    const double lower = fmax(lb[j], xj - p.movlim), upper = fmin(ub[j], xj + p.movlim);
    alpha[j] = fmax(fmax(lower, 0.9 * Lj + 0.1 * xj), xj - 0.5 * (upper - lower));
    beta[j] = fmin(fmin(upper, 0.9 * Uj + 0.1 * xj), xj + 0.5 * (upper - lower));
    const double gpos = fmax(0.0, g[j]), gneg = fmax(0.0, -g[j]);
    p0[j] = (Uj - xj) * (Uj - xj) * ((1.0 + p.delta) * gpos + p.delta * gneg + p.eps / (Uj - Lj));
    q0[j] = (xj - Lj) * (xj - Lj) * ((1.0 + p.delta) * gneg + p.delta * gpos + p.eps / (Uj - Lj));
  }
}
int k_mma_coef(Ctx *c, const double *x, const double *lb, const double *ub, const double *L, const double *U,
               const double *g, const MmaParams &p, int64_t n, double *alpha, double *beta, double *p0,
               double *q0) {
  if (n <= 0) return PO_OK;
  PO_MLAUNCH(mma_coef_kernel, grid_for(c, n), x, lb, ub, L, U, g, p, n, alpha, beta, p0, q0);
  return PO_OK;
}

// constraint coefficients :692-712
__global__ void __launch_bounds__(kBlock)
    mma_pq_kernel(const double *__restrict__ x, const double *__restrict__ L, const double *__restrict__ U,
                  const double *__restrict__ A, int64_t n, double *__restrict__ pi, double *__restrict__ qi,
                  double *__restrict__ partials) {
  __shared__ double sm[4];
  double s = 0.0;
  PO_M_LOOP(j, n) {
    const double xj = x[j], du = U[j] - xj, dl = xj - L[j];
    const double gpos = fmax(0.0, -A[j]), gneg = fmax(0.0, A[j]);
    const double pv = du * du * gpos, qv = dl * dl * gneg;
    pi[j] = pv;
    qi[j] = qv;
    s += pv / du + qv / dl;
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) sm[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
int k_mma_pq(Ctx *c, const double *x, const double *L, const double *U, const double *A, int64_t n, double *pi,
             double *qi, double *bsum) {
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid));
  PO_MLAUNCH(mma_pq_kernel, grid, x, L, U, A, n, pi, qi, c->d_partials);
  return reduce_finish(c, grid, 1, 0, 0, bsum);
}

__global__ void __launch_bounds__(kBlock)
    mma_inv_kernel(const double *__restrict__ x, const double *__restrict__ L, const double *__restrict__ U,
                   int64_t n, double *__restrict__ uinv, double *__restrict__ linv) {
  PO_M_LOOP(j, n) {
    uinv[j] = 1.0 / (U[j] - x[j]);
    linv[j] = 1.0 / (x[j] - L[j]);
  }
}
int k_mma_inv(Ctx *c, const double *x, const double *L, const double *U, int64_t n, double *uinv, double *linv) {
  if (n <= 0) return PO_OK;
  PO_MLAUNCH(mma_inv_kernel, grid_for(c, n), x, L, U, n, uinv, linv);
  return PO_OK;
}

// gradients of the rational approximations :871-924, all constraints in one pass
__global__ void __launch_bounds__(kBlock)
    mma_grad_kernel(const double *__restrict__ x, const double *__restrict__ L, const double *__restrict__ U,
                    PtrTable P, PtrTable Q, int nv, int64_t n, PtrTableW out) {
  PO_M_LOOP(j, n) {
    const double ui = 1.0 / (U[j] - x[j]), li = 1.0 / (x[j] - L[j]);
    const double u2 = ui * ui, l2 = li * li;
    out.p[0][j] = u2 * P.p[0][j] - l2 * Q.p[0][j];
    for (int i = 1; i < nv; i++) out.p[i][j] = l2 * Q.p[i][j] - u2 * P.p[i][j];
  }
}
int k_mma_grad(Ctx *c, const double *x, const double *L, const double *U, const double *const *P,
               const double *const *Q, int nv, int64_t n, double *const *out) {
  if (n <= 0 || nv <= 0) return PO_OK;
  if (nv > kMaxPanel) {
    set_error("MMA: %d constraints exceed the panel width %d", nv - 1, kMaxPanel - 1);
    return PO_ERR_ARG;
  }
  PtrTable pt, qt;
  PtrTableW ot;
  for (int i = 0; i < kMaxPanel; i++) {
    pt.p[i] = i < nv ? P[i] : nullptr;
    qt.p[i] = i < nv ? Q[i] : nullptr;
    ot.p[i] = i < nv ? out[i] : nullptr;
  }
  PO_MLAUNCH(mma_grad_kernel, grid_for(c, n), x, L, U, pt, qt, nv, n, ot);
  return PO_OK;
}

// diagonal Hessian of the subproblem Lagrangian :967-1010
__global__ void __launch_bounds__(kBlock)
    mma_hdiag_kernel(const double *__restrict__ x, const double *__restrict__ L, const double *__restrict__ U,
                     PtrTable P, PtrTable Q, CoefTable w, int nv, int64_t n, double *__restrict__ h) {
  PO_M_LOOP(j, n) {
    const double ui = 1.0 / (U[j] - x[j]), li = 1.0 / (x[j] - L[j]);
    const double u3 = ui * ui * ui, l3 = li * li * li;
    double s = 0.0;
    for (int i = 0; i < nv; i++) s += 2.0 * w.a[i] * (u3 * P.p[i][j] + l3 * Q.p[i][j]);
    h[j] = s;
  }
}
int k_mma_hdiag(Ctx *c, const double *x, const double *L, const double *U, const double *const *P,
                const double *const *Q, const double *w, int nv, int64_t n, double *h) {
  if (n <= 0) return PO_OK;
  PtrTable pt, qt;
  CoefTable ct;
  for (int i = 0; i < kMaxPanel; i++) {
    pt.p[i] = i < nv ? P[i] : nullptr;
    qt.p[i] = i < nv ? Q[i] : nullptr;
    ct.a[i] = i < nv ? w[i] : 0.0;
  }
  PO_MLAUNCH(mma_hdiag_kernel, grid_for(c, n), x, L, U, pt, qt, ct, nv, n, h);
  return PO_OK;
}

}  // namespace po
