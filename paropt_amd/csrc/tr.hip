// Element-wise kernels of the trust-region driver (reference src/ParOptTrustRegion.cpp).
#include <math.h>

#include "core.hpp"
#include "tr.hpp"

namespace po {

#define PO_T_LOOP(i, n)                                                                   \
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (n);               \
       i += (int64_t)gridDim.x * blockDim.x)

// setTrustRegionBounds :156-173: lk = max(-tr, lb - xk), uk = min(tr, ub - xk)
__global__ void __launch_bounds__(kBlock)
    tr_bounds_kernel(const double *__restrict__ xk, const double *__restrict__ lb,
                     const double *__restrict__ ub, double tr, int64_t n, double *__restrict__ lk,
                     double *__restrict__ uk) {
  PO_T_LOOP(i, n) {
    const double x = xk[i];
    lk[i] = fmax(-tr, lb[i] - x);
    uk[i] = fmin(tr, ub[i] - x);
  }
}
int k_tr_bounds(Ctx *c, const double *xk, const double *lb, const double *ub, double tr, int64_t n,
                double *lk, double *uk) {
  if (n <= 0) return PO_OK;
  hipLaunchKernelGGL(tr_bounds_kernel, dim3(grid_for(c, n)), dim3(kBlock), 0, c->stream, xk, lb, ub, tr,
                     n, lk, uk);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return PO_OK;
}

// computeKKTError :2391-2472: r = g - A^T z (already in t); components pushing into an active bound
// (within `relax`) are dropped; out = {l1, linf}
__global__ void __launch_bounds__(kBlock)
    kkt_error_kernel(const double *__restrict__ xk, const double *__restrict__ lb,
                     const double *__restrict__ ub, const double *__restrict__ t, double relax,
                     int64_t n, double *__restrict__ partials) {
  __shared__ double sm[8];
  double s = 0.0, m = 0.0;
  PO_T_LOOP(i, n) {
    double w = t[i];
    const double x = xk[i];
    if (x <= lb[i] + relax && w > 0.0) {
      w = 0.0;
    } else if (x >= ub[i] - relax && w < 0.0) {
      w = 0.0;
    }
    const double a = fabs(w);
    s += a;
    m = fmax(m, a);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    s += __shfl_xor(s, o, 64);
    m = fmax(m, __shfl_xor(m, o, 64));
  }
  if (lane == 0) {
    sm[wave] = s;
    sm[4 + wave] = m;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
    partials[(size_t)gridDim.x + blockIdx.x] = fmax(fmax(sm[4], sm[5]), fmax(sm[6], sm[7]));
  }
}
int k_kkt_error(Ctx *c, const double *xk, const double *lb, const double *ub, const double *t,
                double relax, int64_t n, double out[2]) {
  const int grid = grid_for(c, n);
  PO_TRY(ensure_partials(c, (size_t)grid * 2));
  hipLaunchKernelGGL(kkt_error_kernel, dim3(grid), dim3(kBlock), 0, c->stream, xk, lb, ub, t, relax, n,
                     c->d_partials);
  c->n_launches++;
  PO_HIP(hipGetLastError());
  return reduce_finish(c, grid, 1, 0, 1, out);
}

}  // namespace po
