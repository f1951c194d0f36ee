// Tuning harness (not part of the product): sweeps register-batch size, grid size and
// non-temporal loads for the panel-dot streaming pattern.  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double f64x2 __attribute__((ext_vector_type(2)));
struct PtrTable { const double* p[64]; };
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NT> __device__ __forceinline__ f64x2 ldv(const double* p) {
  if (NT) return __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(p));
  return *reinterpret_cast<const f64x2*>(p);
}

template <int NVB, int B, int NT, int U>
__global__ void __launch_bounds__(256) mdot_k(const double* __restrict__ x, PtrTable V, long n, double* __restrict__ out) {
  constexpr int NB = (NVB + B - 1) / B;
  double acc[NVB];
  const double* vp[NVB];
#pragma unroll
  for (int j = 0; j < NVB; j++) { acc[j] = 0.0; vp[j] = V.p[j]; }
  const long npairs = n >> 1;
  for (long q0 = ((long)blockIdx.x * 256 + threadIdx.x) * U; q0 < npairs; q0 += (long)gridDim.x * 256 * U) {
#pragma unroll
    for (int u = 0; u < U; u++) {
      const long q = q0 + u;
      const f64x2 xv = ldv<NT>(x + 2 * q);
      f64x2 v[2][B];
#pragma unroll
      for (int j = 0; j < B; j++) v[0][j] = ldv<NT>(vp[j] + 2 * q);
#pragma unroll
      for (int b = 0; b < NB; b++) {
        if (b + 1 < NB) {
#pragma unroll
          for (int j = 0; j < B; j++) if ((b + 1) * B + j < NVB) v[(b + 1) & 1][j] = ldv<NT>(vp[(b + 1) * B + j] + 2 * q);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < B; j++) if (b * B + j < NVB) { const f64x2 w = v[b & 1][j]; acc[b * B + j] = fma(xv.x, w.x, fma(xv.y, w.y, acc[b * B + j])); }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < NVB; j++) s += acc[j];
  if (s == 123.456) out[0] = s;  // keep alive
}

// serialized variant (what hipcc generates by default): high occupancy, one load at a time
template <int NVB, int NT>
__global__ void __launch_bounds__(256) mdot_serial(const double* __restrict__ x, PtrTable V, long n, double* __restrict__ out) {
  double acc[NVB];
#pragma unroll
  for (int j = 0; j < NVB; j++) acc[j] = 0.0;
  const long npairs = n >> 1;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (long)gridDim.x * 256) {
    const f64x2 xv = ldv<NT>(x + 2 * q);
#pragma unroll
    for (int j = 0; j < NVB; j++) { const f64x2 w = ldv<NT>(V.p[j] + 2 * q); acc[j] = fma(xv.x, w.x, fma(xv.y, w.y, acc[j])); }
  }
  double s = 0;
#pragma unroll
  for (int j = 0; j < NVB; j++) s += acc[j];
  if (s == 123.456) out[0] = s;
}

template <typename F> double timeit(F f, int reps) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  f(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(a));
  for (int r = 0; r < reps; r++) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b));
  return ms / reps;
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 50000000;
  const int NVMAX = 32;
  std::vector<double*> bufs(NVMAX + 1);
  for (auto& p : bufs) { CK(hipMalloc(&p, (n + 2) * 8)); CK(hipMemset(p, 0, (n + 2) * 8)); }
  double* out; CK(hipMalloc(&out, 8));
  PtrTable pt; for (int j = 0; j < 64; j++) pt.p[j] = bufs[1 + (j % NVMAX)];
  const double* x = bufs[0];
  int cus = 256;
  printf("n=%ld\n", n);
#define RUN(NAME, NV, KERNEL)                                                        \
  for (int bpc : {2, 3, 4, 6, 8, 16}) {                                              \
    int grid = cus * bpc;                                                            \
    double ms = timeit([&] { hipLaunchKernelGGL(KERNEL, dim3(grid), dim3(256), 0, 0, x, pt, n, out); }, 10); \
    printf("%-28s nv=%2d blocks/CU=%2d  %.3f ms  %.0f GB/s\n", NAME, NV, bpc, ms, 8.0 * (NV + 1) * n / ms * 1e-6); \
  }
  RUN("serial32", 32, (mdot_serial<32, 0>))
  RUN("serial32_nt", 32, (mdot_serial<32, 1>))
  RUN("b2_32", 32, (mdot_k<32, 2, 0, 1>))
  RUN("b4_32", 32, (mdot_k<32, 4, 0, 1>))
  RUN("b8_32", 32, (mdot_k<32, 8, 0, 1>))
  RUN("b8_32_nt", 32, (mdot_k<32, 8, 1, 1>))
  RUN("b4_32_nt", 32, (mdot_k<32, 4, 1, 1>))
  RUN("b16_32", 32, (mdot_k<32, 16, 0, 1>))
  RUN("b4_16", 16, (mdot_k<16, 4, 0, 1>))
  RUN("b8_16", 16, (mdot_k<16, 8, 0, 1>))
  RUN("b8_16_nt", 16, (mdot_k<16, 8, 1, 1>))
  RUN("serial16", 16, (mdot_serial<16, 0>))
  RUN("serial8", 8, (mdot_serial<8, 0>))
  RUN("b8_8", 8, (mdot_k<8, 8, 0, 1>))
  RUN("b4_8", 8, (mdot_k<8, 4, 0, 1>))
  RUN("b8_8_u2", 8, (mdot_k<8, 8, 0, 2>))
  RUN("b8_8_nt", 8, (mdot_k<8, 8, 1, 1>))
  return 0;
}
