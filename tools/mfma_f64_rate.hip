// Measures the issue rate of v_mfma_f64_16x16x4_f64 on one SIMD (independent accumulators).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ void __launch_bounds__(256) k(double* out, int iters, double a, double b) {
  f64x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0, 0, 0, 0};
  double av = a + threadIdx.x * 1e-9, bv = b - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(int wpc) {
  double* out; hipMalloc(&out, 256 * 256 * wpc * 8);
  int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(256 * wpc), dim3(256), 0, 0, out, 100, 1.0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(256 * wpc), dim3(256), 0, 0, out, iters, 1.0, 2.0);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double nmfma_per_simd = (double)iters * NACC * wpc;  // one wave per SIMD per block, wpc blocks per CU
  double tflops = (double)iters * NACC * 2048.0 * 4 * 256 * wpc / (ms * 1e-3) * 1e-12;
  printf("NACC=%d blocks/CU=%d: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD (%.1f cycles @2.4GHz)\n", NACC, wpc, ms, tflops,
         ms * 1e6 / nmfma_per_simd, ms * 1e6 / nmfma_per_simd * 2.4);
  hipFree(out);
}
int main() { run<1>(1); run<2>(1); run<4>(1); run<6>(1); run<8>(1); run<6>(2); run<6>(3); return 0; }
