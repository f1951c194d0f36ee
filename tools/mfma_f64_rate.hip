// Issue-rate sweep of the fp64 matrix instructions of gfx950 against the fp64 vector FMA:
//   kind 0: v_mfma_f64_16x16x4_f64 (2048 flop / instruction / wave)
//   kind 1: v_mfma_f64_4x4x4_4b_f64 (4 blocks x 4x4x4, 512 flop / instruction / wave)
//   kind 2: v_fma_f64 (128 flop / instruction / wave)
// NACC independent accumulators per wave, 1..8 waves per SIMD (256-thread workgroups, wpc per CU).
// The shader clock during the run is estimated from s_memtime (core clock) over s_memrealtime (100 MHz).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_rate.hip -o tools/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int KIND, int NACC>
__global__ void __launch_bounds__(256) k(double *out, unsigned long long *clk, int iters, double a, double b) {
  double av = a + threadIdx.x * 1e-9, bv = b - threadIdx.x * 1e-9;
  if (a < 0.0) {  // random mantissas per lane (the power drawn, and with it the clock the chip holds, depends on data)
    unsigned long long z = (blockIdx.x * 256ull + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    av = 0.5 + (double)(z >> 11) * (1.0 / 9007199254740992.0);
    z = (z ^ (z >> 29)) * 0xBF58476D1CE4E5B9ull;
    bv = -(0.5 + (double)(z >> 11) * (1.0 / 9007199254740992.0)) * 1e-3;
  }
  double s = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  if (KIND == 0) {
    f64x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = (f64x4){0, 0, 0, 0};
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (KIND == 3) {
    // the Gram kernel's pattern: a different A operand per instruction, a B operand shared by runs of 8
    double acc[NACC], aa[8], bb[2];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) aa[i] = av * (1.0 + 0.01 * i);
    bb[0] = bv;
    bb[1] = bv * 1.5;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(aa[i & 7], bb[(i >> 3) & 1], acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
  } else if (KIND == 1) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 0.0;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
  } else {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) acc[i] = 1e-3 * i;
    for (int it = 0; it < iters; it++) {
#pragma unroll
      for (int i = 0; i < NACC; i++) acc[i] = __builtin_fma(av, acc[i], bv);
    }
#pragma unroll
    for (int i = 0; i < NACC; i++) s += acc[i];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (blockIdx.x == 0 && threadIdx.x == 0) {
    clk[0] = t1 - t0;
    clk[1] = r1 - r0;
  }
}

template <int KIND, int NACC>
void run(int wpc, double a0 = 1.0) {
  double *out;
  unsigned long long *clk, h[2];
  hipMalloc(&out, 256 * 256 * wpc * 8);
  hipMalloc(&clk, 16);
  const int iters = (KIND == 0 ? 6000 : 20000) * (a0 < 0.0 ? 20 : 1);  // long enough for the clock to settle
  const double flop = KIND == 0 ? 2048.0 : ((KIND == 1 || KIND == 3) ? 512.0 : 128.0);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<KIND, NACC>), dim3(256 * wpc), dim3(256), 0, 0, out, clk, 100, a0, 2.0);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<KIND, NACC>), dim3(256 * wpc), dim3(256), 0, 0, out, clk, iters, a0, 2.0);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  const double mhz = h[1] ? (double)h[0] / (double)h[1] * 100.0 : 0.0;
  const double n_per_simd = (double)iters * NACC * wpc;  // one wave per SIMD per workgroup
  const double tflops = (double)iters * NACC * flop * 4 * 256 * wpc / (ms * 1e-3) * 1e-12;
  const double ns = ms * 1e6 / n_per_simd;
  printf("%skind=%s NACC=%d waves/SIMD=%d: %.3f ms, %.1f TFLOP/s, %.2f ns per instr per SIMD (%.1f cycles @2.4GHz; "
         "memtime/realtime -> %.0f MHz -> %.1f cycles)\n",
         a0 < 0.0 ? "[random operands] " : "", KIND == 0 ? "mfma16x16x4" : (KIND == 1 ? "mfma4x4x4" : (KIND == 3 ? "mfma4x4x4/varied-operands" : "v_fma_f64")), NACC, wpc, ms, tflops, ns, ns * 2.4, mhz,
         ns * mhz * 1e-3);
  hipFree(out);
  hipFree(clk);
}

int main() {
  run<0, 1>(1); run<0, 6>(1); run<0, 6>(2); run<0, 6>(3); run<0, 4>(4); run<0, 4>(8);
  run<1, 1>(1); run<1, 8>(1); run<1, 16>(1); run<1, 16>(2); run<1, 16>(3); run<1, 16>(4); run<1, 8>(8);
  run<2, 8>(1); run<2, 8>(2); run<2, 8>(4); run<2, 8>(8);
  // the same with random mantissas, 20x longer: what a real Gram pass can expect
  run<1, 16>(1, -1.0); run<1, 16>(3, -1.0); run<0, 6>(3, -1.0); run<2, 8>(8, -1.0);
  run<3, 16>(1, -1.0); run<3, 16>(3, -1.0); run<3, 16>(1); run<3, 16>(3);
  return 0;
}
