// Does the panel's memory layout limit the solve passes?  The same pass -- acc = sum_j a_j P_j over m = 42 columns,
// ten more operand streams, four output streams (the shape of solve2_kernel) -- over
//   (A) m separate column vectors (the product's layout: ParOptVec per constraint gradient / quasi-Newton column)
//   (B) one tile-interleaved array: for every tile of 512 rows the m column segments of 4 KB lie back to back
// and the read-only reduction (the shape of mdot) over both.  n = 50 M rows, fp64.
// Build: hipcc --offload-arch=gfx950 -O3 tools/layout_probe.hip -o tools/layout_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int M = 42, NOP = 10, NOUT = 4, TILE_PAIRS = 256;

struct Ptrs {
  const double *p[M];
};
struct Coef {
  double a[M];
};
struct Ops {
  const double *in[NOP];
  double *out[NOUT];
};

__device__ __forceinline__ f64x2 ldnt(const double *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
}
__device__ __forceinline__ void stnt(double *p, f64x2 v) {
  __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(p));
}

template <int LAYOUT, int WRITES>
__global__ void __launch_bounds__(256) pass_kernel(Ptrs P, const double *__restrict__ PB, Coef a, Ops ops, long npairs,
                                                    double *__restrict__ partial) {
  double red = 0.0;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (long)gridDim.x * 256) {
    f64x2 acc = (f64x2){0.0, 0.0};
    const long tile = q / TILE_PAIRS, r = q % TILE_PAIRS;
    const double *base = PB + tile * (long)(M * 2 * TILE_PAIRS) + 2 * r;
#pragma unroll
    for (int j0 = 0; j0 < M; j0 += 8) {
      f64x2 v[8];
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u;
        if (j < M) v[u] = LAYOUT == 0 ? ldnt(P.p[j] + 2 * q) : ldnt(base + (long)j * 2 * TILE_PAIRS);
      }
#pragma unroll
      for (int u = 0; u < 8; u++) {
        const int j = j0 + u;
        if (j < M) acc += a.a[j] * v[u];
      }
    }
    if (WRITES == 1 || WRITES == 3) {
      f64x2 e = (f64x2){0.0, 0.0};
      if (WRITES == 1) {
#pragma unroll
        for (int i = 0; i < NOP; i++) e += ldnt(ops.in[i] + 2 * q);
      }
      const f64x2 o = acc + e;
#pragma unroll
      for (int i = 0; i < NOUT; i++) stnt(ops.out[i] + 2 * q, o * (double)(i + 1));
    } else if (WRITES == 2) {
      f64x2 e = (f64x2){0.0, 0.0};
#pragma unroll
      for (int i = 0; i < NOP; i++) e += ldnt(ops.in[i] + 2 * q);
      red += (acc.x + e.x) + (acc.y + e.y);
    } else {
      const f64x2 x = ldnt(ops.in[0] + 2 * q);
      red += acc.x * x.x + acc.y * x.y;
    }
  }
  if (WRITES == 0 || WRITES == 2) {
    for (int o = 32; o > 0; o >>= 1) red += __shfl_xor(red, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(partial + (blockIdx.x & 1023), red);
  }
}

template <int LAYOUT, int WRITES>
double run(const char *name, Ptrs P, const double *PB, Coef a, Ops ops, long npairs, double *partial, int bpc) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256 * bpc;
  float best = 1e30f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((pass_kernel<LAYOUT, WRITES>), dim3(grid), dim3(256), 0, 0, P, PB, a, ops, npairs, partial);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r == 0) continue;
    sum += ms;
    if (ms < best) best = ms;
  }
  const double bytes = 16.0 * npairs * (M + (WRITES == 1 ? NOP + NOUT : (WRITES == 2 ? NOP : (WRITES == 3 ? NOUT : 1))));
  printf("%-44s bpc=%d  avg %.3f ms  best %.3f ms  %.0f GB/s (best %.0f)\n", name, bpc, sum / (reps - 1), best,
         bytes / (sum / (reps - 1) * 1e-3) * 1e-9, bytes / (best * 1e-3) * 1e-9);
  return best;
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 50000000L;
  const long npairs = n / 2, ntiles = (npairs + TILE_PAIRS - 1) / TILE_PAIRS;
  Ptrs P;
  Coef a;
  Ops ops;
  for (int j = 0; j < M; j++) {
    double *p;
    if (hipMalloc(&p, n * 8) != hipSuccess) return 1;
    hipMemset(p, 0, n * 8);
    P.p[j] = p;
    a.a[j] = 1.0 + 0.01 * j;
  }
  double *PB;
  if (hipMalloc(&PB, ntiles * (long)M * 2 * TILE_PAIRS * 8) != hipSuccess) return 1;
  hipMemset(PB, 0, ntiles * (long)M * 2 * TILE_PAIRS * 8);
  for (int i = 0; i < NOP; i++) {
    double *p;
    hipMalloc(&p, n * 8);
    hipMemset(p, 0, n * 8);
    ops.in[i] = p;
  }
  for (int i = 0; i < NOUT; i++) {
    double *p;
    hipMalloc(&p, n * 8);
    ops.out[i] = p;
  }
  double *partial;
  hipMalloc(&partial, 1024 * 8);
  hipMemset(partial, 0, 1024 * 8);
  hipDeviceSynchronize();
  for (int round = 0; round < 2; round++) {
    for (int bpc : {3}) {
      run<0, 1>("solve-shaped, 42 separate columns", P, PB, a, ops, npairs, partial, bpc);
      run<1, 1>("solve-shaped, tile-interleaved panel", P, PB, a, ops, npairs, partial, bpc);
      run<0, 0>("reduction-shaped, 42 separate columns", P, PB, a, ops, npairs, partial, bpc);
      run<1, 0>("reduction-shaped, tile-interleaved panel", P, PB, a, ops, npairs, partial, bpc);
      run<0, 2>("panel + 10 operand reads, separate", P, PB, a, ops, npairs, partial, bpc);
      run<1, 2>("panel + 10 operand reads, interleaved", P, PB, a, ops, npairs, partial, bpc);
      run<0, 3>("panel + 4 output streams, separate", P, PB, a, ops, npairs, partial, bpc);
      run<1, 3>("panel + 4 output streams, interleaved", P, PB, a, ops, npairs, partial, bpc);
    }
  }
  return 0;
}
