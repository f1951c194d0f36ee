// Which cache policy should the n-sized OUTPUT streams of the passes use?  gfx950 global stores carry three policy
// bits (sc0, sc1, nt); MI355X_MICROARCH.md: plain / sc0 / nt keep the written line in the XCD's L2, sc1 / sc0 sc1
// write through and drop it.  Three pass shapes of the interior-point iteration, fp64, n = 50 M rows:
//   solve-shaped   42 input streams + 4 output streams        (solve2r / kkt_res_update)
//   copy-shaped    32 input streams + 32 output streams       (the problem's Jacobian rewrite, panel_lincomb)
//   form-shaped    20 input streams + 10 output streams       (the L-SR1 column formation inside the Gram pass)
// each with the six store flavours.  Loads are non-temporal throughout (the product's choice).
// Build: hipcc --offload-arch=gfx950 -O3 tools/store_policy_probe.hip -o tools/store_policy_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef double f64x2 __attribute__((ext_vector_type(2)));
constexpr int MAXS = 50;

struct Streams {
  const double *in[MAXS];
  double *out[32];
};

__device__ __forceinline__ f64x2 ldnt(const double *p) {
  return __builtin_nontemporal_load(reinterpret_cast<const f64x2 *>(p));
}

template <int POLICY>
__device__ __forceinline__ void st(double *p, f64x2 v) {
  if (POLICY == 0) {
    *reinterpret_cast<f64x2 *>(p) = v;
  } else if (POLICY == 1) {
    __builtin_nontemporal_store(v, reinterpret_cast<f64x2 *>(p));
  } else if (POLICY == 2) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  } else if (POLICY == 3) {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(p), "v"(v) : "memory");
  } else if (POLICY == 4) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt" ::"v"(p), "v"(v) : "memory");
  } else {
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(p), "v"(v) : "memory");
  }
}

template <int NIN, int NOUT, int POLICY>
__global__ void __launch_bounds__(256) pass_kernel(Streams s, long npairs, double *sink) {
  double red = 0.0;
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < npairs; q += (long)gridDim.x * 256) {
    if (NOUT >= 8) {
      // copy-/form-shaped: output o is a combination of the inputs o, o + NOUT, ... (eight outputs at a time)
      constexpr int R = NOUT > 0 ? NIN / NOUT : 1;
#pragma unroll
      for (int o0 = 0; o0 < NOUT; o0 += 8) {
        f64x2 v[R][8];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
          for (int u = 0; u < 8; u++)
            if (o0 + u < NOUT) v[r][u] = ldnt(s.in[r * NOUT + o0 + u] + 2 * q);
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (o0 + u < NOUT) {
            f64x2 w = v[0][u] * 1.5;
#pragma unroll
            for (int r = 1; r < R; r++) w -= 0.37 * v[r][u];
            st<POLICY>(s.out[o0 + u] + 2 * q, w);
          }
      }
    } else {
      f64x2 acc = (f64x2){0.0, 0.0};
#pragma unroll
      for (int j0 = 0; j0 < NIN; j0 += 8) {
        f64x2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (j0 + u < NIN) v[u] = ldnt(s.in[j0 + u] + 2 * q);
#pragma unroll
        for (int u = 0; u < 8; u++)
          if (j0 + u < NIN) acc += (1.0 + 0.01 * (j0 + u)) * v[u];
      }
#pragma unroll
      for (int i = 0; i < NOUT; i++) st<POLICY>(s.out[i] + 2 * q, acc * (double)(i + 1));
      if (NOUT == 0) red += acc.x + acc.y;
    }
  }
  if (NOUT == 0) {  // read-only mix: reduced in registers, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) red += __shfl_xor(red, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sink + (blockIdx.x & 255), red);
  }
}

static double *g_sink = nullptr;

template <int NIN, int NOUT, int POLICY>
void run(const char *shape, Streams s, long npairs, int bpc) {
  static const char *names[6] = {"plain", "nt", "sc1", "sc0 sc1", "sc1 nt", "sc0 sc1 nt"};
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float best = 1e30f, sum = 0.f;
  const int reps = 6;
  for (int r = 0; r < reps; r++) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((pass_kernel<NIN, NOUT, POLICY>), dim3(256 * bpc), dim3(256), 0, 0, s, npairs, g_sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (r == 0) continue;
    sum += ms;
    if (ms < best) best = ms;
  }
  const double bytes = 16.0 * npairs * (NIN + NOUT);
  printf("%-14s %2d in %2d out  stores %-11s bpc=%d  avg %.3f ms  best %.3f ms  %.0f GB/s (best %.0f)\n", shape, NIN, NOUT,
         names[POLICY], bpc, sum / (reps - 1), best, bytes / (sum / (reps - 1) * 1e-3) * 1e-9,
         bytes / (best * 1e-3) * 1e-9);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
}

template <int NIN, int NOUT>
void sweep(const char *shape, Streams s, long npairs, int bpc) {
  run<NIN, NOUT, 0>(shape, s, npairs, bpc);
  run<NIN, NOUT, 1>(shape, s, npairs, bpc);
  run<NIN, NOUT, 2>(shape, s, npairs, bpc);
  run<NIN, NOUT, 3>(shape, s, npairs, bpc);
  run<NIN, NOUT, 4>(shape, s, npairs, bpc);
  run<NIN, NOUT, 5>(shape, s, npairs, bpc);
}

int main(int argc, char **argv) {
  const long n = argc > 1 ? atol(argv[1]) : 50000000L;
  const long npairs = n / 2;
  Streams s;
  // PROBE_SKEW=<bytes>: stream j starts (j mod 61) * skew bytes into its allocation (do equal offsets of all streams
  // within their 2 MB-aligned allocations cost DRAM bank conflicts?)
  const long skew = getenv("PROBE_SKEW") ? atol(getenv("PROBE_SKEW")) : 0;
  const long pad = 61 * skew + 256;
  // PROBE_ARENA=k: the streams are carved out of slabs of k streams each (stride rounded up to 2 MB); unset / 0: one
  // hipMalloc per stream (what a vector-per-allocation library does)
  const int slab = getenv("PROBE_ARENA") ? atoi(getenv("PROBE_ARENA")) : 0;
  const long stride = ((n * 8 + pad + (2L << 20) - 1) >> 21) << 21;
  char *base = nullptr;
  int used = 0;
  auto take = [&]() -> char * {
    char *p = nullptr;
    if (slab <= 0) {
      if (hipMalloc(&p, n * 8 + pad) != hipSuccess) return nullptr;
      hipMemset(p, 0, n * 8 + pad);
      return p;
    }
    if (!base || used == slab) {
      if (hipMalloc(&base, stride * slab) != hipSuccess) return nullptr;
      hipMemset(base, 0, stride * slab);
      used = 0;
    }
    return base + stride * (used++);
  };
  for (int j = 0; j < MAXS; j++) {
    char *p = take();
    if (!p) return 1;
    s.in[j] = reinterpret_cast<double *>(p + (j % 61) * skew);
  }
  for (int j = 0; j < 32; j++) {
    char *p = take();
    if (!p) return 1;
    s.out[j] = reinterpret_cast<double *>(p + ((j + 37) % 61) * skew);
  }
  printf("# skew %ld bytes per stream index\n", skew);
  hipMalloc(&g_sink, 256 * sizeof(double));
  hipMemset(g_sink, 0, 256 * sizeof(double));
  hipDeviceSynchronize();
  if (argc > 2) {  // the stream mixes of the product's storing passes, `nt` stores only (what the product uses)
    for (int round = 0; round < 3; round++) {
      run<49, 2, 1>("solve2r mix", s, npairs, 2);
      run<42, 4, 1>("kkt_res_update mix", s, npairs, 2);
      run<50, 10, 1>("gram+form mix", s, npairs, 4);
      run<6, 2, 1>("dinv_d1 mix", s, npairs, 4);
      run<4, 2, 1>("trial mix", s, npairs, 4);
      run<48, 1, 1>("(48 in, 1 out)", s, npairs, 2);
      run<33, 0, 1>("mdot<32> mix", s, npairs, 5);
      run<48, 0, 1>("solve2_dots mix", s, npairs, 2);
      run<44, 0, 1>("plain gram mix", s, npairs, 4);
    }
    return 0;
  }
  for (int round = 0; round < 2; round++) {
    sweep<42, 4>("solve-shaped", s, npairs, 2);
    sweep<32, 32>("copy-shaped", s, npairs, 4);
    sweep<20, 10>("form-shaped", s, npairs, 4);
    sweep<1, 1>("one copy", s, npairs, 4);
  }
  return 0;
}
