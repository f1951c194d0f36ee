// Tuning aid: what does a host-synchronising exchange cost, and what would polling a flag in pinned host memory save?
//   (a) tiny kernel writes its result into pinned host memory, host calls hipStreamSynchronize  (what the library does)
//   (b) the same kernel also writes a sequence number behind a system-scope fence, host spins on it
//   (c) like (b), with a 4-workgroup kernel whose last workgroup (device ticket) raises the flag
// Each timed loop = launch + wait + a dependent second launch (the next kernel of the iteration), 2000 rounds.
//   hipcc --offload-arch=gfx950 -O2 -o tools/sync_probe tools/sync_probe.hip && ./tools/sync_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_plain(double *out, double v) { if (threadIdx.x == 0) out[blockIdx.x] = v + blockIdx.x; }
__global__ void k_flag(double *out, double v, volatile unsigned long long *flag, unsigned long long seq) {
  if (threadIdx.x == 0) {
    out[0] = v;
    __threadfence_system();
    *flag = seq;
  }
}
__global__ void k_flag4(double *out, double v, volatile unsigned long long *flag, unsigned long long seq, unsigned *ticket) {
  if (threadIdx.x == 0) {
    out[blockIdx.x] = v + blockIdx.x;
    __threadfence_system();
    const unsigned t = atomicAdd(ticket, 1u);
    if (t == gridDim.x - 1) {
      *ticket = 0;
      __threadfence_system();
      *flag = seq;
    }
  }
}
__global__ void k_next(double *d, double v) { if (threadIdx.x == 0) d[0] = v; }
int main() {
  hipStream_t s;
  CK(hipStreamCreate(&s));
  double *h, *hd, *d;
  unsigned long long *f, *fd;
  unsigned *ticket;
  CK(hipHostMalloc((void **)&h, 64 * sizeof(double), hipHostMallocDefault));
  CK(hipHostGetDevicePointer((void **)&hd, h, 0));
  CK(hipHostMalloc((void **)&f, 64, hipHostMallocDefault));
  CK(hipHostGetDevicePointer((void **)&fd, f, 0));
  CK(hipMalloc(&d, 64));
  CK(hipMalloc(&ticket, 4));
  CK(hipMemset(ticket, 0, 4));
  *f = 0;
  const int R = 2000;
  for (int mode = 0; mode < 3; mode++) {
    for (int rep = 0; rep < 2; rep++) {
      auto t0 = std::chrono::steady_clock::now();
      for (int i = 1; i <= R; i++) {
        const unsigned long long seq = (unsigned long long)(mode * 2 + rep) * 100000ull + i;
        if (mode == 0) {
          hipLaunchKernelGGL(k_plain, dim3(4), dim3(256), 0, s, hd, (double)i);
          CK(hipStreamSynchronize(s));
        } else if (mode == 1) {
          hipLaunchKernelGGL(k_flag, dim3(1), dim3(256), 0, s, hd, (double)i, fd, seq);
          while (*(volatile unsigned long long *)f != seq) { }
        } else {
          hipLaunchKernelGGL(k_flag4, dim3(4), dim3(256), 0, s, hd, (double)i, fd, seq, ticket);
          while (*(volatile unsigned long long *)f != seq) { }
        }
        if (h[0] != (double)i) { printf("stale value in mode %d\n", mode); return 1; }
        hipLaunchKernelGGL(k_next, dim3(1), dim3(64), 0, s, d, h[0]);
      }
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
      printf("mode %d (%s) rep %d: %.2f us per round (launch + wait + dependent launch)\n", mode,
             mode == 0 ? "hipStreamSynchronize" : mode == 1 ? "flag, 1 workgroup" : "flag, 4 workgroups + ticket", rep, us);
    }
  }
  return 0;
}
