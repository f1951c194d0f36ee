// Probes the operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 on gfx950 with one-hot inputs:
// for every (A lane la, B lane lb) the lanes of D that become non-zero are printed as a compact table.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_layout.hip -o tools/mfma_f64_layout
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(int *hit) {  // hit[la*64+lb] = bitmask-free: the first D lane that is non-zero, count in high bits
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; la++) {
    for (int lb = 0; lb < 64; lb++) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) hit[la * 64 + lb] = m ? (__ffsll((long long)m) - 1) | (__popcll(m) << 8) : -1;
    }
  }
}
int main() {
  int *d, h[4096];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("rows: A lane, cols: B lane, entry: D lane hit (.. = none); popcount>1 marked '*'\n");
  for (int la = 0; la < 64; la++) {
    printf("A%02d:", la);
    for (int lb = 0; lb < 64; lb++) {
      if (h[la * 64 + lb] < 0) printf(" ..");
      else printf(" %02d%s", h[la * 64 + lb] & 255, (h[la * 64 + lb] >> 8) > 1 ? "*" : "");
    }
    printf("\n");
  }
  return 0;
}
