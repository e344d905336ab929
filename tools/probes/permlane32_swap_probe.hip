// What does __builtin_amdgcn_permlane32_swap(a, b, false, false) return on gfx950?  a = lane, b = 100 + lane.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned lane = threadIdx.x;
  const auto r = __builtin_amdgcn_permlane32_swap(lane, 100u + lane, false, false);
  o[lane] = r[0];
  o[64 + lane] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]: lane0 %u lane1 %u lane31 %u | lane32 %u lane33 %u lane63 %u\n", h[0], h[1], h[31], h[32], h[33], h[63]);
  printf("r[1]: lane0 %u lane1 %u lane31 %u | lane32 %u lane33 %u lane63 %u\n", h[64], h[65], h[95], h[96], h[97], h[127]);
  return 0;
}
