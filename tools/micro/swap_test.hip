#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned* p) {
  unsigned a = 1000 + threadIdx.x, b = 2000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  p[threadIdx.x] = r[0];
  p[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 512);
  k<<<1, 64>>>(d);
  unsigned h[128]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
  printf("r0: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32], h[63]);
  printf("r1: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[64], h[95], h[96], h[127]);
  return 0;
}
