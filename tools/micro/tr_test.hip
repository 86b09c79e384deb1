// Semantics probe for ds_read_b64_tr_b16 (gfx950): prints what each lane receives from a [16 rows][16 cols] fp16 matrix.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (_Float16)(float)i;   // value = row * 16 + col
  __syncthreads();
  const int lane = threadIdx.x;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  h4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) h4*)(lds + (4 * g + q) * 16 + 4 * p));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (float)v[j];
}
int main() {
  float* d; hipMalloc(&d, 256 * 4); k<<<1, 64>>>(d); float h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  return 0;
}
