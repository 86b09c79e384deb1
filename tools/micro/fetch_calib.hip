// Calibration of rocprofv3's FETCH_SIZE for the access widths conv2d_hs uses: a known number of bytes is streamed
// once with 4-byte-per-lane loads (the patch staging) and once with 16-byte-per-lane loads (the weight slabs).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void read4(const float* __restrict__ x, float* out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += x[i];
  if (s == 123.456f) out[0] = s;
}
__global__ void read16(const f4* __restrict__ x, float* out, size_t n4) {
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) s += x[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = s[0];
}
int main() {
  const size_t bytes = (size_t)1 << 30;       // 1 GiB, far beyond L2 + Infinity Cache
  float *x, *out;
  hipMalloc(&x, bytes); hipMalloc(&out, 4);
  hipMemset(x, 0, bytes);
  hipDeviceSynchronize();
  read4<<<4096, 256>>>(x, out, bytes / 4);
  hipDeviceSynchronize();
  read16<<<4096, 256>>>((const f4*)x, out, bytes / 16);
  hipDeviceSynchronize();
  printf("streamed %zu bytes per kernel\n", bytes);
  return 0;
}
