// Sustained fp16 MFMA rate on random operands re-read from LDS: 32x32x16 vs 16x16x32, same flops per loop trip,
// 4 or 8 waves per CU.  Mirrors conv2d_hs's inner loop (8 x ds_read_b128 per 12 x 32x32x16 MFMAs).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ void __launch_bounds__(256, 2) k(const u32x4* in, float* out, int iters) {
  __shared__ u32x4 lds[4096];                      // 64 KB of random fp16 pairs
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = in[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = lds + wave * 64 + lane;
  if (SHAPE == 32) {
    f32x16 acc[8];
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const u32x4* p = base + ((it * 8) & 2047);
      f16x8 f[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = __builtin_bit_cast(f16x8, p[j * 256]);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          acc[r * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m], f[4 + r], acc[r * 2 + m], 0, 0, 0);
          acc[4 + r * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[m], f[6 + r], acc[4 + r * 2 + m], 0, 0, 0);
          acc[4 + r * 2 + m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[2 + m], f[4 + r], acc[4 + r * 2 + m], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    // same flops: 12 x (32x32x16) = 48 x (16x16x32) per trip... keep it at 48 per trip over 32 accumulators
    f32x4 acc[32];
    for (int t = 0; t < 32; ++t) for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
      const u32x4* p = base + ((it * 8) & 2047);
      f16x8 f[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) f[j] = __builtin_bit_cast(f16x8, p[j * 128]);
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          acc[a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[a], f[8 + b], acc[a * 4 + b], 0, 0, 0);
          acc[16 + a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[a], f[12 + b], acc[16 + a * 4 + b], 0, 0, 0);
          acc[16 + a * 4 + b] = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[4 + a], f[8 + b], acc[16 + a * 4 + b], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int t = 0; t < 32; ++t) for (int i = 0; i < 4; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

int main() {
  u32x4* in; float* out;
  hipMalloc(&in, 4096 * 16); hipMalloc(&out, 2048 * 256 * 4);
  unsigned short* h = (unsigned short*)malloc(4096 * 16);
  srand(1);
  for (int i = 0; i < 4096 * 8; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0FFF) + ((rand() & 1) << 15));  // random halves ~[0.1, 1)
  hipMemcpy(in, h, 4096 * 16, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wg = 256; wg <= 512; wg += 256)
    for (int shape = 32; shape >= 16; shape -= 16) {
      const int iters = 4000;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 32) k<32><<<wg, 256>>>(in, out, iters); else k<16><<<wg, 256>>>(in, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)wg * 4 * iters * (shape == 32 ? 12 * 32768.0 : 48 * 16384.0);
      printf("%s  %d workgroups (%d per CU): %.3f ms  %.0f TFLOP/s\n", shape == 32 ? "32x32x16" : "16x16x32", wg, wg / 256, ms, flops / ms / 1e9);
    }
  return 0;
}
