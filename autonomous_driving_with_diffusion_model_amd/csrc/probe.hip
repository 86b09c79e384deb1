// Measurement probe, not part of the hot path: the fp16 MFMA rate this chip SUSTAINS, with the operand traffic of the 3x3
// convolution's inner loop (8 x ds_read_b128 per 12 x v_mfma_f32_32x32x16_f16, conv2d_hs.hip) and nothing else -- no global
// memory, no staging, no epilogue.  bench.py times it next to the convolution so that `roofline` can state the ceiling the
// power limit leaves (the datasheet peak assumes the boost clock; under matrix load the shader clock settles far below it,
// and the same instruction stream runs faster on all-zero operands than on random ones).
#include "adx_common.h"

namespace adx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256, 2) probe_mfma_kernel(const u32x4* __restrict__ operands, float* __restrict__ out, int iters) {
  __shared__ u32x4 lds[4096];                      // 64 KB of fp16 operand cells
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = operands[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = lds + wave * 64 + lane;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
  for (int it = 0; it < iters; ++it) {
    const u32x4* p = base + ((it * 8) & 2047);
    f16x8 f[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = __builtin_bit_cast(f16x8, p[j * 256]);
    // hi*hi, hi*lo, lo*hi of a 2 x 2 block of 32 x 32 tiles: the product scheme of the convolution
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
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same flops per trip and the same LDS reads per flop on v_mfma_f32_16x16x32_f16 (conv2d_hs16.hip's loop: 16 operand reads per
// 48 MFMAs, a 4 x 4 block of 16 x 16 tiles per wave)
__global__ void __launch_bounds__(256, 2) probe_mfma16_kernel(const u32x4* __restrict__ operands, float* __restrict__ out, int iters) {
  __shared__ u32x4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = operands[i];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const u32x4* base = lds + wave * 64 + lane;
  f32x4 acc[32];
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = 0.f;
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
#pragma unroll
  for (int t = 0; t < 32; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[t][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

}  // namespace adx

extern "C" int adx_probe_mfma_fp16_16x16x32(const void* operands, float* out, int32_t workgroups, int32_t iters, double* flops,
                                            adx_stream stream) {
  ADX_REQUIRE(operands && out && workgroups > 0 && iters > 0, "adx_probe_mfma_fp16_16x16x32: bad argument");
  adx::probe_mfma16_kernel<<<dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream>>>(
      reinterpret_cast<const adx::u32x4*>(operands), out, iters);
  ADX_LAUNCH_CHECK();
  if (flops != nullptr) *flops = (double)workgroups * 4 * iters * 48 * 16384.0;     // 48 MFMAs of 2 * 16 * 16 * 32 per wave and trip
  return ADX_OK;
}

extern "C" int adx_probe_mfma_fp16(const void* operands, float* out, int32_t workgroups, int32_t iters, double* flops,
                                   adx_stream stream) {
  ADX_REQUIRE(operands && out && workgroups > 0 && iters > 0, "adx_probe_mfma_fp16: bad argument");
  adx::probe_mfma_kernel<<<dim3((unsigned)workgroups), dim3(256), 0, (hipStream_t)stream>>>(
      reinterpret_cast<const adx::u32x4*>(operands), out, iters);
  ADX_LAUNCH_CHECK();
  if (flops != nullptr) *flops = (double)workgroups * 4 * iters * 12 * 32768.0;     // 12 MFMAs of 2 * 32 * 32 * 16 per wave and trip
  return ADX_OK;
}
