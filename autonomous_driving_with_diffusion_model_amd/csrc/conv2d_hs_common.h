// Shared by the split-fp16 convolution kernels (conv2d_hs.hip, conv2d_hs16.hip): operand types, the hi / lo split, constants.
#pragma once
#include "adx_common.h"

namespace adx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

constexpr int kHsCout = 64;          // output channels per workgroup
constexpr int kHsCC = 16;            // channels per chunk = K of one MFMA
constexpr float kLoScale = 2048.f;   // 2^11

#ifdef ADX_HS_M16_TIMING
// TIMING-ONLY build (garbage results): every v_mfma_f32_32x32x16_f16 of the pipelined 3x3 kernel issued as two
// v_mfma_f32_16x16x32_f16 on the same operand registers -- the same flops per instruction slot pair; what the smaller shape
// buys before the LDS images are re-laid for it (profiles/README.md, round 5)
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 hs_mfma(f16x8 a, f16x8 b, f32x16 c) {
  f32x4_t c0 = __builtin_shufflevector(c, c, 0, 1, 2, 3), c1 = __builtin_shufflevector(c, c, 4, 5, 6, 7);
  c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
  c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
  c[0] = c0[0]; c[1] = c0[1]; c[2] = c0[2]; c[3] = c0[3];
  c[4] = c1[0]; c[5] = c1[1]; c[6] = c1[2]; c[7] = c1[3];
  return c;
}
#else
__device__ __forceinline__ f32x16 hs_mfma(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
#endif

template <int CTRL>
__device__ __forceinline__ float hs_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// x = hi + lo / 2^11 with hi = fp16(x), lo = fp16((x - hi) * 2^11) (conv2d_hs.hip header)
__device__ __forceinline__ void split8(const float* v, float xs, u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x = v[j] * xs;     // xs is a power of two: exact
    const _Float16 hj = (_Float16)x;
    h[j] = hj;
    l[j] = (_Float16)((x - (float)hj) * kLoScale);
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}

}  // namespace adx
