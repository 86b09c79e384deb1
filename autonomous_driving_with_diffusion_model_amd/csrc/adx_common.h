// Shared device/host helpers for libadx (gfx950 only: wave64, fp32 MFMA).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/adx.h"

namespace adx {

void set_error(const char* fmt, ...);

#define ADX_CHECK_HIP(expr)                                                              \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      adx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return ADX_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

#define ADX_REQUIRE(cond, ...)                                                           \
  do {                                                                                   \
    if (!(cond)) {                                                                       \
      adx::set_error(__VA_ARGS__);                                                       \
      return ADX_ERR_INVALID;                                                            \
    }                                                                                    \
  } while (0)

#define ADX_LAUNCH_CHECK()                                                               \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) {                                                              \
      adx::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); \
      return ADX_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// __builtin_bit_cast(T, vec[i]) on an ext_vector ELEMENT lvalue is miscompiled by this clang (ROCm 7.2: it reads element 0
// whatever i is); an element passed by value is a scalar rvalue and safe -- use these for vector elements.
__device__ __forceinline__ float u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWave = 64;

static inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Mish(x) = x * tanh(softplus(x)), softplus threshold 20 as in torch (modeling/helpers.py:108).
// tanh(log(1+e^x)) = n / (n + 2) with n = e^x (e^x + 2): one exp, one divide, no cancellation.
__device__ __forceinline__ float mish_f(float x) {
  if (x > 20.f) return x;
  const float e = expf(x);
  const float n = e * (e + 2.f);
  return x * (n / (n + 2.f));
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

}  // namespace adx
