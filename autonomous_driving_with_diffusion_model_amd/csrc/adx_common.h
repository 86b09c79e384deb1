// Shared device/host helpers for libadx (gfx950 only: wave64, fp32 MFMA).
#pragma once
#include <atomic>
#include <mutex>
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

// Debug switches of the library: the environment is parsed ONCE, here, into this struct (api.cpp).  Every one of them is an
// A/B or diagnostic configuration that a tracked test or tool runs (INTEGRATION.md lists them); there are no others.
struct DebugSwitches {
  bool conv_exact = false;     // ADX_CONV_EXACT=1   every 2-D conv (and weight gradient) on the exact-fp32 MFMA kernels
  bool wgrad_exact = false;    // ADX_WGRAD_EXACT=1  the 2-D weight gradients only
  bool tconv_exact = false;    // ADX_TCONV_EXACT=1  the temporal stack on the exact-fp32 MFMA kernel
  bool unet_chain = true;      // ADX_UNET_CHAIN=0   every temporal level layer by layer (no chained launches)
  unsigned chain_mask = ~0u;   // ADX_CHAIN_MASK=<bits>  which levels are chained (bit i: down level i, bit 8 + i: up level i)
  bool unet_pipe = true;       // ADX_UNET_PIPE=0    the deepest level's layer run as launches, not as one pipeline launch (tconv_pipe.hip)
  bool conv_vrow = true;       // ADX_CONV_VROW=0    fp32-layout 3x3 launches tile every image on its own (no virtual row over the batch)
  bool conv_cells = true;      // ADX_CONV_CELLS=0   fp32 NCHW between all perception convs (no pre-split cell tensors)
  int train_cells = 5;         // ADX_TRAIN_CELLS=0  the training executor keeps every tensor as fp32 NCHW and no mask bits; =1 only what changes
                               //                    no bit (the map between a block's convs as cells, the ReLU mask as bits, the identity
                               //                    gradient added through the bits); =2 the blocks' outputs as cells too (the next block's
                               //                    identity is then hi + lo / 2^11); =3 + the 16x16x32 kernel for the forward launches it
                               //                    tiles (conv2d_hs16.hip: TRAIN); default 4: + the conv-output gradients of the stride-1
                               //                    3x3 convs as cells under a scale taken from a bound (bn_bwd_apply_groups_kernel);
                               //                    5 (default): + their data gradients with Cout % 128 == 0 on the 16x16x32 kernel (TRAIN == 2)
  int hs_mode = -1;            // ADX_HS_MODE=0|1|2  pins the tile mode of the pipelined 3x3 kernel
  bool wgrad_deterministic = false;   // ADX_WGRAD_DETERMINISTIC=1  the 3x3 weight gradients (conv2d_wgrad_hs) reduce per-workgroup partial
                               //                    sums in index order instead of with float atomics: bit-reproducible, one more pass
  bool hs_dma = true;          // ADX_HS_DMA=0       conv2d_hs3x3q stages its operands through registers instead of with LDS-DMA loads (conv2d_hs16.hip)
  bool hs_persist = true;      // ADX_HS_PERSIST=0   conv2d_hs3x3q launches one workgroup per tile instead of one per CU that walks the tiles
  int resnet_streams = 2;      // ADX_RESNET_STREAMS=1..4  sub-batches of an inference perception pass at B >= 32, each on a stream of its own
  int resnet_split_from = -1;  // ADX_RESNET_SPLIT_FROM=<block>  first BasicBlock that runs per sub-batch (default: the first downsample block; 0: the stem too)
  bool check_range = false;    // ADX_CHECK_RANGE=1  perception forward: fail with the first layer whose activations leave the
                               //                    fp16 range of the split kernels instead of propagating inf (synchronises)
};
const DebugSwitches& debug_switches();

// "once per device" for per-device state such as hipFuncSetAttribute(MaxDynamicSharedMemorySize): a process-wide flag would
// leave a second GPU of the same process without the attribute (one process per GPU is the deployment, but not a rule).
//   if (DeviceOnce once{flag}; once) { ADX_CHECK_HIP(hipFuncSetAttribute(...)); ...; once.commit(); }
// The body runs under a mutex the first time the calling thread's current device meets `flag` (devices folded modulo 64) and
// the device's bit is published only by commit(), i.e. after every attribute call has succeeded: a second host thread on the
// same device either waits for the setup or finds it done -- it never launches in between --, and a setup that failed (the
// ADX_CHECK_HIP in the body returned) is retried by the next call instead of leaving every later launch to fail.
struct DeviceOnce {
  std::atomic<uint64_t>& done;
  uint64_t bit = 0;
  bool owner = false;
  static std::mutex& mu() { static std::mutex m; return m; }
  explicit DeviceOnce(std::atomic<uint64_t>& d) : done(d) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    bit = 1ull << (dev & 63);
    if (done.load(std::memory_order_acquire) & bit) return;
    mu().lock();
    if (done.load(std::memory_order_relaxed) & bit) { mu().unlock(); return; }
    owner = true;
  }
  ~DeviceOnce() { if (owner) mu().unlock(); }
  DeviceOnce(const DeviceOnce&) = delete;
  DeviceOnce& operator=(const DeviceOnce&) = delete;
  explicit operator bool() const { return owner; }
  void commit() { done.fetch_or(bit, std::memory_order_release); }
};

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
