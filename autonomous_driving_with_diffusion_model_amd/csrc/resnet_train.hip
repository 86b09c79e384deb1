// Training-mode perception: ResNet-34 forward with batch-statistics BatchNorm (running buffers updated
// like nn.BatchNorm2d, momentum 0.1) that keeps what the backward needs, and the backward itself
// (training step T1, train.py:242-251; modeling/resnet.py:87-102,277-293).
//
//   forward, per conv:   raw = conv2d(x)                         (the inference MFMA kernel, no epilogue)
//                        per-channel sum / sum-of-squares         (fp64 accumulation, one pass)
//                        out = relu(raw * scale + shift [+ identity])
//   backward, per conv:  dz = dout * (out > 0); sums of dz and dz*xhat per channel (fp64)
//                        draw = gamma*rstd * (dz - mean(dz) - xhat*mean(dz*xhat))
//                        dW   = conv2d_wgrad(x, draw)             (MFMA, pixels as K, atomically reduced)
//                        dx   = conv2d(draw, W flipped/transposed) -- the same forward kernel;
//                               stride-2 convs go through a zero-dilated draw (3 of 36 layers)
#include <algorithm>

#include "batch_ops.h"
#include "conv2d_internal.h"
#include "conv2d_hs_common.h"

namespace adx {

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// BatchNorm's affine form of one channel, evaluated identically in the forward apply pass and wherever the backward
// pass re-derives the ReLU mask from the conv output (explicit fma: no dependence on the compiler's contraction)
__device__ __forceinline__ void bn_affine(float gamma, float beta, float mean, float rstd, float& scale, float& shift) {
  scale = gamma * rstd;
  shift = __builtin_fmaf(-mean, scale, beta);
}
__device__ __forceinline__ float bn_eval(float raw, float scale, float shift) { return __builtin_fmaf(raw, scale, shift); }

// sums[c][0] += sum a,  sums[c][1] += sum a*a          (MODE 0, BatchNorm forward statistics)
// sums[c][0] += sum dz, sums[c][1] += sum dz*xhat      (MODE 1, BatchNorm backward), dz = dout * ReLU mask
// relu_mask: 0 none, 1 read from `out` (ReLU after the residual add), 2 re-derived from the conv output (ReLU
// directly after BN: out > 0 <=> bn_eval(raw) > 0, one tensor read less)
template <int MODE>
__global__ void __launch_bounds__(256) channel_sums_kernel(const float* __restrict__ a, const float* __restrict__ out,
                                                            const float* __restrict__ raw, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, double* __restrict__ sums,
                                                            int C, int HW, int relu_mask, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, const uint8_t* __restrict__ bits = nullptr,
                                                            uint32_t* __restrict__ dmax = nullptr) {
  // dmax (MODE 1): dmax[c] = atomic maximum of |dz| over channel c, as bits (bn_bwd_consts_kernel bounds the range of draw with
  // the largest of them; one slot per channel: a single word would serialise batch x C atomics)
  // bits (relu_mask == 1): the forward's mask, one bit per element -- byte [n][c / 8][pixel], bit c % 8 (bn_apply_groups_kernel) --
  // read instead of `out`
  const int plane = blockIdx.x;            // n * C + c
  const int c = plane % C, tid = threadIdx.x;
  const size_t base = (size_t)plane * HW;
  const uint8_t* bp = bits != nullptr ? bits + ((size_t)(plane / C) * (C >> 3) + (c >> 3)) * HW : nullptr;
  const int bit = c & 7;
  double s0 = 0.0, s1 = 0.0;
  float vmax = 0.f;
  const float mu = MODE == 1 ? mean[c] : 0.f, rs = MODE == 1 ? rstd[c] : 0.f;
  float sc = 0.f, sh = 0.f;
  if (MODE == 1 && relu_mask == 2) bn_affine(gamma[c], beta[c], mu, rs, sc, sh);
  auto one = [&](float v, float rw, float o) {
    if (MODE == 0) {
      s0 += v;
      s1 += (double)v * v;
    } else {
      if (relu_mask == 1 && !(o > 0.f)) v = 0.f;
      if (relu_mask == 2 && !(bn_eval(rw, sc, sh) > 0.f)) v = 0.f;
      vmax = __builtin_fmaxf(vmax, __builtin_fabsf(v));
      s0 += v;
      s1 += (double)v * (double)((rw - mu) * rs);
    }
  };
  const bool use_bits = MODE == 1 && relu_mask == 1 && bp != nullptr;
  const bool vec = (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | (MODE == 1 ? reinterpret_cast<uintptr_t>(raw) : 0) |
                                      (MODE == 1 && relu_mask == 1 && !use_bits ? reinterpret_cast<uintptr_t>(out) : 0)) & 15) == 0 &&
                   (!use_bits || (reinterpret_cast<uintptr_t>(bits) & 3) == 0);
  if (vec) {     // 16-byte loads: a plane starts on a 16-byte boundary when HW % 4 == 0
    const f32x4* a4 = reinterpret_cast<const f32x4*>(a + base);
    const f32x4* r4 = reinterpret_cast<const f32x4*>(raw + base);
    const f32x4* o4 = reinterpret_cast<const f32x4*>(out + base);
    for (int i = tid; i < (HW >> 2); i += 256) {
      const f32x4 v = a4[i];
      f32x4 rw = f32x4{0.f, 0.f, 0.f, 0.f}, o = f32x4{0.f, 0.f, 0.f, 0.f};
      if (MODE == 1) rw = r4[i];
      if (use_bits) {
        const uint32_t m = reinterpret_cast<const uint32_t*>(bp)[i] >> bit;       // four pixels' bytes
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (m >> (8 * k)) & 1u ? 1.f : 0.f;
      } else if (MODE == 1 && relu_mask == 1) o = o4[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) one(v[k], rw[k], o[k]);
    }
  } else {
    for (int i = tid; i < HW; i += 256)
      one(a[base + i], MODE == 1 ? raw[base + i] : 0.f,
          use_bits ? (float)((bp[i] >> bit) & 1) : (MODE == 1 && relu_mask == 1 ? out[base + i] : 0.f));
  }
  __shared__ double red[8];
  s0 = wave_sum_d(s0);
  s1 = wave_sum_d(s1);
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = s0; red[(tid >> 6) * 2 + 1] = s1; }
  __syncthreads();
  if (tid == 0) {
    atomicAdd(sums + 2 * c, (red[0] + red[2]) + (red[4] + red[6]));
    atomicAdd(sums + 2 * c + 1, (red[1] + red[3]) + (red[5] + red[7]));
  }
  if (MODE == 1 && dmax != nullptr) {
    __shared__ float mred[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) vmax = __builtin_fmaxf(vmax, __shfl_xor(vmax, off, 64));
    if ((tid & 63) == 0) mred[tid >> 6] = vmax;
    __syncthreads();
    if (tid == 0) {
      const float m = __builtin_fmaxf(__builtin_fmaxf(mred[0], mred[1]), __builtin_fmaxf(mred[2], mred[3]));
      if (m > 0.f) atomicMax(dmax + c, f2u(m));       // (non-negative floats order like their bit patterns)
    }
  }
}

// The pipelined 3x3 conv leaves per-workgroup partial sums of its output and of its squares ([C][2][P] floats, conv2d_hs.hip:
// STATS); this adds the P partials of every (channel, moment) in fp64 -- one workgroup each, fixed order -- into the same
// sums[c][2] slots channel_sums_kernel<0> would have filled from a pass over the whole output tensor.
// One launch per conv: workgroup c adds both moments' partials of channel c in fp64 (fixed order: thread i of a moment's 128
// takes partials i, i + 128, ...) and finishes the channel (what a reduce launch + a finalize launch did before)
__device__ __forceinline__ void bn_finalize_channel(int c, double s0, double s1, const float* gamma, const float* beta, float* scale,
                                                    float* shift, float* mean, float* rstd, float* running_mean, float* running_var,
                                                    double count);
__global__ void __launch_bounds__(256) stats_reduce_finalize_kernel(const float* __restrict__ part, double* __restrict__ sums, int P,
                                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                     float* __restrict__ scale, float* __restrict__ shift,
                                                                     float* __restrict__ mean, float* __restrict__ rstd,
                                                                     float* running_mean, float* running_var, double count) {
  const int c = blockIdx.x, moment = threadIdx.x >> 7, t = threadIdx.x & 127;
  const float* src = part + (size_t)(2 * c + moment) * P;
  double s = 0.0;
  for (int i = t; i < P; i += 128) s += (double)src[i];
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s0 = red[0] + red[1], s1 = red[2] + red[3];
    sums[2 * c] = s0;
    sums[2 * c + 1] = s1;
    bn_finalize_channel(c, s0, s1, gamma, beta, scale, shift, mean, rstd, running_mean, running_var, count);
  }
}

// the same reduction without the forward's finalisation: the partial sums are those of a BatchNorm BACKWARD (sum dz, sum dz
// xhat), left by the data-gradient conv that produced dz's tensor (conv2d_hs.hip: STATS == 2)
// Workgroup C (one past the channels): the maximum of the launch's per-(slab, tile) max |dz| partials, behind the sums in `part`
__global__ void __launch_bounds__(256) stats_reduce_kernel(const float* __restrict__ part, double* __restrict__ sums, int P,
                                                            uint32_t* __restrict__ dmax = nullptr, int C = 0) {
  if ((int)blockIdx.x == C && dmax != nullptr) {
    const float* src = part + (size_t)2 * C * P;
    float m = 0.f;
    for (int i = threadIdx.x; i < (C / 64) * P; i += 256) m = __builtin_fmaxf(m, src[i]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = __builtin_fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float mred[4];
    if ((threadIdx.x & 63) == 0) mred[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) *dmax = f2u(__builtin_fmaxf(__builtin_fmaxf(mred[0], mred[1]), __builtin_fmaxf(mred[2], mred[3])));
    return;
  }
  const int c = blockIdx.x, moment = threadIdx.x >> 7, t = threadIdx.x & 127;
  const float* src = part + (size_t)(2 * c + moment) * P;
  double s = 0.0;
  for (int i = t; i < P; i += 128) s += (double)src[i];
  __shared__ double red[4];
  s = wave_sum_d(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    sums[2 * c] = red[0] + red[1];
    sums[2 * c + 1] = red[2] + red[3];
  }
}

// batch statistics -> scale/shift for the apply pass, saved mean/rstd, running-buffer update
__global__ void bn_finalize_kernel(const double* __restrict__ sums, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ scale, float* __restrict__ shift,
                                   float* __restrict__ mean, float* __restrict__ rstd, float* running_mean,
                                   float* running_var, int C, double count) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  bn_finalize_channel(c, sums[2 * c], sums[2 * c + 1], gamma, beta, scale, shift, mean, rstd, running_mean, running_var, count);
}

__device__ __forceinline__ void bn_finalize_channel(int c, double s0, double s1, const float* gamma, const float* beta, float* scale,
                                                    float* shift, float* mean, float* rstd, float* running_mean, float* running_var,
                                                    double count) {
  const double m = s0 / count;
  double var = s1 / count - m * m;
  if (var < 0.0) var = 0.0;
  const float r = (float)(1.0 / sqrt(var + 1e-5));
  float sc, sh;
  bn_affine(gamma[c], beta[c], (float)m, r, sc, sh);
  scale[c] = sc;
  shift[c] = sh;
  mean[c] = (float)m;
  rstd[c] = r;
  if (running_mean != nullptr) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = 0.9f * running_mean[c] + 0.1f * (float)m;
    running_var[c] = 0.9f * running_var[c] + 0.1f * (float)unbiased;
  }
}

__global__ void __launch_bounds__(256) bn_apply_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ res,
                                                        float* __restrict__ out, int C, int HW, size_t total, int relu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  // 32-bit channel arithmetic whenever the tensor allows it: a 64-bit divide per element costs more than the loads
  const int c = total <= 0xFFFFFFFFull ? (int)(((uint32_t)i / (uint32_t)HW) % (uint32_t)C) : (int)((i / HW) % C);
  float v = bn_eval(raw[i], scale[c], shift[c]);
  if (res != nullptr) v += res[i];
  if (relu) v = v > 0.f ? v : 0.f;
  out[i] = v;
}

// plane kernels apply when a plane is a whole number of 16-byte quads and every tensor involved starts on one
static bool bn_planes_ok(int HW, const void* a, const void* b, const void* c) {
  return (HW & 3) == 0 && ((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(c)) & 15) == 0;
}

// The same pass for maps with HW % 4 == 0 (every ResNet-34 map at 256x900): one wave per (image, channel) plane, the
// channel's constants in scalars, 16-byte loads and stores, no per-element index arithmetic.  Workgroup w takes the
// planes 4w .. 4w+3, 4w + 4 gridDim ...
__global__ void __launch_bounds__(256) bn_apply_planes_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const float* __restrict__ res,
                                                               float* __restrict__ out, int C, int HW, int planes, int relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hw4 = HW >> 2;
  for (int pl = blockIdx.x * 4 + wave; pl < planes; pl += gridDim.x * 4) {
    const int c = pl % C;
    const float sc = scale[c], sh = shift[c];
    const size_t base = (size_t)pl * hw4;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(raw) + base;
    const f32x4* i4 = reinterpret_cast<const f32x4*>(res) + base;
    f32x4* o4 = reinterpret_cast<f32x4*>(out) + base;
    for (int i = lane; i < hw4; i += 64) {
      const f32x4 rw = r4[i];
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = bn_eval(rw[k], sc, sh);
      if (res != nullptr) {
        const f32x4 id = i4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += id[k];
      }
      if (relu) {
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      }
      o4[i] = v;
    }
  }
}

// BatchNorm apply [+ identity] [+ ReLU] by 8-CHANNEL GROUPS: a thread owns one pixel of one group of 8 channels (eight coalesced
// plane reads per fp32 operand).  What the group form buys:
//  * OUT_CELLS: the output as a CELL tensor -- per image [C/8][hi, lo][H][W] 16-byte cells of 8 channels, the pre-split operand
//    layout of the pipelined 3x3 kernel (conv2d_hs.hip: XCELLS) and of the weight gradient (conv2d_wgrad_hs.hip: XC) -- for an
//    activation only convolutions read: the map between a BasicBlock's two convs (modeling/resnet.py:87-93).  The halves are the
//    ones those kernels' staging would compute from the fp32 map (split8), so nothing downstream changes by a bit; the
//    conversion happens once here instead of once per workgroup column of the consumer.  Same bytes as the fp32 map.
//  * BITS: the ReLU mask of a block's output (ReLU after the residual add, modeling/resnet.py:99-100) as one BIT per element --
//    byte [n][c / 8][pixel], bit c % 8 -- so that the backward pass reads 1/32 of the bytes of `out` to know where the ReLU let
//    the gradient through (bn_bwd_apply*, channel_sums_kernel<1>, the data-gradient epilogue of conv2d_hs.hip).
// RES: 0 no identity, 1 fp32 NCHW, 2 a cell tensor (hi + lo / 2^11).
template <bool OUT_CELLS, int RES, bool BITS>
__global__ void __launch_bounds__(256) bn_apply_groups_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, const void* __restrict__ res,
                                                               void* __restrict__ out, uint8_t* __restrict__ bits, int C, int HW,
                                                               int groups, int relu) {
  const int per = (HW + 255) / 256;                 // workgroups per (image, channel group)
  for (int w = blockIdx.x; w < groups * per; w += gridDim.x) {
    const int g = w / per, pix = (w - g * per) * 256 + threadIdx.x;      // g = n * (C / 8) + channel group
    if (pix >= HW) continue;
    const int c0 = (g % (C >> 3)) * 8;
    const size_t p0 = (size_t)g * 8 * HW + pix;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = raw[p0 + (size_t)j * HW];
    float id[8];
    if constexpr (RES == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) id[j] = reinterpret_cast<const float*>(res)[p0 + (size_t)j * HW];
    } else if constexpr (RES == 2) {
      const u32x4* rc = reinterpret_cast<const u32x4*>(res) + (size_t)g * 2 * HW + pix;
      const f16x8 h = __builtin_bit_cast(f16x8, rc[0]), l = __builtin_bit_cast(f16x8, rc[HW]);
#pragma unroll
      for (int j = 0; j < 8; ++j) id[j] = __builtin_fmaf((float)l[j], 1.f / kLoScale, (float)h[j]);
    }
    uint32_t m = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      v[j] = bn_eval(v[j], scale[c0 + j], shift[c0 + j]);
      if constexpr (RES != 0) v[j] += id[j];
      if (BITS && v[j] > 0.f) m |= 1u << j;
      if (relu) v[j] = v[j] > 0.f ? v[j] : 0.f;
    }
    if constexpr (OUT_CELLS) {
      u32x4 hi, lo;
      split8(v, 1.f, hi, lo);
      u32x4* o = reinterpret_cast<u32x4*>(out) + (size_t)g * 2 * HW + pix;
      o[0] = hi;
      o[HW] = lo;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) reinterpret_cast<float*>(out)[p0 + (size_t)j * HW] = v[j];
    }
    if constexpr (BITS) bits[(size_t)g * HW + pix] = (uint8_t)m;
  }
}

constexpr size_t kStatsPartFloats = (size_t)1 << 20;   // per-workgroup partial sums of one conv launch ([C][2][tiles])
constexpr size_t kAmaxPartials = 4096;   // workgroups of bn_bwd_apply_kernel = partial maxima handed to the dgrad conv

// draw = gamma*rstd * (dz - m1 - xhat*m2); also d gamma / d beta (one thread per channel does that part)
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                            const float* __restrict__ raw, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                            const double* __restrict__ sums, float* __restrict__ draw,
                                                            float* __restrict__ dz_out, int C, int HW, size_t total,
                                                            double count, int relu_mask, uint32_t* __restrict__ amax,
                                                            const float* __restrict__ beta, float* __restrict__ dgamma,
                                                            float* __restrict__ dbeta, const uint8_t* __restrict__ bits = nullptr) {
  // grid-stride: a fixed number of workgroups, each leaving max |draw| of its share in amax[blockIdx.x] (bits:
  // monotonic for non-negative floats); the data-gradient conv reduces those partials for its dynamic range
  __shared__ uint32_t red[4];
  if (blockIdx.x == 0 && dgamma != nullptr)          // the affine parameters' gradients are the finished sums themselves
    for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] = (float)sums[2 * c]; dgamma[c] = (float)sums[2 * c + 1]; }
  uint32_t b = 0;
  const bool small = total <= 0xFFFFFFFFull;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = small ? (int)(((uint32_t)i / (uint32_t)HW) % (uint32_t)C) : (int)((i / HW) % C);
    float dz = dout[i];
    const float rw = raw[i];
    if (relu_mask == 1) {
      bool on;
      if (bits != nullptr) {        // byte [n][c / 8][pixel], bit c % 8
        const size_t pl = i / HW;
        on = (bits[((pl / C) * (C >> 3) + (c >> 3)) * HW + (i - pl * HW)] >> (c & 7)) & 1;
      } else {
        on = out[i] > 0.f;
      }
      if (!on) dz = 0.f;
    }
    if (relu_mask == 2) {
      float sc, sh;
      bn_affine(gamma[c], beta[c], mean[c], rstd[c], sc, sh);
      if (!(bn_eval(rw, sc, sh) > 0.f)) dz = 0.f;
    }
    if (dz_out != nullptr) dz_out[i] = dz;
    const float xh = (rw - mean[c]) * rstd[c];
    const float m1 = (float)(sums[2 * c] / count), m2 = (float)(sums[2 * c + 1] / count);
    const float v = gamma[c] * rstd[c] * (dz - m1 - xh * m2);
    draw[i] = v;
    const uint32_t vb = __builtin_bit_cast(uint32_t, v) & 0x7FFFFFFFu;
    b = vb > b ? vb : b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
    amax[blockIdx.x] = m01 > m23 ? m01 : m23;
  }
}

// plane form of bn_bwd_apply_kernel (HW % 4 == 0): one wave per plane, channel constants (including the two fp64
// divisions) once per plane, 16-byte accesses; amax[blockIdx.x] as above
__global__ void __launch_bounds__(256) bn_bwd_apply_planes_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                   const float* __restrict__ raw, const float* __restrict__ mean,
                                                                   const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                   const double* __restrict__ sums, float* __restrict__ draw,
                                                                   float* __restrict__ dz_out, int C, int HW, int planes,
                                                                   double count, int relu_mask, uint32_t* __restrict__ amax,
                                                                   const float* __restrict__ beta, float* __restrict__ dgamma,
                                                                   float* __restrict__ dbeta, const uint8_t* __restrict__ bits = nullptr) {
  __shared__ uint32_t red[4];
  if (blockIdx.x == 0 && dgamma != nullptr)
    for (int c = threadIdx.x; c < C; c += 256) { dbeta[c] = (float)sums[2 * c]; dgamma[c] = (float)sums[2 * c + 1]; }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int hw4 = HW >> 2;
  uint32_t b = 0;
  for (int pl = blockIdx.x * 4 + wave; pl < planes; pl += gridDim.x * 4) {
    const int c = pl % C;
    const float mu = mean[c], rs = rstd[c], ga = gamma[c];
    float sc = 0.f, sh = 0.f;
    if (relu_mask == 2) bn_affine(ga, beta[c], mu, rs, sc, sh);
    const float m1 = (float)(sums[2 * c] / count), m2 = (float)(sums[2 * c + 1] / count);
    const size_t base = (size_t)pl * hw4;
    const f32x4* d4 = reinterpret_cast<const f32x4*>(dout) + base;
    const f32x4* r4 = reinterpret_cast<const f32x4*>(raw) + base;
    const f32x4* o4 = reinterpret_cast<const f32x4*>(out) + base;
    f32x4* w4 = reinterpret_cast<f32x4*>(draw) + base;
    f32x4* z4 = reinterpret_cast<f32x4*>(dz_out) + base;
    // the forward's mask bits (byte [n][c / 8][pixel], bit c % 8): four pixels' bytes in one dword
    const uint32_t* b4 = bits != nullptr ? reinterpret_cast<const uint32_t*>(bits + ((size_t)(pl / C) * (C >> 3) + (c >> 3)) * HW) : nullptr;
    for (int i = lane; i < hw4; i += 64) {
      f32x4 dz = d4[i];
      const f32x4 rw = r4[i];
      if (relu_mask == 1 && b4 != nullptr) {
        const uint32_t m = b4[i] >> (c & 7);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!((m >> (8 * k)) & 1u)) dz[k] = 0.f;
      } else if (relu_mask == 1) {
        const f32x4 o = o4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!(o[k] > 0.f)) dz[k] = 0.f;
      }
      if (relu_mask == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!(bn_eval(rw[k], sc, sh) > 0.f)) dz[k] = 0.f;
      }
      if (dz_out != nullptr) z4[i] = dz;
      f32x4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xh = (rw[k] - mu) * rs;
        v[k] = ga * rs * (dz[k] - m1 - xh * m2);
        const uint32_t vb = f2u(v[k]) & 0x7FFFFFFFu;          // (adx_common.h: never bit_cast a vector element in place)
        b = vb > b ? vb : b;
      }
      w4[i] = v;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
    b = o > b ? o : b;
  }
  if (lane == 0) red[wave] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
    amax[blockIdx.x] = m01 > m23 ? m01 : m23;
  }
}

// ---- the BatchNorm-backward apply pass by 8-channel groups, with draw written as a CELL tensor ----
// A gradient's range is only known once it has been written (the `amax` partials of the plane pass above), and a cell tensor needs
// its power-of-two scale WHEN it is written.  What is known beforehand is a bound: with D = max |dz| (from the pass that computed
// the sums: channel_sums_kernel<1> / the data-gradient epilogue) and |xhat| <= sqrt(count - 1) (Samuelson),
//     |draw_c| <= |gamma_c rstd_c| (D + |m1_c| + sqrt(count - 1) |m2_c|).
// The scale moves the largest such bound into [2^14, 2^15): whatever the bound overshoots costs RANGE below (values smaller than
// 2^-29 of the bound lose bits), never precision of the values that matter -- fp16 hi + lo keeps 22 bits across 29 octaves.
// bn_bwd_consts_kernel: one workgroup per record: per-channel constants [C][8] = {gamma rstd, m1, m2, mean, rstd, mask scale, mask
// shift, 0}, the affine gradients (the finished sums), and {xs, 1 / xs}.
__global__ void __launch_bounds__(256) bn_bwd_consts_kernel(const double* __restrict__ sums, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const uint32_t* __restrict__ dmax,
                                                             double count, int C, float* __restrict__ consts,
                                                             float* __restrict__ xscale, float* __restrict__ dgamma,
                                                             float* __restrict__ dbeta) {
  // D: the largest of the per-channel maxima (the data-gradient epilogue's path leaves the tensor's maximum in slot 0)
  __shared__ float red[4];
  float dm = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) dm = __builtin_fmaxf(dm, u2f(dmax[c]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) dm = __builtin_fmaxf(dm, __shfl_xor(dm, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dm;
  __syncthreads();
  const float D = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
  const float X = (float)sqrt(count > 1.0 ? count - 1.0 : 1.0);
  __syncthreads();
  float bound = 0.f;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float mu = mean[c], rs = rstd[c], ga = gamma[c];
    float sc, sh;
    bn_affine(ga, beta[c], mu, rs, sc, sh);
    const float m1 = (float)(sums[2 * c] / count), m2 = (float)(sums[2 * c + 1] / count);
    float* k = consts + (size_t)c * 8;
    k[0] = ga * rs; k[1] = m1; k[2] = m2; k[3] = mu; k[4] = rs; k[5] = sc; k[6] = sh; k[7] = 0.f;
    if (dgamma != nullptr) { dbeta[c] = (float)sums[2 * c]; dgamma[c] = (float)sums[2 * c + 1]; }
    bound = __builtin_fmaxf(bound, __builtin_fabsf(ga * rs) * (D + __builtin_fabsf(m1) + X * __builtin_fabsf(m2)));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) bound = __builtin_fmaxf(bound, __shfl_xor(bound, off, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = bound;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t b = f2u(__builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3])));
    const int e = (int)((b >> 23) & 0xFF);
    float xs = 1.f, xs_inv = 1.f;
    if (e != 0 && e != 255) {
      int sh = 127 + 14 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      xs = u2f((uint32_t)(127 + sh) << 23);
      xs_inv = u2f((uint32_t)(127 - sh) << 23);
    }
    xscale[0] = xs;
    xscale[1] = xs_inv;
  }
}

// MASK: 0 none, 1 the forward's bits, 2 re-derived from the conv output.  KEEP: also leave dz (fp32 NCHW: the identity path's
// gradient and the residual of the next data gradient's epilogue).
template <int MASK, bool KEEP>
__global__ void __launch_bounds__(256) bn_bwd_apply_groups_kernel(const float* __restrict__ dout, const float* __restrict__ raw,
                                                                   const uint8_t* __restrict__ bits, const float* __restrict__ consts,
                                                                   const float* __restrict__ xscale, u32x4* __restrict__ draw,
                                                                   float* __restrict__ dz_out, int C, int HW, int groups) {
  const float xs = xscale[0];
  const int per = (HW + 255) / 256;
  for (int w = blockIdx.x; w < groups * per; w += gridDim.x) {
    const int g = w / per, pix = (w - g * per) * 256 + threadIdx.x;
    if (pix >= HW) continue;
    const float* k = consts + (size_t)(g % (C >> 3)) * 64;        // this group's eight channels
    const size_t p0 = (size_t)g * 8 * HW + pix;
    float dz[8], rw[8], v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { dz[j] = dout[p0 + (size_t)j * HW]; rw[j] = raw[p0 + (size_t)j * HW]; }
    uint32_t m = 0xFFu;
    if constexpr (MASK == 1) m = bits[(size_t)g * HW + pix];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* kj = k + 8 * j;
      if (MASK == 1 && !((m >> j) & 1u)) dz[j] = 0.f;
      if (MASK == 2 && !(bn_eval(rw[j], kj[5], kj[6]) > 0.f)) dz[j] = 0.f;
      const float xh = (rw[j] - kj[3]) * kj[4];
      v[j] = kj[0] * (dz[j] - kj[1] - xh * kj[2]);
    }
    if constexpr (KEEP) {
#pragma unroll
      for (int j = 0; j < 8; ++j) dz_out[p0 + (size_t)j * HW] = dz[j];
    }
    u32x4 hi, lo;
    split8(v, xs, hi, lo);
    u32x4* o = draw + (size_t)g * 2 * HW + pix;
    o[0] = hi;
    o[HW] = lo;
  }
}

// dst[n][c][2y][2x] (+)= src[n][c][y][x]; everything else of dst untouched (caller zeroes when accumulate == 0)
__global__ void __launch_bounds__(256) dilate2_kernel(const float* __restrict__ src, float* __restrict__ dst, int planes,
                                                       int h, int w, int H, int W, int accumulate) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)planes * h * w) return;
  const int x = i % w, y = (i / w) % h;
  const size_t p = i / ((size_t)w * h);
  float* d = dst + (p * H + 2 * y) * W + 2 * x;
  *d = accumulate ? *d + src[i] : src[i];
}

// MaxPool2d(3, 2, 1) backward without atomics.  The forward left one byte per pooled element: which of its window's 9 taps is
// the first maximum (torch's tie rule).  maxpool_bwd_gather4: an input pixel lies in at most 2 x 2 windows; it receives dy of
// those whose code points back at it.

// Training forward of the stem's tail in ONE pass over the conv output: BatchNorm apply + ReLU + MaxPool2d(3, 2, 1) + the
// first-maximum tap of every window (one byte, for the backward).  The 944 MB post-BN stem map (B = 64,
// 3x256x900) is then never written nor re-read: bn_apply (read + write), maxpool (read) and a separate arg-max pass of the
// backward (read) collapse into one read.  Same values as the separate passes: v = relu(fma(raw, scale, shift)) as bn_apply forms it,
// the maximum with torch's NaN rule as maxpool_kernel, the code = the first tap that attains the maximum (torch's tie rule).
constexpr int kPoolRowsPerWave = 4;       // pooled rows a wave produces one after the other (fewer, longer workgroups)
__device__ __forceinline__ void bn_relu_pool_code_row(const float* __restrict__ raw, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ pooled,
                                                      uint8_t* __restrict__ code, int C, int H, int W, int OH, int OW, int pl, int oy,
                                                      int lane) {
  const float sc = scale[pl % C], sh = shift[pl % C];
  const float* src = raw + (size_t)pl * H * W;
  float* dst = pooled + ((size_t)pl * OH + oy) * OW;
  uint8_t* cdst = code + ((size_t)pl * OH + oy) * OW;
  if ((W & 1) == 0 && ((size_t)raw & 7) == 0) {
    // even widths (every map of the 256x900 geometry): a lane loads the two columns 2 ox, 2 ox + 1 of its window as one
    // 8-byte word and takes column 2 ox - 1 from its left neighbour (only lane 0 of a 64-column chunk loads it itself), like
    // maxpool_kernel: every element of a row is read once per window row instead of 1.5 times in 4-byte pieces
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    auto act = [&](float r) { const float a = bn_eval(r, sc, sh); return a > 0.f ? a : 0.f; };
    for (int ox0 = lane; ox0 - lane < OW; ox0 += 256) {     // wave-uniform trip count: the shuffles need every lane
      float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, pm[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
      int arg[4] = {-1, -1, -1, -1};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;                 // wave-uniform
        const float* row = src + (size_t)iy * W;
        f32x2 v[4];
        float left[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ox = ox0 + 64 * q;
          const bool in = 2 * ox + 1 < W;                // W even: both columns exist or neither
          v[q] = in ? *reinterpret_cast<const f32x2*>(row + 2 * ox) : f32x2{-INFINITY, -INFINITY};
          if (in) { v[q][0] = act(v[q][0]); v[q][1] = act(v[q][1]); }
          left[q] = (lane == 0 && ox > 0 && 2 * ox - 1 < W) ? act(row[2 * ox - 1]) : -INFINITY;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ox = ox0 + 64 * q;
          const float nb = __shfl_up(v[q][1], 1, 64);
          const float t3[3] = {lane == 0 ? left[q] : nb, v[q][0], v[q][1]};
          const bool ok[3] = {ox > 0 && 2 * ox - 1 < W, 2 * ox < W, 2 * ox + 1 < W};
#pragma unroll
          for (int dx = 0; dx < 3; ++dx) {
            const float x = t3[dx];
            if (ok[dx] && (x > m[q] || arg[q] < 0)) { m[q] = x; arg[q] = dy * 3 + dx; }
            if (ok[dx]) pm[q] = (x > pm[q] || x != x) ? x : pm[q];
          }
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ox0 + 64 * q < OW) {
          dst[ox0 + 64 * q] = pm[q];
          cdst[ox0 + 64 * q] = (uint8_t)arg[q];
        }
    }
    return;
  }
  for (int ox0 = lane; ox0 < OW; ox0 += 256) {
    float v[4][9];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
      const int iy = oy * 2 - 1 + dy;
      const bool rok = iy >= 0 && iy < H;
      const float* row = src + (size_t)(rok ? iy : 0) * W;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          const int ix = (ox0 + 64 * q) * 2 - 1 + dx;
          float a = -INFINITY;
          if (rok && ix >= 0 && ix < W) {
            a = bn_eval(row[ix], sc, sh);
            a = a > 0.f ? a : 0.f;
          }
          v[q][dy * 3 + dx] = a;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float m = -INFINITY, pm = -INFINITY;
      int arg = -1;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy * 2 - 1 + t / 3, ix = (ox0 + 64 * q) * 2 - 1 + t % 3;
        const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (in && (v[q][t] > m || arg < 0)) { m = v[q][t]; arg = t; }
        pm = (v[q][t] > pm || v[q][t] != v[q][t]) ? v[q][t] : pm;
      }
      if (ox0 + 64 * q < OW) {
        dst[ox0 + 64 * q] = pm;
        cdst[ox0 + 64 * q] = (uint8_t)arg;
      }
    }
  }
}

__global__ void __launch_bounds__(256) bn_relu_pool_code_kernel(const float* __restrict__ raw, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, float* __restrict__ pooled,
                                                                 uint8_t* __restrict__ code, int C, int H, int W, int OH, int OW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int rr = 0; rr < kPoolRowsPerWave; ++rr) {
    const int oy = (blockIdx.y * kPoolRowsPerWave + rr) * 4 + wave;
    if (oy < OH) bn_relu_pool_code_row(raw, scale, shift, pooled, code, C, H, W, OH, OW, blockIdx.x, oy, lane);
  }
}

// d(pool input) of the 4 consecutive pixels 4j .. 4j+3 of input row iy: their windows lie in output columns 2j, 2j+1, 2j+2,
// so 3 codes + 3 gradients per window row serve all four (cp / gp: the plane's codes and pooled-map gradient)
__device__ __forceinline__ void maxpool_bwd_gather4(const uint8_t* __restrict__ cp, const float* __restrict__ gp, int iy, int j,
                                                    int OH, int OW, float (&out)[4]) {
  // windows containing row iy: oy with 2 oy - 1 <= iy <= 2 oy + 1 (one for even rows, two for odd ones)
  const int oys[2] = {(iy + 1) >> 1, iy >> 1};
  const int noy = oys[0] == oys[1] ? 1 : 2;
  int cd[2][3];
  float g[2][3];
#pragma unroll
  for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int oy = oys[a2], ox = 2 * j + c;
      const bool ok = a2 < noy && oy < OH && ox < OW;
      cd[a2][c] = ok ? cp[(size_t)oy * OW + ox] : 255;
      g[a2][c] = ok ? gp[(size_t)oy * OW + ox] : 0.f;
    }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int ix = 4 * j + q;
    const int oxa = (ix + 1) >> 1, oxb = ix >> 1;
    float acc = 0.f;
#pragma unroll
    for (int a2 = 0; a2 < 2; ++a2) {
      const int ty = iy - (2 * oys[a2] - 1);
#pragma unroll
      for (int b2 = 0; b2 < 2; ++b2) {
        const int ox = b2 == 0 ? oxa : oxb;
        if (b2 == 1 && oxb == oxa) continue;
        const int c = ox - 2 * j;                        // 0..2 by construction
        const int tx = ix - (2 * ox - 1);
        acc += cd[a2][c] == ty * 3 + tx ? g[a2][c] : 0.f;
      }
    }
    out[q] = acc;
  }
}

// The stem's BatchNorm backward with the max-pool backward folded in: d(stem map) = the pooled map's gradient gathered
// through the arg-max codes is formed on the fly in both passes and never written (944 MB at B = 64, 3x256x900: one
// write and two reads less).  PASS 0 = channel_sums_kernel<1> with relu_mask 2 (one workgroup per plane, one fp64 atomic
// pair per plane); PASS 1 = bn_bwd_apply (one wave per input row): draw = gamma rstd (dz - m1 - xhat m2), and the affine
// parameters' gradients (the finished sums) by workgroup (0, 0).  Same arithmetic as the separate passes, except PASS 0's
// shortcut through the pooled tensors (below), which is gated to channels where it is accurate.
constexpr int kStemBwdRows = 4;
template <int PASS>
__global__ void __launch_bounds__(256) stem_pool_bn_bwd_kernel(const uint8_t* __restrict__ code, const float* __restrict__ dpool,
                                                                const float* __restrict__ raw, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, double* __restrict__ sums,
                                                                float* __restrict__ draw, int C, int H, int W, int OH, int OW,
                                                                double count, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                uint32_t* __restrict__ amax = nullptr, int n_amax = 0,
                                                                const float* __restrict__ pooled = nullptr) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t mx = 0;        // PASS 1: max |draw| of this workgroup (bit pattern), for the split-fp16 weight gradient's range
  const int pl = blockIdx.x, c = pl % C;
  // (the shortcut recovers xhat as (pooled - beta) / gamma, whose rounding error is eps |beta| / |gamma|: only where |gamma| is
  // at least 2^-6 of |beta| -- eps * 64 on xhat -- otherwise the channel takes the exact gather pass below)
  if (PASS == 0 && pooled != nullptr && fabsf(gamma[c]) >= 1e-12f && fabsf(gamma[c]) * 64.f >= fabsf(beta[c])) {
    // The sums from the POOLED tensors alone (472 MB instead of the 944 MB conv output + codes + gathers): d(stem map) is the
    // pooled gradient scattered to each window's arg-max position, so sum dz = sum over the windows whose maximum passed the ReLU
    // of their gradient, and sum dz xhat = the same sum weighted with xhat AT the arg-max -- which the pooled value itself gives
    // back: pooled = gamma xhat + beta there (> 0).  (gamma ~ 0 would lose xhat: such a channel takes the pass below.)
    const float ga = gamma[c], be = beta[c], inv = 1.f / ga;
    const size_t n = (size_t)OH * OW;
    const float* gp = dpool + (size_t)pl * n;
    const float* pp = pooled + (size_t)pl * n;
    double s0 = 0.0, s1 = 0.0;
    if ((n & 3) == 0 && (((size_t)gp | (size_t)pp) & 15) == 0) {
      for (size_t i = threadIdx.x; 4 * i < n; i += 256) {
        const f32x4 g4 = *reinterpret_cast<const f32x4*>(gp + 4 * i), a4 = *reinterpret_cast<const f32x4*>(pp + 4 * i);
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (a4[q] > 0.f) { s0 += g4[q]; s1 += (double)g4[q] * (double)((a4[q] - be) * inv); }
      }
    } else {
      for (size_t i = threadIdx.x; i < n; i += 256)
        if (pp[i] > 0.f) { s0 += gp[i]; s1 += (double)gp[i] * (double)((pp[i] - be) * inv); }
    }
    __shared__ double redp[8];
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) { redp[wave * 2] = s0; redp[wave * 2 + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(sums + 2 * c, (redp[0] + redp[2]) + (redp[4] + redp[6]));
      atomicAdd(sums + 2 * c + 1, (redp[1] + redp[3]) + (redp[5] + redp[7]));
    }
    return;
  }
  if (PASS == 1 && blockIdx.x == 0 && blockIdx.y == 0 && dgamma != nullptr)
    for (int k = threadIdx.x; k < C; k += 256) { dbeta[k] = (float)sums[2 * k]; dgamma[k] = (float)sums[2 * k + 1]; }
  const uint8_t* cp = code + (size_t)pl * OH * OW;
  const float* gp = dpool + (size_t)pl * OH * OW;
  const float mu = mean[c], rs = rstd[c], ga = gamma[c];
  float sc, sh;
  bn_affine(ga, beta[c], mu, rs, sc, sh);
  float m1 = 0.f, m2 = 0.f;
  if (PASS == 1) { m1 = (float)(sums[2 * c] / count); m2 = (float)(sums[2 * c + 1] / count); }
  double s0 = 0.0, s1 = 0.0;
  // PASS 1: a wave applies kStemBwdRows input rows one after the other (rows 4 apart: the workgroup's waves stay on neighbouring rows)
  const int row0 = PASS == 0 ? wave : blockIdx.y * 4 * kStemBwdRows + wave;
  const int row_end = PASS == 0 ? H : min(H, (int)(blockIdx.y + 1) * 4 * kStemBwdRows);
  for (int iy = row0; iy < row_end; iy += 4) {
    const float* rrow = raw + ((size_t)pl * H + iy) * W;
    float* drow = PASS == 1 ? draw + ((size_t)pl * H + iy) * W : nullptr;
    for (int j = lane; 4 * j < W; j += 64) {
      float dz[4];
      maxpool_bwd_gather4(cp, gp, iy, j, OH, OW, dz);
      // a lane's 4 pixels are 16 contiguous bytes, 4-byte aligned (rows of 450 floats start on 8-byte boundaries only)
      typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
      const bool whole = 4 * j + 3 < W;
      float rw4[4] = {0.f, 0.f, 0.f, 0.f};
      if (whole) {
        const f32x4u t = *reinterpret_cast<const f32x4u*>(rrow + 4 * j);
        rw4[0] = t[0]; rw4[1] = t[1]; rw4[2] = t[2]; rw4[3] = t[3];
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) if (4 * j + q < W) rw4[q] = rrow[4 * j + q];
      }
      float o4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float rw = rw4[q];
        float v = (4 * j + q < W) ? dz[q] : 0.f;
        if (!(bn_eval(rw, sc, sh) > 0.f)) v = 0.f;          // the stem's ReLU sits straight behind its BatchNorm
        const float xh = (rw - mu) * rs;
        if (PASS == 0) {
          s0 += v;
          s1 += (double)v * (double)xh;
        }
        o4[q] = ga * rs * (v - m1 - xh * m2);
        if (PASS == 1 && 4 * j + q < W) {
          const uint32_t ob = f2u(o4[q]) & 0x7FFFFFFFu;
          mx = ob > mx ? ob : mx;
        }
      }
      if (PASS == 1) {
        if (whole) {
          *reinterpret_cast<f32x4u*>(drow + 4 * j) = f32x4u{o4[0], o4[1], o4[2], o4[3]};
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) if (4 * j + q < W) drow[4 * j + q] = o4[q];
        }
      }
    }
  }
  if (PASS == 1 && amax != nullptr) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)mx, off, 64);
      mx = o > mx ? o : mx;
    }
    // one slot per (plane, row band) modulo the table: the slots were cleared by the caller
    if (lane == 0 && mx != 0) atomicMax(amax + (blockIdx.x * gridDim.y + blockIdx.y) % n_amax, mx);
  }
  if (PASS == 0) {
    __shared__ double red[8];
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    if (lane == 0) { red[wave * 2] = s0; red[wave * 2 + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
      atomicAdd(sums + 2 * c, (red[0] + red[2]) + (red[4] + red[6]));
      atomicAdd(sums + 2 * c + 1, (red[1] + red[3]) + (red[5] + red[7]));
    }
  }
}

// avgpool + fc backward.  Grid = (image, channel slice): each workgroup recomputes the pooled values of its 64 channels
// (one wave per 16 of them), forms their share of d(pooled) and of the fc weight gradient, and broadcasts d(pooled) over
// the map with row-contiguous stores.
constexpr int kPoolSlice = 64;
__global__ void __launch_bounds__(256) avgpool_fc_bwd_kernel(const float* __restrict__ x, const float* __restrict__ fw,
                                                              const float* __restrict__ dfeat, float* __restrict__ dx,
                                                              float* __restrict__ dfw, float* __restrict__ dfb, int C,
                                                              int HW, int out_dim) {
  __shared__ float pooled[kPoolSlice];
  __shared__ float dpool[kPoolSlice];
  const int nslice = C / kPoolSlice;
  const int n = blockIdx.x / nslice, c0 = (blockIdx.x % nslice) * kPoolSlice;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = x + ((size_t)n * C + c0) * HW;
  const float* df = dfeat + (size_t)n * out_dim;
  const float inv = 1.0f / (float)HW;
  for (int c = wave; c < kPoolSlice; c += 4) {
    float s = 0.f;
    for (int i = lane; i < HW; i += 64) s += src[(size_t)c * HW + i];
    s = wave_sum(s);
    if (lane == 0) pooled[c] = s * inv;
  }
  if (tid < kPoolSlice) {
    float s = 0.f;
    for (int j = 0; j < out_dim; ++j) s += df[j] * fw[(size_t)j * C + c0 + tid];
    dpool[tid] = s * inv;
  }
  __syncthreads();
  float* dst = dx + ((size_t)n * C + c0) * HW;
  for (int c = wave; c < kPoolSlice; c += 4) {
    const float v = dpool[c];
    for (int i = lane; i < HW; i += 64) dst[(size_t)c * HW + i] = v;
  }
  for (int i = tid; i < out_dim * kPoolSlice; i += 256) {
    const int j = i / kPoolSlice, c = i - j * kPoolSlice;
    atomicAdd(dfw + (size_t)j * C + c0 + c, df[j] * pooled[c]);
  }
  if (c0 == 0)
    for (int j = tid; j < out_dim; j += 256) atomicAdd(dfb + j, df[j]);
}

// ---------------------------------------------------------------------------------------------
// conv2d weight gradient on v_mfma_f32_32x32x2_f32.
//   A (32 x 2): draw[co][pixel], B (2 x 32): x[(ci, tap) combination][pixel shifted by the tap],
//   K = output pixels.  Lane j of column group g stands for one (input channel, tap) pair, so the same
//   kernel serves 3x3 / 1x1 convs (group = tap, lane = channel) and the 3-channel 7x7 stem
//   (147 pairs spread over 5 groups).  One workgroup owns a 32-cout x NG*32-pair weight tile, walks a
//   range of (image, 4x32 pixel tile) units with the register double buffer of the forward conv, sums
//   its 4 waves through LDS and adds the tile to dW with float atomics.
struct WgradArgs2 {
  const float* x;        // [N][Cin][H][W]
  const float* dy;       // [N][Cout][OH][OW]
  float* dw;             // [Cout][Cin][K][K], zeroed by the caller
  int N, Cin, H, W, Cout, OH, OW, pad;
  int tiles_x, tiles_y, units, units_per_wg, n_ci_tiles, n_co_tiles;
  int ci_per_tile, pairs;   // channels staged per tile (32, or 3 for the stem), valid (ci, tap) pairs per tile
};

template <int STRIDE, int K, int NG, int CIT>
__global__ void __launch_bounds__(256) conv2d_wgrad_kernel(const WgradArgs2 a) {
  constexpr int PH = (kTileH - 1) * STRIDE + K, PW = (kTileW - 1) * STRIDE + K;
  constexpr int PLANE = PH * PW + ((PH * PW) % 2 == 0 ? 1 : 0);   // odd plane pitch: lanes = channels hit distinct banks
  constexpr int NP = CIT * PH * PW;
  constexpr int PITEMS = (NP + 255) / 256;
  constexpr int DP = kTileH * kTileW + 1;                          // odd pitch of the draw tile
  constexpr int ND = 32 * kTileH * kTileW;
  constexpr int DITEMS = ND / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                 // [CIT][PLANE]
  float* dyt = smem + CIT * PLANE;     // [32][DP]
  constexpr int kZero = CIT * PLANE + 32 * DP;   // one always-zero word behind the staged tiles
  const float* pz = smem + kZero;      // B operands are addressed relative to it
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int ci_t = bid % a.n_ci_tiles; bid /= a.n_ci_tiles;
  const int co_t = bid % a.n_co_tiles; bid /= a.n_co_tiles;
  const int u0 = bid * a.units_per_wg, u1 = min(u0 + a.units_per_wg, a.units);
  const int ci0 = ci_t * a.ci_per_tile, co0 = co_t * 32;
  const int l31 = lane & 31, khalf = lane >> 5;
  const size_t hw = (size_t)a.H * a.W, ohw = (size_t)a.OH * a.OW;

  // (ci, tap) pair of this lane in every column group, as an offset into the patch
  int poff[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int q = g * 32 + l31;
    int cl, tap;
    if (K != 7) { cl = l31; tap = g; } else { cl = q / (K * K); tap = q - cl * (K * K); }
    const bool ok = q < a.pairs && ci0 + cl < a.Cin && cl < CIT;
    poff[g] = ok ? cl * PLANE + (tap / K) * PW + (tap % K) - kZero : 0;   // invalid pairs read the zero word
  }
  f32x16 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  float pv[PITEMS], dv[DITEMS];
  auto load_unit = [&](int u) {
    const int tx = u % a.tiles_x, ty = (u / a.tiles_x) % a.tiles_y, n = u / (a.tiles_x * a.tiles_y);
    const int oy0 = ty * kTileH, ox0 = tx * kTileW;
    const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
    const float* xin = a.x + ((size_t)n * a.Cin + ci0) * hw;
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
      const int e = tid + 256 * k;
      const int c = e / (PH * PW), rem = e - c * (PH * PW);
      const int py = rem / PW, px = rem - py * PW;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool ok = e < NP && ci0 + c < a.Cin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      pv[k] = ok ? xin[c * hw + (size_t)iy * a.W + ix] : 0.f;
    }
    const float* dyin = a.dy + ((size_t)n * a.Cout + co0) * ohw;
#pragma unroll
    for (int k = 0; k < DITEMS; ++k) {
      const int e = tid + 256 * k;
      const int c = e / (kTileH * kTileW), rem = e - c * (kTileH * kTileW);
      const int oy = oy0 + rem / kTileW, ox = ox0 + rem % kTileW;
      const bool ok = co0 + c < a.Cout && oy < a.OH && ox < a.OW;
      dv[k] = ok ? dyin[c * ohw + (size_t)oy * a.OW + ox] : 0.f;
    }
  };
  auto store_unit = [&]() {
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
      const int e = tid + 256 * k;
      if (e < NP) {
        const int c = e / (PH * PW);
        patch[c * PLANE + (e - c * (PH * PW))] = pv[k];
      }
    }
#pragma unroll
    for (int k = 0; k < DITEMS; ++k) {
      const int e = tid + 256 * k;
      const int c = e / (kTileH * kTileW);
      dyt[c * DP + (e - c * (kTileH * kTileW))] = dv[k];
    }
  };

  if (tid == 0) smem[kZero] = 0.f;
  if (u0 < u1) load_unit(u0);
  for (int u = u0; u < u1; ++u) {
    if (u > u0) __syncthreads();
    store_unit();
    __syncthreads();
    if (u + 1 < u1) load_unit(u + 1);
    // this wave: output row `wave` of the tile, 16 K-steps of 2 pixels
    const float* ap = dyt + l31 * DP + wave * kTileW + khalf;
    const int prow = (wave * STRIDE) * PW + khalf * STRIDE;
#pragma unroll 4
    for (int ks = 0; ks < kTileW / 2; ++ks) {
      const float av = ap[2 * ks];
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float bv = pz[poff[g] == 0 ? 0 : poff[g] + prow + 2 * ks * STRIDE];
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[g], 0, 0, 0);
      }
    }
  }
  // sum the 4 waves into one LDS tile (one wave at a time), then one atomic per weight
  __syncthreads();
  float* red = smem;   // [NG][32 x 32]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * khalf;   // cout within the tile
          float* dst = red + (g * 32 + row) * 32 + l31;
          *dst = w == 0 ? acc[g][r] : *dst + acc[g][r];
        }
    }
    __syncthreads();
  }
  // walk the tile in the order of dW ([co][ci][tap]) so that consecutive lanes add to consecutive addresses
  constexpr int KK = K * K;
  const int rowf = a.ci_per_tile * KK;   // contiguous floats per output channel in this tile
  for (int e = tid; e < 32 * rowf; e += 256) {
    const int row = e / rowf, r = e - row * rowf;
    const int cl = r / KK, tap = r - cl * KK;
    int g, j;
    if (K != 7) { g = tap; j = cl; } else { const int q = cl * KK + tap; g = q >> 5; j = q & 31; }
    const int co = co0 + row, ci = ci0 + cl;
    if (co < a.Cout && ci < a.Cin) atomicAdd(a.dw + ((size_t)co * a.Cin + ci) * KK + tap, red[(g * 32 + row) * 32 + j]);
  }
}

// dw[i] += dw9[9 i + 4]: the centre tap of a 3x3 gradient is the gradient of the 1x1 conv at the same stride
__global__ void centre_tap_add_kernel(float* __restrict__ dw, const float* __restrict__ dw9, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) dw[i] += dw9[(size_t)i * 9 + 4];
}

// scratch9 (optional, Cout * Cin * 9 floats): lets the 1x1 stride-2 downsample convs ride on the split-fp16 3x3
// stride-2 kernel -- x[2 oy][2 ox] is exactly that kernel's centre tap -- which is twice as fast as the exact-fp32 1x1
// kernel even though eight of its nine taps are thrown away
int conv2d_wgrad(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int k, int stride,
                 int pad, hipStream_t s, const uint32_t* dy_amax = nullptr, int dy_amax_n = 0, bool zero = true,
                 float* scratch9 = nullptr, bool x_cells = false, bool dy_cells = false) {
  ADX_REQUIRE(x && dy && dw, "conv2d_wgrad: null tensor");
  ADX_REQUIRE(!dy_cells || (conv2d_wgrad_hs_eligible(Cin, Cout, k, stride, pad) && k == 3 && stride == 1),
              "conv2d_wgrad: a cell-layout gradient belongs to the split-fp16 3x3 stride-1 weight gradient");
  ADX_REQUIRE(!x_cells || (conv2d_wgrad_hs_eligible(Cin, Cout, k, stride, pad) && k == 3) ||
                  (scratch9 != nullptr && k == 1 && stride == 2 && pad == 0 && conv2d_wgrad_hs_eligible(Cin, Cout, 3, 2, 1)),
              "conv2d_wgrad: a cell-layout input belongs to the split-fp16 3x3 weight gradient");
  // zero = false: the caller has already cleared dw (the training executor clears every weight gradient in one batch)
  if (zero) ADX_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cout * Cin * k * k, s));
  if (scratch9 != nullptr && k == 1 && stride == 2 && pad == 0 && conv2d_wgrad_hs_eligible(Cin, Cout, 3, 2, 1)) {
    ADX_CHECK_HIP(hipMemsetAsync(scratch9, 0, sizeof(float) * (size_t)Cout * Cin * 9, s));
    const int rc = conv2d_wgrad_hs(x, dy, scratch9, N, Cin, H, W, Cout, 2, dy_amax, dy_amax_n, s, x_cells);
    if (rc != ADX_OK) return rc;
    centre_tap_add_kernel<<<dim3(ceil_div(Cout * Cin, 256)), dim3(256), 0, s>>>(dw, scratch9, Cout * Cin);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  if (conv2d_wgrad_hs_eligible(Cin, Cout, k, stride, pad))
    return conv2d_wgrad_hs(x, dy, dw, N, Cin, H, W, Cout, stride, dy_amax, dy_amax_n, s, x_cells, dy_cells);
  // the stem: split-fp16 kernel when the gradient's range is known (it is far below fp16's), the exact-fp32 kernel otherwise
  if (dy_amax != nullptr && conv2d_wgrad_stem_hs_eligible(Cin, Cout, k, stride, pad))
    return conv2d_wgrad_stem_hs(x, dy, dw, N, H, W, dy_amax, dy_amax_n, s);
  WgradArgs2 a;
  a.x = x; a.dy = dy; a.dw = dw;
  a.N = N; a.Cin = Cin; a.H = H; a.W = W; a.Cout = Cout; a.pad = pad;
  a.OH = conv_out_dim(H, k, stride, pad); a.OW = conv_out_dim(W, k, stride, pad);
  a.tiles_x = ceil_div(a.OW, kTileW); a.tiles_y = ceil_div(a.OH, kTileH);
  a.units = N * a.tiles_x * a.tiles_y;
  const bool stem = k == 7;
  const bool half_tile = stride == 2 && k == 3;   // 9 x 65 patch rows per channel: stage 16 channels to stay in registers
  a.ci_per_tile = stem ? 3 : (half_tile ? 16 : 32);
  ADX_REQUIRE(stem ? Cin == 3 : Cin % 32 == 0, "conv2d_wgrad: cin %d unsupported", Cin);
  ADX_REQUIRE(Cout % 32 == 0, "conv2d_wgrad: cout %d must be a multiple of 32", Cout);
  a.n_ci_tiles = stem ? 1 : Cin / a.ci_per_tile;
  a.n_co_tiles = Cout / 32;
  a.pairs = stem ? 3 * 49 : 32 * k * k;
  const int tiles = a.n_ci_tiles * a.n_co_tiles;
  int splits = ceil_div(1024, tiles);
  if (splits > a.units) splits = a.units;
  a.units_per_wg = ceil_div(a.units, splits);
  splits = ceil_div(a.units, a.units_per_wg);
  const dim3 grid((unsigned)(tiles * splits)), blk(256);
  auto lds_bytes = [&](int cit, int ng) {
    const int ph = (kTileH - 1) * stride + k, pw = (kTileW - 1) * stride + k;
    const int plane = ph * pw + ((ph * pw) % 2 == 0 ? 1 : 0);
    const size_t stage = (size_t)cit * plane + 32 * (kTileH * kTileW + 1) + 4;
    const size_t red = (size_t)ng * 1024;
    return sizeof(float) * std::max(stage, red);
  };
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    const void* fns[4] = {reinterpret_cast<const void*>(&conv2d_wgrad_kernel<1, 3, 9, 32>),
                          reinterpret_cast<const void*>(&conv2d_wgrad_kernel<2, 3, 9, 16>),
                          reinterpret_cast<const void*>(&conv2d_wgrad_kernel<2, 1, 1, 32>),
                          reinterpret_cast<const void*>(&conv2d_wgrad_kernel<2, 7, 5, 3>)};
    for (const void* f : fns) ADX_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    once.commit();
  }
  if (stride == 1 && k == 3) conv2d_wgrad_kernel<1, 3, 9, 32><<<grid, blk, lds_bytes(32, 9), s>>>(a);
  else if (stride == 2 && k == 3) conv2d_wgrad_kernel<2, 3, 9, 16><<<grid, blk, lds_bytes(16, 9), s>>>(a);
  else if (stride == 2 && k == 1) conv2d_wgrad_kernel<2, 1, 1, 32><<<grid, blk, lds_bytes(32, 1), s>>>(a);
  else if (stride == 2 && k == 7) conv2d_wgrad_kernel<2, 7, 5, 3><<<grid, blk, lds_bytes(3, 5), s>>>(a);
  else {
    set_error("conv2d_wgrad: no kernel for k=%d stride=%d", k, stride);
    return ADX_ERR_INVALID;
  }
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// max |v| over a tensor as gridDim.x partial maxima (bit patterns), for the split-fp16 kernels' dynamic range
__global__ void __launch_bounds__(256) amax_partials_kernel(const float* __restrict__ v, size_t total, uint32_t* __restrict__ out) {
  __shared__ uint32_t red[4];
  uint32_t b = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const uint32_t vb = __builtin_bit_cast(uint32_t, v[i]) & 0x7FFFFFFFu;
    b = vb > b ? vb : b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
    b = o > b ? o : b;
  }
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = b;
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t m01 = red[0] > red[1] ? red[0] : red[1], m23 = red[2] > red[3] ? red[2] : red[3];
    out[blockIdx.x] = m01 > m23 ? m01 : m23;
  }
}

}  // namespace adx

extern "C" {

size_t adx_conv2d_wgrad_scratch_bytes(void) {
  return adx::kAmaxPartials * sizeof(uint32_t) + adx::conv2d_wgrad_partials_floats() * sizeof(float);     // [range partials | ADX_WGRAD_DETERMINISTIC's copies]
}

int adx_conv2d_wgrad_ex(const adx_conv2d_desc* d, const float* x, const float* dy, float* dw, int32_t n, int32_t h,
                        int32_t w, void* scratch, int32_t estimate_range, adx_stream stream) {
  using namespace adx;
  ADX_REQUIRE(d && x && dy && dw, "adx_conv2d_wgrad: null argument");
  ADX_REQUIRE(n >= 1 && h + 2 * d->pad >= d->k && w + 2 * d->pad >= d->k, "adx_conv2d_wgrad: input too small");
  ADX_REQUIRE(!estimate_range || scratch != nullptr, "adx_conv2d_wgrad: the range estimate needs the scratch buffer");
  hipStream_t s = (hipStream_t)stream;
  // the scratch holds [range partials | ADX_WGRAD_DETERMINISTIC's per-split copies]; the two uses are independent: the range
  // estimate (which also moves the stem onto the split-fp16 kernel) runs only when asked for
  uint32_t* amax = estimate_range ? (uint32_t*)scratch : nullptr;
  int n_amax = 0;
  if (amax != nullptr) {
    const size_t total = (size_t)n * d->cout * conv_out_dim(h, d->k, d->stride, d->pad) * conv_out_dim(w, d->k, d->stride, d->pad);
    n_amax = (int)std::min<size_t>(kAmaxPartials, (total + 255) / 256);
    amax_partials_kernel<<<dim3(n_amax), dim3(256), 0, s>>>(dy, total, amax);
    ADX_LAUNCH_CHECK();
  }
  struct PartsScope {
    ~PartsScope() { conv2d_wgrad_set_partials(nullptr, 0); }
  } parts_scope;
  if (scratch != nullptr && conv2d_wgrad_partials_floats() > 0)
    conv2d_wgrad_set_partials(reinterpret_cast<float*>((uint32_t*)scratch + kAmaxPartials), conv2d_wgrad_partials_floats());
  return conv2d_wgrad(x, dy, dw, n, d->cin, h, w, d->cout, d->k, d->stride, d->pad, s, amax, n_amax);
}

int adx_conv2d_wgrad(const adx_conv2d_desc* d, const float* x, const float* dy, float* dw, int32_t n, int32_t h,
                     int32_t w, void* scratch, adx_stream stream) {
  return adx_conv2d_wgrad_ex(d, x, dy, dw, n, h, w, scratch, scratch != nullptr ? 1 : 0, stream);
}

int adx_conv2d_wgrad_cells(const adx_conv2d_desc* d, const void* x_cells, const void* dy_cells, const float* dy_scale, float* dw,
                           int32_t n, int32_t h, int32_t w, void* scratch, adx_stream stream) {
  using namespace adx;
  ADX_REQUIRE(d && x_cells && dy_cells && dy_scale && dw, "adx_conv2d_wgrad_cells: null argument");
  ADX_REQUIRE(d->k == 3 && d->stride == 1 && d->pad == 1 && conv2d_wgrad_hs_eligible(d->cin, d->cout, 3, 1, 1),
              "adx_conv2d_wgrad_cells: 3x3 stride-1 pad-1 convolutions with cin, cout multiples of 64 (and no ADX_*_EXACT switch)");
  ADX_REQUIRE(n >= 1 && h >= 1 && w >= 1, "adx_conv2d_wgrad_cells: empty input");
  struct PartsScope {
    ~PartsScope() { conv2d_wgrad_set_partials(nullptr, 0); }
  } parts_scope;
  if (scratch != nullptr && conv2d_wgrad_partials_floats() > 0)
    conv2d_wgrad_set_partials(reinterpret_cast<float*>((uint32_t*)scratch + kAmaxPartials), conv2d_wgrad_partials_floats());
  return conv2d_wgrad(reinterpret_cast<const float*>(x_cells), reinterpret_cast<const float*>(dy_cells), dw, n, d->cin, h, w, d->cout, 3, 1, 1,
                      (hipStream_t)stream, reinterpret_cast<const uint32_t*>(dy_scale), -1, true, nullptr, true, true);
}

}  // extern "C"

namespace adx {

static size_t al64(size_t v) { return (v + 63) / 64 * 64; }

struct Bump2 {
  float* base; size_t off, cap; bool ok = true;
  float* take(size_t n) {
    const size_t o = off;
    off = al64(off + n);
    if (off > cap) { ok = false; return base; }
    return base + o;
  }
};

}  // namespace adx

// what the forward leaves behind for the backward, per conv launch
struct adx_resnet_tape {
  struct Rec {
    const adx::ConvSpec* L = nullptr;
    const float* x = nullptr;       // conv input
    float* raw = nullptr;           // conv output before BN
    float* out = nullptr;           // after BN [+identity] [ReLU]
    const float* identity = nullptr;
    float* mean = nullptr; float* rstd = nullptr;
    int H = 0, W = 0, OH = 0, OW = 0, relu = 0;
    bool x_cells = false;           // x is a cell tensor (bn_apply_groups_kernel), `out` of the record that produced it likewise
    bool out_cells = false;
    uint8_t* bits = nullptr;        // ReLU mask of `out`, one bit per element (ReLU after the residual add), or null: read `out`
  };
  std::vector<Rec> recs;
  int batch = 0, h = 0, w = 0;
  float* pool_in = nullptr; float* pool_out = nullptr; int ph = 0, pw = 0, poh = 0, pow_ = 0;
  uint8_t* pool_code = nullptr;   // first-maximum tap of every pooling window, written by the forward's fused stem tail
  float* final_map = nullptr; int fh = 0, fw_ = 0;
  float* stats_part = nullptr;    // kStatsPartFloats floats of the forward's workspace: conv-epilogue partial sums (both passes)
  size_t fwd_floats = 0;
};

using namespace adx;

extern "C" {

int adx_resnet_tape_create(adx_resnet_tape** out) {
  ADX_REQUIRE(out != nullptr, "adx_resnet_tape_create: null argument");
  *out = new adx_resnet_tape();
  return ADX_OK;
}
void adx_resnet_tape_destroy(adx_resnet_tape* t) { delete t; }

size_t adx_resnet_train_workspace_bytes(const adx_resnet* r, int32_t batch, int32_t h, int32_t w) {
  if (!r || batch < 1 || h < 32 || w < 32) return 0;
  size_t f = al64(r->convs.size() * 2 * 512 * 2) + 2 * al64(512) + al64(kStatsPartFloats);   // forward: per-conv sums, scale, shift, conv-epilogue partial sums
  size_t big = 0, wmax = 0, wall = 0;
  auto conv = [&](const ConvSpec& L, int H, int W, bool apply = true) {
    const int OH = conv_out_dim(H, L.k, L.stride, L.pad), OW = conv_out_dim(W, L.k, L.stride, L.pad);
    const size_t n = (size_t)batch * L.cout * OH * OW;
    f += (apply ? 2 : 1) * al64(n) + 2 * al64(L.cout) + al64(n / 32 + 1);      // conv output (+ post-BN map) + saved mean / rstd + mask bits
    big = std::max(big, n);
    big = std::max(big, (size_t)batch * L.cin * H * W);
    wmax = std::max(wmax, (size_t)L.k * L.k * L.cout * L.cin);
    if (L.k == 3 && L.stride == 1) wall += al64((size_t)L.k * L.k * L.cout * L.cin);
  };
  size_t ci = 0;
  conv(r->convs[ci++], h, w, false);        // the stem's post-BN map is never formed (fused BN + ReLU + pool pass)
  const int h1 = conv_out_dim(h, 7, 2, 3), w1 = conv_out_dim(w, 7, 2, 3);
  int H = conv_out_dim(h1, 3, 2, 1), W = conv_out_dim(w1, 3, 2, 1);
  f += al64((size_t)batch * 64 * H * W) + al64(((size_t)batch * 64 * H * W + 3) / 4);    // pooled map + its arg-max codes
  big = std::max(big, (size_t)batch * 64 * h1 * w1);
  for (size_t b = 0; b < r->block_has_ds.size(); ++b) {
    const ConvSpec& c1 = r->convs[ci++];
    const ConvSpec& c2 = r->convs[ci++];
    conv(c1, H, W);
    const int OH = conv_out_dim(H, 3, c1.stride, 1), OW = conv_out_dim(W, 3, c1.stride, 1);
    if (r->block_has_ds[b]) conv(r->convs[ci++], H, W);
    conv(c2, OH, OW);
    H = OH; W = OW;
  }
  f += al64(r->convs.size() * 512) + al64(512 * 8) + al64(2) + al64(r->convs.size() * 2 * 512 * 2) + al64(kAmaxPartials) + al64((size_t)512 * 256 * 9) + al64(conv2d_wgrad_partials_floats()) + 5 * al64(big) + 2 * al64(wmax) + wall;   // backward: sums, amax, 3x3 image of a 1x1 gradient, 5 gradient buffers, dgrad weights (one scratch pair + a slot per stride-1 3x3 conv)
  return (f + 1024) * sizeof(float);
}

// tensors / running buffers as in adx_resnet_pack (state_dict order without num_batches_tracked); the
// running_mean / running_var entries are UPDATED in place (momentum 0.1) when update_running != 0.
int adx_resnet_forward_train(adx_resnet* r, const float* const* T, int32_t n_tensors, void* packed, void* workspace,
                             size_t workspace_bytes, const float* img, int32_t batch, int32_t h, int32_t w,
                             float* feature, adx_resnet_tape* tape, int32_t update_running, adx_stream stream) {
  ADX_REQUIRE(r && T && packed && workspace && img && feature && tape, "adx_resnet_forward_train: null argument");
  ADX_REQUIRE(n_tensors == r->n_tensors, "adx_resnet_forward_train: expected %d tensors, got %d", r->n_tensors, n_tensors);
  ADX_REQUIRE(batch >= 1 && h >= 32 && w >= 32, "adx_resnet_forward_train: image too small");
  hipStream_t s = (hipStream_t)stream;
  float* base = (float*)packed;
  // weights change every step: re-lay them here (conv images only; BN is applied from batch statistics) -- the split-fp16
  // images of the 3x3 convs in ONE launch, the stem and the exact-fp32 1x1 images on their own
  {
    std::vector<HsPackJob> jobs;
    for (const ConvSpec& L : r->convs) {
      // (a downsample conv that rides on its block's stride-2 conv1 -- one fused launch below -- takes that kernel's image)
      if (conv2d_hs_pack_batchable(L, 0) || (L.fuse_with >= 0 && resnet_fuses_ds(r->convs[L.fuse_with], L))) {
        jobs.push_back(HsPackJob{T[L.t_w], base + L.o_w, L.cout, L.cin_pad, L.cin, L.k * L.k, 0});
      } else {
        int rc = conv2d_pack_spec(L, T[L.t_w], base + L.o_w, 0, s);
        if (rc != ADX_OK) return rc;
      }
    }
    if (!jobs.empty()) {
      int rc = conv2d_hs_pack_many(jobs.data(), (int)jobs.size(), s);
      if (rc != ADX_OK) return rc;
    }
  }
  Bump2 ws{(float*)workspace, 0, workspace_bytes / sizeof(float)};
  tape->recs.clear();
  tape->batch = batch; tape->h = h; tape->w = w;
  // one statistics slot per conv, all cleared by a single memset (they are accumulated atomically)
  const size_t n_convs = r->convs.size();
  double* sums_all = reinterpret_cast<double*>(ws.take(n_convs * 2 * 512 * 2));
  if (ws.ok) ADX_CHECK_HIP(hipMemsetAsync(sums_all, 0, sizeof(double) * n_convs * 2 * 512, s));
  float* scale = ws.take(512);
  float* shift = ws.take(512);
  float* stats_part = ws.take(kStatsPartFloats);
  tape->stats_part = stats_part;
  int rc = ADX_OK;
  // x_cells: x is a cell tensor; out_cells: leave the post-BN map as one (its only readers are the next conv and that conv's
  // weight gradient)
  auto conv_bn = [&](const ConvSpec& L, const float* x, int H, int W, const float* identity, int relu,
                     bool apply = true, float* raw_done = nullptr, bool x_cells = false, bool out_cells = false,
                     bool id_cells = false) -> float* {
    adx_resnet_tape::Rec rec;
    rec.L = &L; rec.x = x; rec.H = H; rec.W = W; rec.relu = relu; rec.identity = identity;
    rec.x_cells = x_cells; rec.out_cells = out_cells;
    rec.OH = conv_out_dim(H, L.k, L.stride, L.pad); rec.OW = conv_out_dim(W, L.k, L.stride, L.pad);
    const size_t n = (size_t)batch * L.cout * rec.OH * rec.OW;
    // apply == false (the stem): the post-BN map is never formed, so it gets no storage either (0.94 GB at B = 64)
    // raw_done: the conv output already sits there (the fused stride-2 launch of a downsample block)
    rec.raw = raw_done != nullptr ? raw_done : ws.take(n);
    rec.out = apply ? ws.take(n) : nullptr; rec.mean = ws.take(L.cout); rec.rstd = ws.take(L.cout);
    // a block's output: the ReLU mask as bits for the backward pass (ADX_TRAIN_CELLS=0: it reads `out`)
    const bool want_bits = apply && relu && identity != nullptr && L.cout % 8 == 0 && debug_switches().train_cells;
    if (rc == ADX_OK && ((out_cells && identity != nullptr && !want_bits) || (id_cells && !want_bits))) {
      set_error("adx_resnet_forward_train: a cell-layout block output needs the group form of the BatchNorm apply pass");
      rc = ADX_ERR_STATE;
    }
    if (want_bits) rec.bits = reinterpret_cast<uint8_t*>(ws.take((n / 8 + 3) / 4));
    if (!ws.ok || rc != ADX_OK) { tape->recs.push_back(rec); return rec.out; }
    int stats_p = 0;     // > 0: the conv's own epilogue left per-workgroup partial sums (the 3x3 stride-1 layers)
    if (raw_done == nullptr)
      rc = conv2d_launch_raw(L, x, base + L.o_w, nullptr, nullptr, nullptr, rec.raw, batch, H, W, 0, s, nullptr, 0, stats_part,
                             kStatsPartFloats, &stats_p, x_cells ? kFmtXCells : 0);
    if (rc != ADX_OK) { tape->recs.push_back(rec); return rec.out; }
    const int HW = rec.OH * rec.OW;
    double* sums = sums_all + (size_t)(&L - r->convs.data()) * 2 * 512;
    float* const run_m = update_running ? const_cast<float*>(T[L.t_m]) : nullptr;
    float* const run_v = update_running ? const_cast<float*>(T[L.t_v]) : nullptr;
    if (stats_p > 0) {       // partial sums from the conv epilogue: reduce and finish the channel in one launch
      stats_reduce_finalize_kernel<<<dim3(L.cout), dim3(256), 0, s>>>(stats_part, sums, stats_p, T[L.t_g], T[L.t_b], scale, shift,
                                                                      rec.mean, rec.rstd, run_m, run_v, (double)batch * HW);
    } else {
      channel_sums_kernel<0><<<dim3(batch * L.cout), dim3(256), 0, s>>>(rec.raw, nullptr, nullptr, nullptr, nullptr, sums,
                                                                        L.cout, HW, 0, nullptr, nullptr);
      bn_finalize_kernel<<<dim3(ceil_div(L.cout, 256)), dim3(256), 0, s>>>(sums, T[L.t_g], T[L.t_b], scale, shift, rec.mean, rec.rstd,
                                                                           run_m, run_v, L.cout, (double)batch * HW);
    }
    if (!apply) {
      // the caller consumes (raw, scale, shift) itself before the next conv_bn overwrites scale / shift (stream order)
    } else if (out_cells || rec.bits != nullptr) {
      const int groups = batch * (L.cout / 8), per = ceil_div(HW, 256);
      const dim3 grid((unsigned)std::min<long>((long)groups * per, 1L << 20));
#define ADX_BN_GROUPS(OC, RES, BITS) \
  bn_apply_groups_kernel<OC, RES, BITS><<<grid, dim3(256), 0, s>>>(rec.raw, scale, shift, identity, rec.out, rec.bits, L.cout, HW, groups, relu)
      if (rec.bits == nullptr) ADX_BN_GROUPS(true, 0, false);             // the map between a block's convs
      else if (out_cells && id_cells) ADX_BN_GROUPS(true, 2, true);       // a block's output, by layout of (output, identity)
      else if (out_cells) ADX_BN_GROUPS(true, 1, true);
      else if (id_cells) ADX_BN_GROUPS(false, 2, true);
      else ADX_BN_GROUPS(false, 1, true);
#undef ADX_BN_GROUPS
    } else if (bn_planes_ok(HW, rec.raw, identity, rec.out)) {
      const int planes = batch * L.cout;
      bn_apply_planes_kernel<<<dim3(std::min(ceil_div(planes, 4), 8192)), dim3(256), 0, s>>>(rec.raw, scale, shift, identity,
                                                                                            rec.out, L.cout, HW, planes, relu);
    } else {
      bn_apply_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(rec.raw, scale, shift, identity, rec.out,
                                                                            L.cout, HW, n, relu);
    }
    tape->recs.push_back(rec);
    return rec.out;
  };
  // The map between a block's convs as a cell tensor: when conv2's forward launch and its weight gradient both read cells
  // (the split-fp16 kernels; ADX_TRAIN_CELLS=0 / the exact-fp32 switches keep fp32 NCHW)
  auto mid_cells = [&](const ConvSpec& c1, const ConvSpec& c2, int OH, int OW) {
    return c1.cout % 8 == 0 && c2.k == 3 && c2.stride == 1 && c2.pad == 1 && conv2d_wgrad_hs_eligible(c2.cin, c2.cout, 3, 1, 1) &&
           conv2d_hs3x3_train_cells(c2, batch, OH, OW, kStatsPartFloats) &&
           (reinterpret_cast<uintptr_t>(workspace) & 15) == 0;
  };
  size_t ci = 0;
  // stem: conv -> batch statistics -> [BN apply + ReLU + MaxPool + arg-max code] in one pass over the conv output; the
  // post-BN stem map (rec.out of this record) is never formed -- nothing reads it: the backward re-derives the ReLU mask
  // from the conv output (mask mode 2)
  float* stem = conv_bn(r->convs[ci++], img, h, w, nullptr, 1, false);
  const int h1 = conv_out_dim(h, 7, 2, 3), w1 = conv_out_dim(w, 7, 2, 3);
  const int h2 = conv_out_dim(h1, 3, 2, 1), w2 = conv_out_dim(w1, 3, 2, 1);
  float* pooled = ws.take((size_t)batch * 64 * h2 * w2);
  uint8_t* pcode = reinterpret_cast<uint8_t*>(ws.take(((size_t)batch * 64 * h2 * w2 + 3) / 4));
  if (ws.ok && rc == ADX_OK) {
    bn_relu_pool_code_kernel<<<dim3(batch * 64, ceil_div(h2, 4 * kPoolRowsPerWave)), dim3(256), 0, s>>>(tape->recs[0].raw, scale, shift, pooled, pcode,
                                                                                 64, h1, w1, h2, w2);
  }
  tape->pool_in = stem; tape->pool_out = pooled; tape->ph = h1; tape->pw = w1; tape->poh = h2; tape->pow_ = w2;
  tape->pool_code = pcode;
  // A block's OUTPUT as a cell tensor: when everything that reads it reads cells -- the next block's conv1 (3x3 stride 1, or the
  // fused stride-2 conv1 + downsample launch), the weight gradients of those convs, and the residual add of the next block's
  // BatchNorm apply pass (group form).  The pooled map (block 0's input) and the last block's output (average pool, its
  // backward) stay fp32.  What is rounded: the identity the next block adds is hi + lo / 2^11 (22 bits) instead of the fp32
  // value -- what the inference executor does at every block.
  auto block_in_cells = [&](size_t b, size_t first_conv, int Hb, int Wb) {
    if (debug_switches().train_cells < 2 || b == 0 || b >= r->block_has_ds.size() || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0) return false;
    const ConvSpec& a1 = r->convs[first_conv];
    const ConvSpec& a2 = r->convs[first_conv + 1];
    if (a1.cin % 8 != 0 || a2.cout % 8 != 0 || (size_t)batch * a1.cin * Hb * Wb * sizeof(float) >= 0xC0000000u) return false;
    if (r->block_has_ds[b]) {
      const ConvSpec& ad = r->convs[first_conv + 2];
      return resnet_fuses_ds(a1, ad) && conv2d_wgrad_hs_eligible(a1.cin, a1.cout, 3, 2, 1);     // (the 1x1 gradient runs as that kernel's centre tap)
    }
    return a1.k == 3 && a1.stride == 1 && a1.pad == 1 && conv2d_wgrad_hs_eligible(a1.cin, a1.cout, 3, 1, 1) &&
           conv2d_hs3x3_train_cells(a1, batch, Hb, Wb, kStatsPartFloats);
  };
  float* cur = pooled;
  bool cur_cells = false;
  int H = h2, W = w2;
  for (size_t b = 0; b < r->block_has_ds.size(); ++b) {
    const ConvSpec& c1 = r->convs[ci++];
    const ConvSpec& c2 = r->convs[ci++];
    const int OH = conv_out_dim(H, 3, c1.stride, 1), OW = conv_out_dim(W, 3, c1.stride, 1);
    float* o1;
    const float* identity = cur;
    bool id_cells = cur_cells;
    const size_t o1_rec = tape->recs.size();         // conv1's record is the next one pushed
    if (r->block_has_ds[b] && resnet_fuses_ds(c1, r->convs[ci])) {
      // conv1 (3x3 stride 2) and the downsample (1x1 stride 2) read the same pixels: ONE launch leaves both raw outputs (the
      // downsample alone was a launch of the exact-fp32 1x1 kernel: 0.12 ms x 3 per step)
      const ConvSpec& ds = r->convs[ci++];
      const size_t n = (size_t)batch * c1.cout * OH * OW;
      float* raw1 = ws.take(n);
      float* rawd = ws.take(n);
      if (ws.ok && rc == ADX_OK)
        rc = conv2d_hs_launch_block_s2(c1, ds, cur, base + c1.o_w, nullptr, nullptr, raw1, base + ds.o_w, nullptr, nullptr, rawd,
                                       batch, H, W, s, cur_cells ? 1 : 0, 0, 0);
      o1 = conv_bn(c1, cur, H, W, nullptr, 1, true, raw1, cur_cells, mid_cells(c1, c2, OH, OW));
      identity = conv_bn(ds, cur, H, W, nullptr, 0, true, rawd, cur_cells);
      id_cells = false;
    } else {
      if (rc == ADX_OK && cur_cells && r->block_has_ds[b]) {
        set_error("adx_resnet_forward_train: a cell-layout block input needs the fused stride-2 launch");
        rc = ADX_ERR_STATE;
      }
      o1 = conv_bn(c1, cur, H, W, nullptr, 1, true, nullptr, cur_cells, mid_cells(c1, c2, OH, OW));
      if (r->block_has_ds[b]) { identity = conv_bn(r->convs[ci++], cur, H, W, nullptr, 0); id_cells = false; }
    }
    const bool out_cells = block_in_cells(b + 1, ci, OH, OW);
    cur = conv_bn(c2, o1, OH, OW, identity, 1, true, nullptr, tape->recs[o1_rec].out_cells, out_cells, id_cells);
    cur_cells = out_cells;
    H = OH; W = OW;
  }
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(ws.ok, "adx_resnet_forward_train: workspace of %zu bytes too small", workspace_bytes);
  tape->final_map = cur; tape->fh = H; tape->fw_ = W;
  tape->fwd_floats = ws.off;
  ADX_LAUNCH_CHECK();
  if (debug_switches().check_range) {
    // ADX_CHECK_RANGE=1, training: every tensor a split-fp16 conv multiplies -- the input of each conv, fp32 or cells -- scanned
    // after the pass (the tape holds them all); names the first conv whose operand leaves the fp16 range
    for (size_t i = 1; i < tape->recs.size(); ++i) {
      const adx_resnet_tape::Rec& q = tape->recs[i];
      if (q.x == nullptr || q.L == nullptr) continue;
      int rc2 = conv2d_range_check("the training forward's input of conv", (int)i, q.x, (size_t)batch * q.L->cin * q.H * q.W, q.x_cells, s);
      if (rc2 != ADX_OK) return rc2;
    }
  }
  return avgpool_fc_launch(cur, T[r->t_fcw], T[r->t_fcb], feature, batch, 512, H * W, r->out_dim, s);
}

// d_feature [batch][out_dim] -> one gradient per tensor slot of adx_resnet_pack's list (conv weight, bn gamma,
// bn beta written; running-stat slots untouched/NULL allowed), fc weight/bias included.
int32_t adx_resnet_backward_groups(const adx_resnet* r) { return r ? (int32_t)r->block_has_ds.size() + 2 : 0; }

int32_t adx_resnet_tensor_group(const adx_resnet* r, int32_t tensor) {
  if (r == nullptr || tensor < 0 || tensor >= r->n_tensors) return -1;
  const int nb = (int)r->block_has_ds.size();
  if (tensor == r->t_fcw || tensor == r->t_fcb) return 0;
  // convs: stem, then per block conv1, conv2, [downsample]
  size_t ci = 0;
  auto owns = [&](const adx::ConvSpec& L) { return tensor == L.t_w || tensor == L.t_g || tensor == L.t_b; };
  if (owns(r->convs[ci++])) return nb + 1;
  for (int b = 0; b < nb; ++b) {
    const int ncv = r->block_has_ds[b] ? 3 : 2;
    for (int k = 0; k < ncv; ++k)
      if (owns(r->convs[ci++])) return 1 + (nb - 1 - b);
  }
  return -1;      // running statistics: no gradient
}

int adx_resnet_backward(adx_resnet* r, const float* const* T, float* const* G, int32_t n_tensors, void* workspace,
                        size_t workspace_bytes, adx_resnet_tape* tape, const float* d_feature, adx_stream stream) {
  return adx_resnet_backward_events(r, T, G, n_tensors, workspace, workspace_bytes, tape, d_feature, nullptr, 0, stream);
}

int adx_resnet_backward_events(adx_resnet* r, const float* const* T, float* const* G, int32_t n_tensors, void* workspace,
                               size_t workspace_bytes, adx_resnet_tape* tape, const float* d_feature, void* const* events,
                               int32_t n_events, adx_stream stream) {
  ADX_REQUIRE(r && T && G && workspace && tape && d_feature, "adx_resnet_backward: null argument");
  ADX_REQUIRE(n_tensors == r->n_tensors && !tape->recs.empty(), "adx_resnet_backward: bad tape / tensor count");
  ADX_REQUIRE(events == nullptr || n_events == adx_resnet_backward_groups(r), "adx_resnet_backward_events: %d events, the backward has %d groups",
              n_events, adx_resnet_backward_groups(r));
  hipStream_t s = (hipStream_t)stream;
  // group g's event is recorded behind the last launch that writes a gradient of group g (adx_resnet_tensor_group): a
  // reduction of those gradients can start on another stream while the layers below are still being differentiated
  auto mark = [&](int group) -> int {
    if (events != nullptr && events[group] != nullptr) ADX_CHECK_HIP(hipEventRecord((hipEvent_t)events[group], s));
    return ADX_OK;
  };
  const int batch = tape->batch;
  Bump2 ws{(float*)workspace, tape->fwd_floats, workspace_bytes / sizeof(float)};
  const size_t n_convs = r->convs.size();
  double* sums_all = reinterpret_cast<double*>(ws.take(n_convs * 2 * 512 * 2));
  ADX_CHECK_HIP(hipMemsetAsync(sums_all, 0, sizeof(double) * n_convs * 2 * 512, s));
  for (const ConvSpec& L : r->convs) batch_fill_add(G[L.t_w], (size_t)L.cout * L.cin * L.k * L.k);
  {
    const int rf = batch_fill_flush(s);
    if (rf != ADX_OK) return rf;
  }
  uint32_t* amax = reinterpret_cast<uint32_t*>(ws.take(kAmaxPartials));   // per-workgroup max |draw| of the conv being differentiated
  uint32_t* dmax_all = reinterpret_cast<uint32_t*>(ws.take(n_convs * 512));      // per conv and channel: max |dz| of its incoming gradient (bits)
  float* bconsts = ws.take((size_t)512 * 8);                              // per-channel constants of the record being differentiated
  float* xscale = ws.take(2);                                             // {xs, 1 / xs} of its cell-layout draw
  if (ws.ok) ADX_CHECK_HIP(hipMemsetAsync(dmax_all, 0, sizeof(uint32_t) * n_convs * 512, s));
  float* wgrad9 = ws.take((size_t)512 * 256 * 9);     // 3x3 image of the largest 1x1 downsample gradient (conv2d_wgrad)
  struct PartsScope {
    ~PartsScope() { conv2d_wgrad_set_partials(nullptr, 0); }
  } parts_scope;
  if (const size_t pf = conv2d_wgrad_partials_floats()) conv2d_wgrad_set_partials(ws.take(pf), pf);     // ADX_WGRAD_DETERMINISTIC=1
  size_t big = 0, wmax = 0;
  for (auto& rec : tape->recs) {
    big = std::max(big, (size_t)batch * rec.L->cout * rec.OH * rec.OW);
    big = std::max(big, (size_t)batch * rec.L->cin * rec.H * rec.W);
    wmax = std::max(wmax, (size_t)rec.L->k * rec.L->k * rec.L->cout * rec.L->cin);
  }
  big = std::max(big, (size_t)batch * 64 * tape->ph * tape->pw);
  // rotating gradient buffers: g_out (incoming), dz, draw, dx candidates
  float* gb[5];
  for (auto& p : gb) p = ws.take(big);
  float* wimg = ws.take(wmax);
  float* wbuild = ws.take(wmax);     // the stride-2 data gradient's 2x2 weights before packing (16 cin cout <= 9 x 512 x 512)
  // data-gradient weight images of the stride-1 3x3 convs: all of them in one launch, each in its own slot (the others are
  // built where they are used, in `wimg`)
  std::vector<const float*> dgrad_img(r->convs.size(), nullptr);
  {
    std::vector<HsPackJob> jobs;
    for (auto& rec : tape->recs) {
      const ConvSpec& L = *rec.L;
      if (!(L.k == 3 && L.stride == 1)) continue;
      ConvSpec g{};
      g.cin = L.cout; g.cout = L.cin; g.k = L.k; g.stride = 1; g.pad = L.k - 1 - L.pad; g.cc = 16; g.cin_pad = L.cout; g.dgrad = 1;
      if (!conv2d_hs_pack_batchable(g, 1)) continue;
      float* slot = ws.take((size_t)L.k * L.k * L.cout * L.cin);
      if (!ws.ok) break;
      dgrad_img[&L - r->convs.data()] = slot;
      jobs.push_back(HsPackJob{T[L.t_w], slot, g.cout, g.cin_pad, g.cin, L.k * L.k, 1});
    }
    ADX_REQUIRE(ws.ok, "adx_resnet_backward: workspace of %zu bytes too small", workspace_bytes);
    if (!jobs.empty()) {
      const int rp = conv2d_hs_pack_many(jobs.data(), (int)jobs.size(), s);
      if (rp != ADX_OK) return rp;
    }
  }
  ADX_REQUIRE(ws.ok, "adx_resnet_backward: workspace of %zu bytes too small", workspace_bytes);
  int rc = ADX_OK;

  // fc + avgpool
  float* g_cur = gb[0];
  {
    const int HW = tape->fh * tape->fw_;
    ADX_CHECK_HIP(hipMemsetAsync(G[r->t_fcw], 0, sizeof(float) * (size_t)r->out_dim * 512, s));
    ADX_CHECK_HIP(hipMemsetAsync(G[r->t_fcb], 0, sizeof(float) * r->out_dim, s));
    avgpool_fc_bwd_kernel<<<dim3(batch * (512 / kPoolSlice)), dim3(256), 0, s>>>(tape->final_map, T[r->t_fcw], d_feature, g_cur, G[r->t_fcw],
                                                            G[r->t_fcb], 512, HW, r->out_dim);
    ADX_LAUNCH_CHECK();
    if (int rm = mark(0)) return rm;
  }
  // one conv+BN(+identity)(+ReLU) backward.  dout -> (dz for the identity path), d(conv input) accumulated
  // into dx (dx_has tells whether dx already holds a contribution).
  // `sums_ready`: the record whose BatchNorm-backward sums (sum dz, sum dz xhat) the last data-gradient conv already left in
  // its slot of sums_all -- the conv that PRODUCED that record's incoming gradient computed them in its epilogue (conv2d_hs.hip:
  // STATS == 2), so channel_sums_kernel<1>'s pass over that gradient is not needed.  `next`: the record dx flows into (null:
  // nobody's statistics can come from this launch -- dx is completed by a later launch, or feeds no BatchNorm).
  const adx_resnet_tape::Rec* sums_ready = nullptr;
  auto dgrad_spec = [](const ConvSpec& L) {
    ConvSpec g{};
    g.cin = L.cout; g.cout = L.cin; g.k = L.k; g.stride = 1; g.pad = L.k - 1 - L.pad; g.cc = 16; g.cin_pad = L.cout; g.dgrad = 1;
    return g;
  };
  // draw as a cell tensor (ADX_TRAIN_CELLS >= 4): the stride-1 3x3 convs whose weight and data gradient both run on the
  // split-fp16 kernels; the scale comes from a bound (bn_bwd_consts_kernel), `amax` is not produced
  auto draw_cells_of = [&](const adx_resnet_tape::Rec& rec, bool need_dx) {
    const ConvSpec& L = *rec.L;
    const int mask = !rec.relu ? 0 : (rec.identity != nullptr ? 1 : 2);
    return L.k == 3 && L.stride == 1 && L.pad == 1 && L.cout % 8 == 0 && need_dx && dgrad_img[&L - r->convs.data()] != nullptr &&
           conv2d_wgrad_hs_eligible(L.cin, L.cout, 3, 1, 1) && conv2d_hs3x3_dgrad_cells(dgrad_spec(L), batch, rec.OH, rec.OW) &&
           (mask != 1 || rec.bits != nullptr);
  };
  // res_bits: dx's residual is `dx` itself (in place) passed through these mask bits -- the identity path of a block without a
  // downsample: d(block input) += d(block output) where the block's output was positive; only with `next` (statistics epilogue)
  auto conv_bn_bwd = [&](const adx_resnet_tape::Rec& rec, const float* dout, float* dz_keep, float* draw, float* dx,
                         bool dx_has, bool need_dx, const adx_resnet_tape::Rec* next = nullptr, const uint8_t* res_bits = nullptr) -> int {
    const ConvSpec& L = *rec.L;
    const int HW = rec.OH * rec.OW;
    const size_t n = (size_t)batch * L.cout * HW;
    const double count = (double)batch * HW;
    double* sums = sums_all + (size_t)(&L - r->convs.data()) * 2 * 512;
    // ReLU mask: straight after BN it is re-derived from the conv output (one tensor read less in both passes)
    const int mask = !rec.relu ? 0 : (rec.identity != nullptr ? 1 : 2);
    uint32_t* const dmax = dmax_all + (size_t)(&L - r->convs.data()) * 512;
    if (sums_ready != &rec)
      channel_sums_kernel<1><<<dim3(batch * L.cout), dim3(256), 0, s>>>(dout, rec.out, rec.raw, rec.mean, rec.rstd, sums,
                                                                        L.cout, HW, mask, T[L.t_g], T[L.t_b], rec.bits, dmax);
    sums_ready = nullptr;
    const bool draw_cells = draw_cells_of(rec, need_dx) && (reinterpret_cast<uintptr_t>(draw) & 15) == 0;
    int n_amax;
    if (draw_cells) {
      n_amax = -1;
      bn_bwd_consts_kernel<<<dim3(1), dim3(256), 0, s>>>(sums, rec.mean, rec.rstd, T[L.t_g], T[L.t_b], dmax, count, L.cout, bconsts, xscale,
                                                         G[L.t_g], G[L.t_b]);
      const int groups = batch * (L.cout / 8), per = ceil_div(HW, 256);
      const dim3 grid((unsigned)std::min<long>((long)groups * per, 1L << 20));
#define ADX_BWD_GROUPS(MASK, KEEP) \
  bn_bwd_apply_groups_kernel<MASK, KEEP><<<grid, dim3(256), 0, s>>>(dout, rec.raw, rec.bits, bconsts, xscale, reinterpret_cast<u32x4*>(draw), \
                                                                    dz_keep, L.cout, HW, groups)
      if (mask == 1 && dz_keep != nullptr) ADX_BWD_GROUPS(1, true);
      else if (mask == 1) ADX_BWD_GROUPS(1, false);
      else if (mask == 2 && dz_keep != nullptr) ADX_BWD_GROUPS(2, true);
      else if (mask == 2) ADX_BWD_GROUPS(2, false);
      else if (dz_keep != nullptr) ADX_BWD_GROUPS(0, true);
      else ADX_BWD_GROUPS(0, false);
#undef ADX_BWD_GROUPS
    } else if (bn_planes_ok(HW, dout, rec.raw, draw) && bn_planes_ok(HW, rec.out, dz_keep, nullptr)) {
      const int planes = batch * L.cout;
      n_amax = (int)std::min<size_t>(kAmaxPartials, (size_t)ceil_div(planes, 4));
      bn_bwd_apply_planes_kernel<<<dim3(n_amax), dim3(256), 0, s>>>(dout, rec.out, rec.raw, rec.mean, rec.rstd, T[L.t_g], sums,
                                                                    draw, dz_keep, L.cout, HW, planes, count, mask, amax, T[L.t_b],
                                                                    G[L.t_g], G[L.t_b], rec.bits);
    } else {
      n_amax = (int)std::min<size_t>(kAmaxPartials, (n + 255) / 256);
      bn_bwd_apply_kernel<<<dim3(n_amax), dim3(256), 0, s>>>(
          dout, rec.out, rec.raw, rec.mean, rec.rstd, T[L.t_g], sums, draw, dz_keep, L.cout, HW, n, count, mask, amax, T[L.t_b],
          G[L.t_g], G[L.t_b], rec.bits);
    }
    ADX_LAUNCH_CHECK();
    const uint32_t* const range = draw_cells ? reinterpret_cast<const uint32_t*>(xscale) : amax;      // (n_amax < 0: the scale itself)
    int rc2 = conv2d_wgrad(rec.x, draw, G[L.t_w], batch, L.cin, rec.H, rec.W, L.cout, L.k, L.stride, L.pad, s, range, n_amax, false, wgrad9,
                           rec.x_cells, draw_cells);
    if (rc2 != ADX_OK || !need_dx) return rc2;
    // data gradient
    const ConvSpec g = dgrad_spec(L);
    ADX_REQUIRE(res_bits == nullptr || (next != nullptr && dx_has && dgrad_img[&L - r->convs.data()] != nullptr && tape->stats_part != nullptr),
                "adx_resnet_backward: a masked residual needs the statistics epilogue of the pipelined data gradient");
    if (const float* pre = dgrad_img[&L - r->convs.data()]) {      // stride-1 3x3: image packed with the others at the start
      const int next_mask = next == nullptr || !next->relu ? 0 : (next->identity != nullptr ? 1 : 2);
      if (next_mask != 0 && tape->stats_part != nullptr) {
        const ConvSpec& Ln = *next->L;
        ADX_REQUIRE(Ln.cout == L.cin && next->OH == rec.H && next->OW == rec.W, "adx_resnet_backward: consumer record does not match dx");
        const BnBwdStats bst{next->raw, next->out, next->mean, next->rstd, T[Ln.t_g], T[Ln.t_b], next_mask, next->bits, res_bits};
        int stats_p = 0;
        rc2 = conv2d_launch_raw(g, draw, pre, nullptr, nullptr, dx_has ? dx : nullptr, dx, batch, rec.OH, rec.OW, 0, s, range, n_amax,
                                tape->stats_part, kStatsPartFloats, &stats_p, draw_cells ? (kFmtXCells | kFmtXScaled) : 0, &bst);
        if (rc2 == ADX_OK && stats_p > 0) {
          stats_reduce_kernel<<<dim3(Ln.cout + 1), dim3(256), 0, s>>>(tape->stats_part, sums_all + (size_t)(&Ln - r->convs.data()) * 2 * 512,
                                                                     stats_p, dmax_all + (size_t)(&Ln - r->convs.data()) * 512, Ln.cout);
          sums_ready = next;
        }
        return rc2;
      }
      return conv2d_launch_raw(g, draw, pre, nullptr, nullptr, dx_has ? dx : nullptr, dx, batch, rec.OH, rec.OW, 0, s, range, n_amax, nullptr, 0,
                               nullptr, draw_cells ? (kFmtXCells | kFmtXScaled) : 0);
    }
    rc2 = conv2d_pack_spec(g, T[L.t_w], wimg, 1, s);
    if (rc2 != ADX_OK) return rc2;
    if (L.stride == 1) {
      return conv2d_launch_raw(g, draw, wimg, nullptr, nullptr, dx_has ? dx : nullptr, dx, batch, rec.OH, rec.OW, 0, s, amax, n_amax);
    }
    if (L.k == 3 && conv2d_hs_dgrad_s2_eligible(L.cin, L.cout) && (size_t)16 * L.cin * L.cout <= wmax) {
      // one stride-1 2x2 conv on the low-resolution gradient with a depth-to-space store (conv2d_hs_dgrad_s2): 16 tap-products
      // per four input pixels instead of the 36 of the zero-dilated form below, and no dilated tensor
      return conv2d_hs_dgrad_s2(T[L.t_w], draw, dx, dx_has ? 1 : 0, batch, L.cin, L.cout, rec.H, rec.W, wbuild, wimg, amax, n_amax, s);
    }
    if (L.k == 3) {
      // zero-dilate draw to the input resolution, then an ordinary 3x3 stride-1 conv with the flipped weights
      float* dil = gb[4];
      ADX_CHECK_HIP(hipMemsetAsync(dil, 0, sizeof(float) * (size_t)batch * L.cout * rec.H * rec.W, s));
      dilate2_kernel<<<dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s>>>(draw, dil, batch * L.cout, rec.OH, rec.OW, rec.H,
                                                                          rec.W, 0);
      ADX_LAUNCH_CHECK();
      return conv2d_launch_raw(g, dil, wimg, nullptr, nullptr, dx_has ? dx : nullptr, dx, batch, rec.H, rec.W, 0, s, amax, n_amax);
    }
    // 1x1 stride 2: a 1x1 stride-1 conv at the output resolution, scattered to the even input positions
    float* t = gb[4];
    rc2 = conv2d_launch_raw(g, draw, wimg, nullptr, nullptr, nullptr, t, batch, rec.OH, rec.OW, 0, s, amax, n_amax);
    if (rc2 != ADX_OK) return rc2;
    if (!dx_has) ADX_CHECK_HIP(hipMemsetAsync(dx, 0, sizeof(float) * (size_t)batch * L.cin * rec.H * rec.W, s));
    const size_t nt = (size_t)batch * L.cin * HW;
    dilate2_kernel<<<dim3((unsigned)((nt + 255) / 256)), dim3(256), 0, s>>>(t, dx, batch * L.cin, rec.OH, rec.OW, rec.H, rec.W, 1);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  };

  // BasicBlocks in reverse.  recs: [stem, (c1, [ds], c2) per block] in forward launch order
  size_t ri = tape->recs.size();
  for (size_t b = r->block_has_ds.size(); b-- > 0 && rc == ADX_OK;) {
    const bool ds = r->block_has_ds[b] != 0;
    const adx_resnet_tape::Rec& c2 = tape->recs[--ri];
    const adx_resnet_tape::Rec* dsr = ds ? &tape->recs[--ri] : nullptr;
    const adx_resnet_tape::Rec& c1 = tape->recs[--ri];
    // buffers: g_cur = d(block out).  dz2 (identity gradient), draw scratch, do1, dx
    float* others[4];
    int k = 0;
    for (auto p : gb) if (p != g_cur && p != gb[4] && k < 4) others[k++] = p;
    float* dz2 = others[0]; float* draw = others[1]; float* do1 = others[2];
    // A block without a downsample, not the first: the identity path's gradient -- d(block out) where the block's output was
    // positive -- is not written as a tensor (dz2): conv1's data gradient adds d(block out) IN PLACE through the mask bits
    const adx_resnet_tape::Rec* prev = b > 0 ? &tape->recs[ri - 1] : nullptr;
    const bool masked_res = !ds && prev != nullptr && prev->relu && c2.bits != nullptr && tape->stats_part != nullptr &&
                            dgrad_img[c1.L - r->convs.data()] != nullptr &&
                            conv2d_hs3x3_dgrad_stats(dgrad_spec(*c1.L), batch, c1.OH, c1.OW,
                                                     draw_cells_of(c1, true) && (reinterpret_cast<uintptr_t>(draw) & 15) == 0, kStatsPartFloats);
    rc = conv_bn_bwd(c2, g_cur, masked_res ? nullptr : dz2, draw, do1, false, true, &c1);   // -> do1 = d(o1) (c1's incoming gradient), dz2 = masked dout
    if (rc != ADX_OK) break;
    float* dx = g_cur;                                                   // d(block out) is dead now: reuse for d(block in)
    if (masked_res) {
      rc = conv_bn_bwd(c1, do1, nullptr, draw, g_cur, true, true, prev, c2.bits);
    } else if (ds) {
      rc = conv_bn_bwd(c1, do1, nullptr, draw, dx, false, true);         // main path: writes every pixel of d(block in)
      if (rc != ADX_OK) break;
      rc = conv_bn_bwd(*dsr, dz2, nullptr, draw, dx, true, true);        // identity path through the downsample conv adds to
                                                                         // the even pixels (no clearing pass over dx)
    } else {
      // identity gradient is dz2 itself: main path = conv(...) + res(dz2).  The result is the incoming gradient of the block
      // before this one (its conv2, ReLU after the residual add); block 0's flows into the max-pool instead
      rc = conv_bn_bwd(c1, do1, nullptr, draw, dz2, true, true, b > 0 ? &tape->recs[ri - 1] : nullptr);
      dx = dz2;
    }
    g_cur = dx;
    if (rc == ADX_OK) rc = mark(1 + (int)(r->block_has_ds.size() - 1 - b));
  }
  if (rc != ADX_OK) return rc;
  // maxpool, then the stem (no data gradient: the image needs none)
  {
    // the max-pool backward is folded into the stem's two BatchNorm-backward passes (stem_pool_bn_bwd_kernel): d(stem map)
    // is gathered through the forward's arg-max codes on the fly and never written
    const uint8_t* code = tape->pool_code;     // written by the forward's fused stem tail
    float* draw = nullptr;
    for (auto p : gb) if (p != g_cur && p != gb[4]) { draw = p; break; }
    ADX_REQUIRE(code != nullptr && draw != nullptr && g_cur != gb[4], "adx_resnet_backward: scratch buffer clash");
    const adx_resnet_tape::Rec& st = tape->recs[0];
    const ConvSpec& L = *st.L;
    ADX_REQUIRE(st.relu && st.identity == nullptr && L.cout == 64 && st.OH == tape->ph && st.OW == tape->pw,
                "adx_resnet_backward: the stem record does not match the pooled map");
    double* sums = sums_all + (size_t)(&L - r->convs.data()) * 2 * 512;
    const double count = (double)batch * st.OH * st.OW;
    stem_pool_bn_bwd_kernel<0><<<dim3(batch * 64), dim3(256), 0, s>>>(code, g_cur, st.raw, st.mean, st.rstd, T[L.t_g], T[L.t_b], sums,
                                                                      nullptr, 64, tape->ph, tape->pw, tape->poh, tape->pow_, count,
                                                                      nullptr, nullptr, nullptr, 0, tape->pool_out);
    // the apply pass also leaves max |draw| (atomic maxima over a cleared table): the stem's weight gradient runs on the fp16
    // matrix cores like the others and needs the gradient's range
    const bool stem_hs = conv2d_wgrad_stem_hs_eligible(L.cin, L.cout, L.k, L.stride, L.pad);
    if (stem_hs) ADX_CHECK_HIP(hipMemsetAsync(amax, 0, sizeof(uint32_t) * kAmaxPartials, s));
    stem_pool_bn_bwd_kernel<1><<<dim3(batch * 64, ceil_div(tape->ph, 4 * kStemBwdRows)), dim3(256), 0, s>>>(
        code, g_cur, st.raw, st.mean, st.rstd, T[L.t_g], T[L.t_b], sums, draw, 64, tape->ph, tape->pw, tape->poh, tape->pow_, count,
        G[L.t_g], G[L.t_b], stem_hs ? amax : nullptr, (int)kAmaxPartials);
    ADX_LAUNCH_CHECK();
    rc = conv2d_wgrad(st.x, draw, G[L.t_w], batch, L.cin, st.H, st.W, L.cout, L.k, L.stride, L.pad, s, stem_hs ? amax : nullptr,
                      stem_hs ? (int)kAmaxPartials : 0, false, wgrad9);
    if (rc == ADX_OK) rc = mark((int)r->block_has_ds.size() + 1);
  }
  return rc;
}

}  // extern "C"
