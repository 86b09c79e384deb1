// Diffusion-time / goal embedding: SinusoidalPosEmb -> Linear -> Mish -> Linear (+ cond_mlp),
// concatenated with the perception feature and passed through the Mish that opens every
// residual block's time_mlp.  modeling/helpers.py:62-74, modeling/temporal.py:34-36,88-98,205-213.
//
// All 16 residual blocks apply Mish to the SAME cond vector before their own Linear, so the
// Mish is done once here; the 16 Linears run as one GEMM (see unet.hip).
#include "adx_common.h"

namespace adx {

constexpr int kMaxDim = 256;

// sum over n (a multiple of 4) of w[i] * x[i]: w in global memory (16-byte aligned), x in LDS; eight 16-byte loads are
// issued before the first is consumed, so a 64-long row costs one memory round trip instead of sixteen
__device__ __forceinline__ float dot_f4(const float* __restrict__ w, const float* x, int n) {
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
  const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
  const int n4 = n >> 2;
  f32x4 a4 = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int i0 = 0; i0 < n4; i0 += 8) {
    f32x4 wv[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) wv[u] = w4[min(i0 + u, n4 - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u < n4) a4 += wv[u] * x4[i0 + u];
  }
  return (a4[0] + a4[1]) + (a4[2] + a4[3]);
}

__global__ void __launch_bounds__(256) embed_kernel(const adx_embed_weights w, const int dim,
                                                     const int64_t* __restrict__ t, const int t_rows,
                                                     const float* __restrict__ cond,
                                                     const float* __restrict__ feat, const int feat_rows,
                                                     float* __restrict__ time_embed, float* __restrict__ mish_cond) {
  __shared__ __attribute__((aligned(16))) float e[kMaxDim];
  __shared__ __attribute__((aligned(16))) float h[4 * kMaxDim];
  __shared__ float parts[4 * kMaxDim];
  __shared__ float te[kMaxDim];
  __shared__ __attribute__((aligned(16))) float ch[kMaxDim];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int half = dim / 2, hid = 4 * dim;
  const float tval = (float)t[row % t_rows];  // int64 * fp32 -> fp32 (helpers.py:71)
  for (int i = tid; i < dim; i += 256) {
    const int fi = i < half ? i : i - half;
    const float arg = tval * w.freqs[fi];
    e[i] = i < half ? sinf(arg) : cosf(arg);
  }
  __syncthreads();
  // time_mlp.1: hid outputs, one per thread
  for (int j = tid; j < hid; j += 256) h[j] = mish_f(w.b1[j] + dot_f4(w.w1 + (size_t)j * dim, e, dim));
  __syncthreads();
  // time_mlp.3: dim outputs x hid inputs; all 256 threads work: thread (j, part) sums a quarter of the row, the four
  // parts meet in LDS and are added in a fixed order
  for (int lin = tid; lin < 4 * dim; lin += 256) {
    const int jj = lin % dim, pp = lin / dim;
    const int seg = hid / 4;
    parts[lin] = dot_f4(w.w3 + (size_t)jj * hid + pp * seg, h + pp * seg, seg);
  }
  __syncthreads();
  for (int jj = tid; jj < dim; jj += 256)
    te[jj] = w.b3[jj] + ((parts[jj] + parts[dim + jj]) + (parts[2 * dim + jj] + parts[3 * dim + jj]));
  if (w.cw0 != nullptr) {
    // FREE_GUIDANCE: time_embed += cond_mlp(cond); cond == None means zeros, whose embedding
    // is cond_mlp(0) and not 0 (temporal.py:207,212)
    const float c0 = cond != nullptr ? cond[2 * row] : 0.f;
    const float c1 = cond != nullptr ? cond[2 * row + 1] : 0.f;
    for (int j = tid; j < dim; j += 256) ch[j] = mish_f(w.cw0[2 * j] * c0 + w.cw0[2 * j + 1] * c1 + w.cb0[j]);
    __syncthreads();                       // also orders the reads of `parts` above before they are overwritten
    // cond_mlp.2: dim x dim, four parts per output like time_mlp.3 (dim % 16 == 0)
    for (int lin = tid; lin < 4 * dim; lin += 256) {
      const int jj = lin % dim, pp = lin / dim;
      const int seg = dim / 4;
      parts[lin] = dot_f4(w.cw2 + (size_t)jj * dim + pp * seg, ch + pp * seg, seg);
    }
    __syncthreads();
    for (int jj = tid; jj < dim; jj += 256)   // same thread wrote te[jj] above
      te[jj] += w.cb2[jj] + ((parts[jj] + parts[dim + jj]) + (parts[2 * dim + jj] + parts[3 * dim + jj]));
  }
  __syncthreads();
  for (int j = tid; j < dim; j += 256) {
    const float v = te[j];
    time_embed[(size_t)row * dim + j] = v;
    mish_cond[(size_t)row * 2 * dim + j] = mish_f(v);
    mish_cond[(size_t)row * 2 * dim + dim + j] = mish_f(feat[(size_t)(row % feat_rows) * dim + j]);
  }
}

__device__ __forceinline__ float mish_grad_e(float x) {
  if (x > 20.f) return 1.f;
  const float e = expf(x);
  const float n = e * (e + 2.f);
  const float t = n / (n + 2.f);
  return t + x * (1.f - t * t) * (e / (1.f + e));
}

// Backward of embed_kernel for one row: recomputes the small MLPs, then
//   d cat = d mish_cond * mish'(cat(time_embed, feat));  d feat = d cat[dim:]  (written)
//   time_mlp / cond_mlp parameter gradients: atomically accumulated (caller zeroes them)
struct EmbedGrads { float* w1; float* b1; float* w3; float* b3; float* cw0; float* cb0; float* cw2; float* cb2; };

__global__ void __launch_bounds__(256) embed_bwd_kernel(const adx_embed_weights w, const EmbedGrads g, const int dim,
                                                         const int64_t* __restrict__ t, const int t_rows,
                                                         const float* __restrict__ cond, const float* __restrict__ feat,
                                                         const int feat_rows, const float* __restrict__ time_embed,
                                                         const float* __restrict__ dmc, const float* __restrict__ dte_extra,
                                                         float* __restrict__ dfeat) {
  __shared__ float e[kMaxDim];
  __shared__ float a1[4 * kMaxDim];   // pre-activation of time_mlp.1, later d(a1)
  __shared__ float h[4 * kMaxDim];
  __shared__ float dte[kMaxDim];
  __shared__ float ca[kMaxDim];       // pre-activation of cond_mlp.0, later d(ca)
  __shared__ float ch[kMaxDim];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int half = dim / 2, hid = 4 * dim;
  const float tval = (float)t[row % t_rows];
  for (int i = tid; i < dim; i += 256) {
    const int fi = i < half ? i : i - half;
    const float arg = tval * w.freqs[fi];
    e[i] = i < half ? sinf(arg) : cosf(arg);
  }
  __syncthreads();
  for (int j = tid; j < hid; j += 256) {
    float acc = w.b1[j];
    const float* wr = w.w1 + (size_t)j * dim;
    for (int i = 0; i < dim; ++i) acc += wr[i] * e[i];
    a1[j] = acc;
    h[j] = mish_f(acc);
  }
  const float c0 = (w.cw0 != nullptr && cond != nullptr) ? cond[2 * row] : 0.f;
  const float c1 = (w.cw0 != nullptr && cond != nullptr) ? cond[2 * row + 1] : 0.f;
  if (w.cw0 != nullptr) {
    for (int j = tid; j < dim; j += 256) {
      const float v = w.cw0[2 * j] * c0 + w.cw0[2 * j + 1] * c1 + w.cb0[j];
      ca[j] = v;
      ch[j] = mish_f(v);
    }
  }
  // d cat = d mc * mish'(cat)
  for (int j = tid; j < dim; j += 256) {
    const float te = time_embed[(size_t)row * dim + j];
    dte[j] = dmc[(size_t)row * 2 * dim + j] * mish_grad_e(te) +
             (dte_extra != nullptr ? dte_extra[(size_t)row * dim + j] : 0.f);   // + gradient from TrajPredict
    const float f = feat[(size_t)(row % feat_rows) * dim + j];
    const float df = dmc[(size_t)row * 2 * dim + dim + j] * mish_grad_e(f);
    if (feat_rows == gridDim.x) dfeat[(size_t)row * dim + j] = df;
    else atomicAdd(dfeat + (size_t)(row % feat_rows) * dim + j, df);
  }
  __syncthreads();
  // time_mlp.3: te0 = W3 h + b3
  for (int idx = tid; idx < dim * hid; idx += 256) {
    const int j = idx / hid, i = idx - j * hid;
    atomicAdd(g.w3 + idx, dte[j] * h[i]);
  }
  for (int j = tid; j < dim; j += 256) atomicAdd(g.b3 + j, dte[j]);
  __syncthreads();
  for (int i = tid; i < hid; i += 256) {
    float acc = 0.f;
    for (int j = 0; j < dim; ++j) acc += dte[j] * w.w3[(size_t)j * hid + i];
    a1[i] = acc * mish_grad_e(a1[i]);   // d(a1)
  }
  __syncthreads();
  for (int idx = tid; idx < hid * dim; idx += 256) {
    const int i = idx / dim, k = idx - i * dim;
    atomicAdd(g.w1 + idx, a1[i] * e[k]);
  }
  for (int i = tid; i < hid; i += 256) atomicAdd(g.b1 + i, a1[i]);
  if (w.cw0 != nullptr) {
    // cond_mlp.2: tc = cw2 ch + cb2;  cond_mlp.0: ca = cw0 c + cb0
    for (int idx = tid; idx < dim * dim; idx += 256) {
      const int j = idx / dim, i = idx - j * dim;
      atomicAdd(g.cw2 + idx, dte[j] * ch[i]);
    }
    for (int j = tid; j < dim; j += 256) atomicAdd(g.cb2 + j, dte[j]);
    __syncthreads();
    for (int i = tid; i < dim; i += 256) {
      float acc = 0.f;
      for (int j = 0; j < dim; ++j) acc += dte[j] * w.cw2[(size_t)j * dim + i];
      const float d = acc * mish_grad_e(ca[i]);
      atomicAdd(g.cw0 + 2 * i, d * c0);
      atomicAdd(g.cw0 + 2 * i + 1, d * c1);
      atomicAdd(g.cb0 + i, d);
    }
  }
}

int embed_backward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                   const float* feat, int feat_rows, int rows, const float* time_embed, const float* d_mish_cond,
                   const float* d_time_embed_extra, float* d_feat, float* const* grads, hipStream_t s) {
  ADX_REQUIRE(w && t && feat && time_embed && d_mish_cond && d_feat && grads, "embed_backward: null argument");
  ADX_REQUIRE(dim >= 4 && dim <= kMaxDim && dim % 2 == 0, "embed_backward: dim %d unsupported", dim);
  EmbedGrads g{grads[0], grads[1], grads[2], grads[3], grads[4], grads[5], grads[6], grads[7]};
  ADX_REQUIRE(g.w1 && g.b1 && g.w3 && g.b3, "embed_backward: time_mlp gradient buffers missing");
  ADX_REQUIRE(w->cw0 == nullptr || (g.cw0 && g.cb0 && g.cw2 && g.cb2), "embed_backward: cond_mlp gradient buffers missing");
  const size_t hd = (size_t)4 * dim * dim;
  ADX_CHECK_HIP(hipMemsetAsync(g.w1, 0, sizeof(float) * hd, s));
  ADX_CHECK_HIP(hipMemsetAsync(g.b1, 0, sizeof(float) * 4 * dim, s));
  ADX_CHECK_HIP(hipMemsetAsync(g.w3, 0, sizeof(float) * hd, s));
  ADX_CHECK_HIP(hipMemsetAsync(g.b3, 0, sizeof(float) * dim, s));
  if (w->cw0 != nullptr) {
    ADX_CHECK_HIP(hipMemsetAsync(g.cw0, 0, sizeof(float) * 2 * dim, s));
    ADX_CHECK_HIP(hipMemsetAsync(g.cb0, 0, sizeof(float) * dim, s));
    ADX_CHECK_HIP(hipMemsetAsync(g.cw2, 0, sizeof(float) * dim * dim, s));
    ADX_CHECK_HIP(hipMemsetAsync(g.cb2, 0, sizeof(float) * dim, s));
  }
  if (feat_rows != rows) ADX_CHECK_HIP(hipMemsetAsync(d_feat, 0, sizeof(float) * (size_t)feat_rows * dim, s));
  embed_bwd_kernel<<<dim3(rows), dim3(256), 0, s>>>(*w, g, dim, t, t_rows, cond, feat, feat_rows, time_embed,
                                                    d_mish_cond, d_time_embed_extra, d_feat);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int embed_forward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                  const float* feat, int feat_rows, int rows, float* time_embed, float* mish_cond, hipStream_t s) {
  ADX_REQUIRE(w != nullptr && w->freqs && w->w1 && w->b1 && w->w3 && w->b3, "embed: missing time_mlp weights");
  ADX_REQUIRE(dim >= 16 && dim <= kMaxDim && dim % 16 == 0, "embed: dim %d unsupported (<= %d, multiple of 16)", dim, kMaxDim);
  ADX_REQUIRE(w->cw2 == nullptr || (reinterpret_cast<uintptr_t>(w->cw2) & 15) == 0, "embed: cond_mlp.2 weight must be 16-byte aligned");
  ADX_REQUIRE(((reinterpret_cast<uintptr_t>(w->w1) | reinterpret_cast<uintptr_t>(w->w3)) & 15) == 0,
              "embed: time_mlp weights must be 16-byte aligned");
  ADX_REQUIRE(rows >= 1 && t_rows >= 1 && feat_rows >= 1, "embed: empty batch");
  ADX_REQUIRE(rows % t_rows == 0 && rows % feat_rows == 0, "embed: rows %d not a multiple of t_rows %d / feat_rows %d",
              rows, t_rows, feat_rows);
  ADX_REQUIRE(t && feat && time_embed && mish_cond, "embed: null tensor");
  ADX_REQUIRE((w->cw0 == nullptr) == (w->cw2 == nullptr), "embed: cond_mlp weights must be all set or all null");
  embed_kernel<<<dim3(rows), dim3(256), 0, s>>>(*w, dim, t, t_rows, cond, feat, feat_rows, time_embed, mish_cond);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
