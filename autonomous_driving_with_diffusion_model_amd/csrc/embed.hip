// Diffusion-time / goal embedding: SinusoidalPosEmb -> Linear -> Mish -> Linear (+ cond_mlp),
// concatenated with the perception feature and passed through the Mish that opens every
// residual block's time_mlp.  modeling/helpers.py:62-74, modeling/temporal.py:34-36,88-98,205-213.
//
// All 16 residual blocks apply Mish to the SAME cond vector before their own Linear, so the
// Mish is done once here; the 16 Linears run as one GEMM (see unet.hip).
#include "adx_common.h"

namespace adx {

constexpr int kMaxDim = 256;

__global__ void __launch_bounds__(256) embed_kernel(const adx_embed_weights w, const int dim,
                                                     const int64_t* __restrict__ t, const int t_rows,
                                                     const float* __restrict__ cond,
                                                     const float* __restrict__ feat, const int feat_rows,
                                                     float* __restrict__ time_embed, float* __restrict__ mish_cond) {
  __shared__ float e[kMaxDim];
  __shared__ float h[4 * kMaxDim];
  __shared__ float te[kMaxDim];
  __shared__ float ch[kMaxDim];
  const int row = blockIdx.x, tid = threadIdx.x;
  const int half = dim / 2, hid = 4 * dim;
  const float tval = (float)t[row % t_rows];  // int64 * fp32 -> fp32 (helpers.py:71)
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
    h[j] = mish_f(acc);
  }
  __syncthreads();
  for (int j = tid; j < dim; j += 256) {
    float acc = w.b3[j];
    const float* wr = w.w3 + (size_t)j * hid;
    for (int i = 0; i < hid; ++i) acc += wr[i] * h[i];
    te[j] = acc;
  }
  if (w.cw0 != nullptr) {
    // FREE_GUIDANCE: time_embed += cond_mlp(cond); cond == None means zeros, whose embedding
    // is cond_mlp(0) and not 0 (temporal.py:207,212)
    const float c0 = cond != nullptr ? cond[2 * row] : 0.f;
    const float c1 = cond != nullptr ? cond[2 * row + 1] : 0.f;
    for (int j = tid; j < dim; j += 256) ch[j] = mish_f(w.cw0[2 * j] * c0 + w.cw0[2 * j + 1] * c1 + w.cb0[j]);
    __syncthreads();
    for (int j = tid; j < dim; j += 256) {
      float acc = w.cb2[j];
      const float* wr = w.cw2 + (size_t)j * dim;
      for (int i = 0; i < dim; ++i) acc += wr[i] * ch[i];
      te[j] += acc;  // same thread wrote te[j] above
    }
  }
  __syncthreads();
  for (int j = tid; j < dim; j += 256) {
    const float v = te[j];
    time_embed[(size_t)row * dim + j] = v;
    mish_cond[(size_t)row * 2 * dim + j] = mish_f(v);
    mish_cond[(size_t)row * 2 * dim + dim + j] = mish_f(feat[(size_t)(row % feat_rows) * dim + j]);
  }
}

int embed_forward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                  const float* feat, int feat_rows, int rows, float* time_embed, float* mish_cond, hipStream_t s) {
  ADX_REQUIRE(w != nullptr && w->freqs && w->w1 && w->b1 && w->w3 && w->b3, "embed: missing time_mlp weights");
  ADX_REQUIRE(dim >= 4 && dim <= kMaxDim && dim % 2 == 0, "embed: dim %d unsupported (<= %d, even)", dim, kMaxDim);
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
