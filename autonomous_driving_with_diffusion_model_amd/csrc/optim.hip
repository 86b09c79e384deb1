// Fused optimizer step of the training loop (train.py:252-261): for every parameter tensor, in one launch,
//   grad = nan_to_num(grad, nan=0, posinf=1e5, neginf=-1e5)            train.py:252-255
//   AdamW (decoupled weight decay, bias-corrected)                       torch.optim.AdamW, train.py:170,256
//   EMA shadow -= (1 - decay) * (shadow - param)                         diffusers EMAModel.step, train.py:261
// The reference runs these as ~5 passes over 37.35 M parameters (nan_to_num, foreach AdamW ops, EMA);
// here each of param / grad / m / v / shadow is read once and written once: 28 B per parameter, HBM-bound.
#include "adx_common.h"

namespace adx {

struct OptTensor {
  float* p; const float* g; float* m; float* v; float* ema;
  int64_t n;
};

struct OptArgs {
  const OptTensor* tensors;     // device table
  const int32_t* block_tensor;  // per block: tensor index
  const int32_t* block_chunk;   // per block: chunk index inside the tensor
  float lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, ema_decay, grad_scale;
  int use_ema, sanitize;
};

constexpr int kOptChunk = 256 * 8;   // elements per block

__global__ void __launch_bounds__(256) adamw_ema_kernel(const OptArgs a) {
  const OptTensor t = a.tensors[a.block_tensor[blockIdx.x]];
  const int64_t base = (int64_t)a.block_chunk[blockIdx.x] * kOptChunk;
  const float step_size = a.lr / a.bc1;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int64_t i = base + threadIdx.x + 256 * k;
    if (i >= t.n) break;
    float g = t.g[i] * a.grad_scale;       // 1 / world when the buckets carry the sum over the ranks; 1 otherwise (exact)
    if (a.sanitize) {
      if (g != g) g = 0.f;
      else if (g == INFINITY) g = 1e5f;
      else if (g == -INFINITY) g = -1e5f;
    }
    float p = t.p[i];
    p = p * (1.f - a.lr * a.weight_decay);
    const float m = a.beta1 * t.m[i] + (1.f - a.beta1) * g;
    const float v = a.beta2 * t.v[i] + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    p = p - step_size * (m / denom);
    t.m[i] = m;
    t.v[i] = v;
    t.p[i] = p;
    if (a.use_ema) {
      const float s = t.ema[i];
      t.ema[i] = s - (1.f - a.ema_decay) * (s - p);
    }
  }
}

}  // namespace adx

using namespace adx;

extern "C" {

// table: device array of n_tensors x {p, g, m, v, ema, n} (6 x 8 bytes each); block_tensor / block_chunk: device
// int32 arrays of n_blocks entries (chunk = 2048 elements).  step >= 1 is the AdamW step count.
// grad_scale multiplies every gradient as it is read (data-parallel training: the all-reduce leaves the SUM over the
// ranks in the buckets and the mean's 1 / world is applied here instead of by a pass of its own).
int adx_adamw_ema_step_scaled(const void* table, const int32_t* block_tensor, const int32_t* block_chunk, int32_t n_blocks,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                              float ema_decay, int32_t use_ema, int32_t sanitize, float grad_scale, adx_stream stream) {
  ADX_REQUIRE(table && block_tensor && block_chunk && n_blocks >= 1 && step >= 1, "adx_adamw_ema_step: bad argument");
  OptArgs a;
  a.tensors = (const OptTensor*)table; a.block_tensor = block_tensor; a.block_chunk = block_chunk;
  a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  a.ema_decay = ema_decay; a.use_ema = use_ema; a.sanitize = sanitize; a.grad_scale = grad_scale;
  adamw_ema_kernel<<<dim3(n_blocks), dim3(256), 0, (hipStream_t)stream>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int adx_adamw_ema_step(const void* table, const int32_t* block_tensor, const int32_t* block_chunk, int32_t n_blocks,
                       float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                       float ema_decay, int32_t use_ema, int32_t sanitize, adx_stream stream) {
  return adx_adamw_ema_step_scaled(table, block_tensor, block_chunk, n_blocks, lr, beta1, beta2, eps, weight_decay, step,
                                   ema_decay, use_ema, sanitize, 1.0f, stream);
}

int adx_optim_chunk(void) { return kOptChunk; }

}  // extern "C"
