// Scheduler step arithmetic: one fused elementwise kernel per denoising step.
//
//   S1 GuidanceDDIMScheduler.step    scheduler/guidance_ddim_scheduler.py:60-173
//   S2 GuidanceDDPMScheduler.step    scheduler/guidance_ddpm_scheduler.py:59-178 (== stock diffusers DDPM, train.py:87)
//   S3 InpaintingDDIMScheduler.step  scheduler/inpainting_ddim_scheduler.py:10-153
//   S4 InpaintingDDPMScheduler.step  scheduler/inpainting_ddpm_scheduler.py:10-146
//   C1 classifier-free combine       interact.py:142-144
//   C2 trajs[:, 0, :3] = 0           interact.py:164, train.py:88
//   add_noise                        diffusers DDPMScheduler.add_noise, train.py:234-235
//
// The reference evaluates these as ~10 separate torch ops on [B,H,7] tensors with 0-dim CPU
// coefficients (a host sync per step).  Here the host passes the fp32-rounded scalars by value
// and the kernel repeats the reference's elementwise operations in the same order with
// contraction disabled, so results are bit-identical to the torch CPU path.
#include "adx_common.h"

namespace adx {

struct StepArgs {
  adx_step_coef c;
  const float* mo;
  const float* x;
  const float* z;
  const float* tgt;
  const float* mask;
  float* prev;
  float* x0;
  int total, horizon, dim;
};

__device__ __forceinline__ float clamp_nan(float v, float lo, float hi) {
  // torch.clamp propagates NaN
  return v < lo ? lo : (v > hi ? hi : v);
}

template <bool DDPM>
__global__ void __launch_bounds__(256) step_kernel(const StepArgs a) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= a.total) return;
  const adx_step_coef& c = a.c;
  float m;
  if (c.cfg_combine) {
    const float cnd = a.mo[e], unc = a.mo[e + a.total];
    const float d = cnd - unc;
    const float sd = c.free_scale * d;
    m = unc + sd;
  } else {
    m = a.mo[e];
  }
  const float xs = a.x[e];
  const float sa = c.sqrt_alpha_t, sb = c.sqrt_beta_t;
  float x0, eps;
  if (c.prediction_type == ADX_PRED_EPSILON) {
    const float p = sb * m;
    x0 = (xs - p) / sa;
    eps = m;
  } else if (c.prediction_type == ADX_PRED_SAMPLE) {
    x0 = m;
    const float p = sa * x0;
    eps = (xs - p) / sb;
  } else {
    const float p = sa * xs, q = sb * m;
    x0 = p - q;
    const float p2 = sa * m, q2 = sb * xs;
    eps = p2 + q2;
  }
  if (c.clip) x0 = clamp_nan(x0, -c.clip_range, c.clip_range);
  const float zn = (a.z != nullptr) ? a.z[e] : 0.f;
  float prev;
  if (!DDPM) {
    if (c.use_clipped_model_output) {
      const float p = sa * x0;
      eps = (xs - p) / sb;
    }
    const float dir = c.c_dir * eps;
    const float p = c.c_x0 * x0;
    prev = p + dir;
    if (c.inpaint) {
      prev = prev + c.c_const;
      if (a.tgt != nullptr && a.mask != nullptr) {
        const float k0 = c.c_known * a.tgt[e];
        const float k1 = c.known_noise ? c.c_known_noise * zn : 0.f;
        const float known = k0 + k1;
        const float mk = a.mask[e];
        const float u = mk * known, v = (1.0f - mk) * prev;
        prev = u + v;
      }
    }
    if (c.add_noise) {
      const float nz = c.c_noise * zn;
      prev = prev + nz;
    }
  } else {
    const float p = c.c_x0 * x0, q = c.c_x * xs;
    prev = p + q;
    if (c.add_noise) {
      const float nz = c.c_noise * zn;
      prev = prev + nz;
    }
    if (c.inpaint && a.tgt != nullptr && a.mask != nullptr) {
      const float k0 = c.c_known * a.tgt[e];
      const float k1 = c.known_noise ? c.c_known_noise * zn : 0.f;
      const float known = k0 + k1;
      const float mk = a.mask[e];
      const float u = mk * known, v = (1.0f - mk) * prev;
      prev = u + v;
    }
  }
  if (c.zero_first) {
    const int d = e % a.dim;
    const int h = (e / a.dim) % a.horizon;
    if (h == 0 && d < 3) prev = 0.f;
  }
  a.prev[e] = prev;
  if (a.x0 != nullptr) a.x0[e] = x0;
}

template <bool DDPM>
static int step_launch(const adx_step_coef* c, const float* mo, const float* x, const float* z, const float* tgt,
                       const float* mask, float* prev, float* x0, int batch, int horizon, int dim, hipStream_t s) {
  ADX_REQUIRE(c && mo && x && prev, "scheduler step: null tensor");
  ADX_REQUIRE(batch >= 1 && horizon >= 1 && dim >= 1, "scheduler step: empty shape");
  ADX_REQUIRE(c->prediction_type >= 0 && c->prediction_type <= 2,
              "prediction_type given as %d must be one of `epsilon`, `sample`, or `v_prediction`", c->prediction_type);
  ADX_REQUIRE(!(c->add_noise || (c->inpaint && c->known_noise && tgt && mask)) || z != nullptr,
              "scheduler step: noise tensor required");
  StepArgs a;
  a.c = *c;
  a.mo = mo; a.x = x; a.z = z; a.tgt = tgt; a.mask = mask; a.prev = prev; a.x0 = x0;
  a.total = batch * horizon * dim; a.horizon = horizon; a.dim = dim;
  step_kernel<DDPM><<<dim3(ceil_div(a.total, 256)), dim3(256), 0, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int ddim_step(const adx_step_coef* c, const float* mo, const float* x, const float* z, const float* tgt,
              const float* mask, float* prev, float* x0, int b, int h, int d, hipStream_t s) {
  return step_launch<false>(c, mo, x, z, tgt, mask, prev, x0, b, h, d, s);
}
int ddpm_step(const adx_step_coef* c, const float* mo, const float* x, const float* z, const float* tgt,
              const float* mask, float* prev, float* x0, int b, int h, int d, hipStream_t s) {
  return step_launch<true>(c, mo, x, z, tgt, mask, prev, x0, b, h, d, s);
}

__global__ void __launch_bounds__(256) add_noise_kernel(const float* __restrict__ x, const float* __restrict__ n,
                                                         const int64_t* __restrict__ t, const float* __restrict__ sa,
                                                         const float* __restrict__ sb, float* __restrict__ out,
                                                         int total, int per, int horizon, int dim, int zero_first) {
#pragma clang fp contract(off)
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b = e / per;
  const int64_t tb = t[b];
  const float p = sa[tb] * x[e], q = sb[tb] * n[e];
  float v = p + q;
  if (zero_first) {
    const int d = e % dim, h = (e / dim) % horizon;
    if (h == 0 && d < 3) v = 0.f;
  }
  out[e] = v;
}

int add_noise(const float* x, const float* n, const int64_t* t, const float* sa, const float* sb, int n_train,
              float* out, int batch, int horizon, int dim, int zero_first, hipStream_t s) {
  ADX_REQUIRE(x && n && t && sa && sb && out, "add_noise: null tensor");
  ADX_REQUIRE(batch >= 1 && horizon >= 1 && dim >= 1 && n_train >= 1, "add_noise: empty shape");
  const int total = batch * horizon * dim;
  add_noise_kernel<<<dim3(ceil_div(total, 256)), dim3(256), 0, s>>>(x, n, t, sa, sb, out, total, horizon * dim,
                                                                  horizon, dim, zero_first);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// Camera front-end of the agents (interact.py:73-78, e2e_driving/diffusion_agent.py:96-101,294):
// torchvision ToTensor + Normalize(mean, std) on a uint8 HWC frame -> fp32 NCHW, one pass.
__global__ void __launch_bounds__(256) image_normalize_kernel(const uint8_t* __restrict__ src, float* __restrict__ dst,
                                                               int n, int h, int w, float m0, float m1, float m2,
                                                               float s0, float s1, float s2) {
#pragma clang fp contract(off)
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;   // one output pixel (all 3 channels)
  const size_t hw = (size_t)h * w;
  if (i >= (size_t)n * hw) return;
  const size_t img = i / hw, pix = i - img * hw;
  const uint8_t* p = src + i * 3;
  const float mean[3] = {m0, m1, m2}, stdv[3] = {s0, s1, s2};
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float v = (float)p[c] / 255.0f;              // ToTensor
    dst[(img * 3 + c) * hw + pix] = (v - mean[c]) / stdv[c];
  }
}

int image_normalize(const uint8_t* src, float* dst, int n, int h, int w, const float* mean, const float* stdv,
                    hipStream_t s) {
  ADX_REQUIRE(src && dst && mean && stdv && n >= 1 && h >= 1 && w >= 1, "image_normalize: bad argument");
  const size_t total = (size_t)n * h * w;
  image_normalize_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(src, dst, n, h, w, mean[0], mean[1],
                                                                                  mean[2], stdv[0], stdv[1], stdv[2]);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
