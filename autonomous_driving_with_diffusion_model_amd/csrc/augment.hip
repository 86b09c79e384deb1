// GPU stand-in for the reference's per-sample image augmentation (dataset/augment.py:10-77, applied in
// dataset/carla_dataset.py:24-31 through imgaug, which this image does not have): the same seven operators with the
// same iteration-dependent strengths -- GaussianBlur, AdditiveGaussianNoise, CoarseDropout, Dropout, Add, Multiply,
// LinearContrast, each applied with probability `frequency_factor`, in a random order, per channel with probability
// `color_factor` -- on a whole batch of uint8 HWC frames that already sits in HBM.  The HOST draws the plan (which
// operators, their order and parameters: dataset/augment.py of this package); the device draws the per-pixel randomness
// from a counter-based hash, so a plan + seed reproduces the same image on any launch geometry, and oracle/augment.py
// restates the arithmetic in numpy for the parity test.  Every operator consumes and produces uint8 (round half to
// even, saturate), like imgaug's augmenters do on uint8 images.
//
// A plan row is 8 floats: code, p0, p1, p2, p3, per_channel, 0, 0 with
//   1 blur      p0 = sigma                         (5x5 separable window, radius 2: sigma <= 0.5)
//   2 noise     p0 = scale (standard deviation in grey levels)
//   3 coarse    p0 = drop probability, p1 = grid rows, p2 = grid columns
//   4 dropout   p0 = drop probability
//   5 add       p0..p2 = value per channel
//   6 multiply  p0..p2 = factor per channel
//   7 contrast  p0..p2 = alpha per channel: 128 + alpha (v - 128)
//   0 nothing
#include "adx_common.h"

namespace adx {

constexpr int kAugSlots = 7;

__device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
__device__ __forceinline__ float u01(uint64_t key) { return (float)(splitmix64(key) >> 40) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint64_t aug_key(uint64_t seed, int slot, int channel, uint64_t index) {
  return seed ^ ((uint64_t)slot << 56) ^ ((uint64_t)channel << 52) ^ index;
}
__device__ __forceinline__ float to_u8(float v) {       // round half to even, saturate
  return fminf(fmaxf(rintf(v), 0.f), 255.f);
}

// pointwise operators of the slots [first, last) of every image, in place
__global__ void __launch_bounds__(256) augment_pointwise_kernel(uint8_t* __restrict__ img, const float* __restrict__ plan,
                                                                 const uint64_t* __restrict__ seeds,
                                                                 const int* __restrict__ range, int phase, int n, int h,
                                                                 int w) {
#pragma clang fp contract(off)
  const size_t hw = (size_t)h * w;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * hw) return;
  const int im = (int)(i / hw);
  const uint64_t pix = i - (size_t)im * hw;
  const int y = (int)(pix / w), x = (int)(pix - (uint64_t)y * w);
  const int first = range[im * 4 + phase * 2], last = range[im * 4 + phase * 2 + 1];
  if (first >= last) return;
  uint8_t* p = img + i * 3;
  float v[3] = {(float)p[0], (float)p[1], (float)p[2]};
  const uint64_t seed = seeds[im];
  for (int s = first; s < last; ++s) {
    const float* r = plan + ((size_t)im * kAugSlots + s) * 8;
    const int code = (int)r[0];
    const bool pc = r[5] != 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int kc = pc ? c : 0;              // one draw shared by the three channels unless per_channel
      float o = v[c];
      if (code == 2) {
        const float u1 = u01(aug_key(seed, s, kc, 2 * pix)), u2 = u01(aug_key(seed, s, kc, 2 * pix + 1));
        const float z = sqrtf(-2.0f * logf(u1 + 2.98023224e-8f)) * cosf(6.28318530718f * u2);
        o = o + r[1] * z;
      } else if (code == 3) {
        const int gy = min((int)((float)y * r[2] / (float)h), (int)r[2] - 1), gx = min((int)((float)x * r[3] / (float)w), (int)r[3] - 1);
        if (u01(aug_key(seed, s, kc, (uint64_t)gy * 65536u + (uint64_t)gx)) < r[1]) o = 0.f;
      } else if (code == 4) {
        if (u01(aug_key(seed, s, kc, pix)) < r[1]) o = 0.f;
      } else if (code == 5) {
        o = o + r[1 + c];
      } else if (code == 6) {
        o = o * r[1 + c];
      } else if (code == 7) {
        o = 128.0f + r[1 + c] * (o - 128.0f);
      }
      v[c] = to_u8(o);
    }
  }
  p[0] = (uint8_t)v[0]; p[1] = (uint8_t)v[1]; p[2] = (uint8_t)v[2];
}

// Gaussian blur of the images whose plan holds an active blur (blur_sigma[im] > 0), src -> dst; the others are copied.
// 5x5 window, weights exp(-d^2 / (2 sigma^2)) normalised per axis, borders by reflection (imgaug / cv2 default: 101).
__global__ void __launch_bounds__(256) augment_blur_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                            const float* __restrict__ sigma, int n, int h, int w) {
#pragma clang fp contract(off)
  const size_t hw = (size_t)h * w;
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)n * hw) return;
  const int im = (int)(i / hw);
  const size_t pix = i - (size_t)im * hw;
  const float sg = sigma[im];
  const uint8_t* s0 = src + (size_t)im * hw * 3;
  uint8_t* d = dst + i * 3;
  if (!(sg > 0.f)) {
    d[0] = s0[pix * 3]; d[1] = s0[pix * 3 + 1]; d[2] = s0[pix * 3 + 2];
    return;
  }
  const int y = (int)(pix / w), x = (int)(pix - (size_t)y * w);
  float k[5], ks = 0.f;
#pragma unroll
  for (int t = 0; t < 5; ++t) { k[t] = expf(-(float)((t - 2) * (t - 2)) / (2.0f * sg * sg)); ks += k[t]; }
#pragma unroll
  for (int t = 0; t < 5; ++t) k[t] = k[t] / ks;
  auto refl = [](int q, int m) { q = q < 0 ? -q : q; return q >= m ? 2 * m - 2 - q : q; };
  float acc[3] = {0.f, 0.f, 0.f};
  for (int dy = 0; dy < 5; ++dy) {
    const int yy = refl(y + dy - 2, h);
    float row[3] = {0.f, 0.f, 0.f};
    for (int dx = 0; dx < 5; ++dx) {
      const int xx = refl(x + dx - 2, w);
      const uint8_t* q = s0 + ((size_t)yy * w + xx) * 3;
#pragma unroll
      for (int c = 0; c < 3; ++c) row[c] = row[c] + k[dx] * (float)q[c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) acc[c] = acc[c] + k[dy] * row[c];
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) d[c] = (uint8_t)to_u8(acc[c]);
}

}  // namespace adx

extern "C" int adx_image_augment(uint8_t* frames_hwc, uint8_t* scratch, int32_t n, int32_t h, int32_t w,
                                 const float* plan /* [n][7][8] device */, const uint64_t* seeds /* [n] device */,
                                 const int32_t* ranges /* [n][4] device */, const float* blur_sigma /* [n] device */,
                                 int32_t any_blur, adx_stream stream) {
  using namespace adx;
  ADX_REQUIRE(frames_hwc && plan && seeds && ranges && blur_sigma && n >= 1 && h >= 3 && w >= 3, "adx_image_augment: bad argument");
  ADX_REQUIRE(!any_blur || scratch != nullptr, "adx_image_augment: a blur needs the scratch image");
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)n * h * w;
  const dim3 grid((unsigned)((total + 255) / 256)), blk(256);
  augment_pointwise_kernel<<<grid, blk, 0, s>>>(frames_hwc, plan, seeds, ranges, 0, n, h, w);
  if (any_blur) {
    augment_blur_kernel<<<grid, blk, 0, s>>>(frames_hwc, scratch, blur_sigma, n, h, w);
    ADX_CHECK_HIP(hipMemcpyAsync(frames_hwc, scratch, total * 3, hipMemcpyDeviceToDevice, s));
  }
  augment_pointwise_kernel<<<grid, blk, 0, s>>>(frames_hwc, plan, seeds, ranges, 1, n, h, w);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}
