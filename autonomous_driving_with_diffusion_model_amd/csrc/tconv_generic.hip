// Temporal convolution family, ANY shape: the fallback behind the two MFMA kernels.
//
// tconv_hs.hip (split-fp16 MFMA) and tconv.hip (exact-fp32 MFMA) tile a layer as 16- or 32-row x 16/32-channel
// fragments and keep every GroupNorm group a tile touches on chip, which works when the group width is a power of two
// and a (sample, group) holds a multiple of 64 elements.  The reference has no such rule: GroupNorm(8, C) for any C
// divisible by 8 (modeling/helpers.py:105-107: MODEL.DIM = 48 gives groups of 6, 12, 24 and 48 channels) and any
// horizon its down / up path survives (modeling/temporal.py:59-75: H = 8 reaches one position with 32-element groups).
// Those layers run here: plain fp32 FMAs, one workgroup per (sample, GroupNorm group) -- or per (sample, 16-channel
// slab) without GroupNorm -- so the statistics never leave the workgroup.  Same fused epilogue, same operand
// conventions (two inputs = skip concat, strides, lin_valid / lout_valid, pre / stats for the training forward).
// It is a correctness path: ~10x the time of the MFMA kernels on the shapes both can run.
#include "tconv_internal.h"

namespace adx {

struct GenArgs {
  adx_tconv_io io;
  int kind, taps, stride, pad;
  int c0, cin, cout, lin, lout, lin_valid, lout_valid;
  int groups, cg, ct, ntiles;      // ct = channels per workgroup (= cg with GroupNorm)
  float eps;
};

// weight image: fp32 [cout][taps][cin] (the reduction axis contiguous)
__global__ void tconv_generic_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int layout, int flip,
                                          int taps, int cin, int cout, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % cin;
  const int tap = (idx / cin) % taps;
  const int n = idx / ((size_t)cin * taps);
  const int ts = flip ? taps - 1 - tap : tap;
  packed[idx] = layout == 0 ? w[((size_t)n * cin + ci) * taps + ts] : w[((size_t)ci * cout + n) * taps + ts];
}

__device__ __forceinline__ float block_sum_256(float v, float* red, int tid) {   // every thread receives the total
  v = wave_sum(v);
  __syncthreads();                      // `red` may still be read from the previous call
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) tconv_generic_kernel(const GenArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* xs = smem;                                   // [cin][lin]
  float* ys = smem + (size_t)a.cin * a.lin;           // [ct][lout]
  float* red = ys + (size_t)a.ct * a.lout;            // [4]
  const int tid = threadIdx.x;
  const int b = blockIdx.x / a.ntiles, tile = blockIdx.x % a.ntiles;
  const int cbase = tile * a.ct;
  // ---- stage the sample's input (both sources), zero beyond lin_valid ----------------------------------------------
  for (int e = tid; e < a.cin * a.lin; e += 256) {
    const int ci = e / a.lin, ip = e - ci * a.lin;
    float v = 0.f;
    if (ip < a.lin_valid) {
      v = ci < a.c0 ? a.io.x0[(int64_t)b * a.io.x0_sb + (int64_t)ci * a.io.x0_sc + (int64_t)ip * a.io.x0_sl]
                    : a.io.x1[(int64_t)b * a.io.x1_sb + (int64_t)(ci - a.c0) * a.io.x1_sc + (int64_t)ip * a.io.x1_sl];
    }
    xs[e] = v;
  }
  __syncthreads();
  // ---- conv + bias: thread -> (channel, position) pairs of this workgroup's slab ----------------------------------------
  const int nel = a.ct * a.lout;
  for (int e = tid; e < nel; e += 256) {
    const int cl = e / a.lout, l = e - cl * a.lout;
    const int c = cbase + cl;
    float acc = 0.f;
    if (c < a.cout && l < a.lout_valid) {
      const float* wr = a.io.packed_w + (size_t)c * a.taps * a.cin;
      for (int tap = 0; tap < a.taps; ++tap) {
        int ip;
        bool ok = true;
        if (a.kind == 0) {
          ip = l * a.stride + tap - a.pad;
        } else {              // ConvTranspose1d, stride 2: o = 2 i - pad + tap
          const int vt = l + a.pad - tap;
          ok = (vt & 1) == 0;
          ip = vt >> 1;
        }
        if (!ok || ip < 0 || ip >= a.lin_valid) continue;
        const float* wt = wr + (size_t)tap * a.cin;
        float s0 = 0.f, s1 = 0.f;
        int ci = 0;
        for (; ci + 1 < a.cin; ci += 2) {
          s0 = fmaf(wt[ci], xs[ci * a.lin + ip], s0);
          s1 = fmaf(wt[ci + 1], xs[(ci + 1) * a.lin + ip], s1);
        }
        if (ci < a.cin) s0 = fmaf(wt[ci], xs[ci * a.lin + ip], s0);
        acc += s0 + s1;
      }
      if (a.io.bias != nullptr) acc += a.io.bias[c];
      if (a.io.pre != nullptr) a.io.pre[((int64_t)b * a.cout + c) * a.lout + l] = acc;
    }
    ys[e] = acc;
  }
  // ---- GroupNorm over the (sample, group) = this workgroup's real elements, two passes ----------------------------------
  float mean = 0.f, rstd = 1.f;
  if (a.groups > 0) {
    const float inv_n = 1.0f / (float)(a.cg * a.lout_valid);
    float s = 0.f;
    __syncthreads();
    for (int e = tid; e < nel; e += 256)
      if (e % a.lout < a.lout_valid) s += ys[e];
    mean = block_sum_256(s, red, tid) * inv_n;
    float q = 0.f;
    for (int e = tid; e < nel; e += 256)
      if (e % a.lout < a.lout_valid) { const float d = ys[e] - mean; q += d * d; }
    rstd = 1.0f / sqrtf(block_sum_256(q, red, tid) * inv_n + a.eps);
    if (a.io.stats != nullptr && tid == 0) {
      a.io.stats[((int64_t)b * a.groups + tile) * 2] = mean;
      a.io.stats[((int64_t)b * a.groups + tile) * 2 + 1] = rstd;
    }
  }
  // ---- affine + Mish + time bias + residual + store (each thread re-reads the elements it wrote) ----------------------
  for (int e = tid; e < nel; e += 256) {
    const int cl = e / a.lout, l = e - cl * a.lout;
    const int c = cbase + cl;
    if (c >= a.cout || l >= a.lout_valid) continue;
    float o = ys[e];
    if (a.groups > 0) o = mish_f((o - mean) * (rstd * a.io.gamma[c]) + a.io.beta[c]);
    if (a.io.tbias != nullptr) o += a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
    if (a.io.res != nullptr) o += a.io.res[(int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)l * a.io.res_sl];
    a.io.y[(int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)l * a.io.y_sl] = o;
  }
}

constexpr size_t kMaxGenericLds = 128 * 1024;

static void generic_geometry(const adx_tconv_desc* d, int* ct, int* ntiles, size_t* lds) {
  const int cin = d->c0 + d->c1;
  *ct = d->groups > 0 ? d->cout / d->groups : 16;
  *ntiles = d->groups > 0 ? d->groups : ceil_div(d->cout, 16);
  *lds = ((size_t)cin * d->lin + (size_t)*ct * d->lout + 4) * sizeof(float);
}

bool tconv_generic_supported(const adx_tconv_desc* d) {
  if (d->groups > 0 && d->cout % d->groups != 0) return false;
  int ct, nt;
  size_t lds;
  generic_geometry(d, &ct, &nt, &lds);
  return lds <= kMaxGenericLds;
}

size_t tconv_generic_packed_floats(const adx_tconv_desc* d) { return (size_t)d->cout * d->taps * (d->c0 + d->c1); }

int tconv_generic_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s) {
  const size_t total = tconv_generic_packed_floats(d);
  tconv_generic_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
      w, packed, d->kind == 1 ? 1 - d->w_layout : d->w_layout, d->w_flip, d->taps, d->c0 + d->c1, d->cout, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int tconv_generic_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s) {
  GenArgs a;
  a.io = *io;
  a.kind = d->kind; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
  a.c0 = d->c0; a.cin = d->c0 + d->c1; a.cout = d->cout; a.lin = d->lin; a.lout = d->lout;
  a.lin_valid = d->lin_valid > 0 ? d->lin_valid : d->lin;
  a.lout_valid = d->lout_valid > 0 ? d->lout_valid : d->lout;
  a.groups = d->groups; a.cg = d->groups > 0 ? d->cout / d->groups : 1; a.eps = d->eps;
  size_t lds;
  generic_geometry(d, &a.ct, &a.ntiles, &lds);
  ADX_REQUIRE(lds <= kMaxGenericLds, "tconv (general-shape kernel): a sample's input of %d x %d floats does not fit the LDS",
              a.cin, a.lin);
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_generic_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxGenericLds));
    once.commit();
  }
  tconv_generic_kernel<<<dim3(io->batch * a.ntiles), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
