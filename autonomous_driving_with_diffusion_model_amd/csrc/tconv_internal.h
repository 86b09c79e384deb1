// Internal pieces shared by the two temporal-convolution kernels (tconv.hip: exact fp32 MFMA, tconv_hs.hip:
// split-fp16 MFMA): the kernel argument block and the fused epilogue
//   K-partials -> (+bias) -> [GroupNorm -> Mish] -> (+time bias) -> (+residual) -> store
// (modeling/helpers.py:95-112, modeling/temporal.py:53-55).
#pragma once
#include "tconv.h"

// -DADX_TCONV_TRACE: every workgroup's thread 0 stamps the shader clock at its phase boundaries into a global table
// that tools/tconv_trace.py reads back (diagnostic builds only: csrc/build.sh -DADX_TCONV_TRACE with ADX_OUT=...).
#if defined(ADX_TCONV_TRACE) && defined(ADX_TCONV_TRACE_TU)   // the table lives in the one translation unit that stamps
namespace adx { __device__ unsigned long long g_tconv_trace[16 * 4096]; }
#define ADX_TSTAMP(i)                                                                                    \
  do {                                                                                                   \
    if (threadIdx.x == 0 && blockIdx.x < 4096) adx::g_tconv_trace[blockIdx.x * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define ADX_TSTAMP(i) do { } while (0)
#endif

namespace adx {

struct TConvArgs {
  adx_tconv_io io;
  int kind, taps, stride, pad;
  int c0, cin, cout, lin, lout, log2_lout;
  int groups, cg;
  float eps;
  int ncb, nkb;
  int bt, ct, log2_ct, pl, lp, rs, ck, ntiles, cin_pad;
  int dense;  // both inputs are plain [B][C][L] tensors with lin % 4 == 0: 16-byte staging loads
  int lin_valid, lout_valid;   // real lengths (<= lin, lout): adx_tconv_desc::lin_valid
};

// On entry P = smem holds NW partial tiles of TILE = bt * ct * lout floats each, laid out
// [wave][sample][channel][pos] (the order of the output tensor); all waves have passed a barrier after writing them.
// Every thread owns EPT elements e = tid + NT*k of the output tile; 64 consecutive elements (one wave's worth) always
// belong to the same (sample, GroupNorm group) pair because cg * lout is a multiple of 64 (checked on the host).
template <int NT, int NW, int TILE>
__device__ __forceinline__ void tconv_epilogue(const TConvArgs& a, float* smem, int tid, int lane, int wave, int nt,
                                               int b0) {
  constexpr int tile_elems = TILE;
  const int batch = a.io.batch;
  float* P = smem;
  const int n0 = nt * a.ct;
  // Every thread owns EPT elements e = tid + NT*k of the output tile; 64 consecutive elements (one
  // wave's worth) always belong to the same (sample, GroupNorm group) pair when n % 64 == 0.
  constexpr int EPT = (tile_elems + NT - 1) / NT;
  float v[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + NT * k;
    v[k] = 0.f;
    if (e < tile_elems) {
      float sum = P[e];
#pragma unroll
      for (int w = 1; w < NW; ++w) sum += P[e + w * tile_elems];  // fixed order: deterministic
      const int c = n0 + ((e >> a.log2_lout) & (a.ct - 1));
      if (a.io.bias != nullptr && c < a.cout) sum += a.io.bias[c];
      v[k] = sum;
      if (a.io.pre != nullptr) {
        const int b = b0 + (e >> (a.log2_lout + a.log2_ct));
        if (b < batch && c < a.cout)
          a.io.pre[((int64_t)b * a.cout + c) * a.lout + (e & (a.lout - 1))] = sum;
      }
    }
  }
  ADX_TSTAMP(5);
  const int n = a.cg << a.log2_lout;  // elements per (sample, group), padded positions included
  const bool ragged = a.lout_valid != a.lout;      // uniform: a real length that is not a power of two
  float* red = smem + NW * tile_elems; // 2 x (tile_elems / 64) partial sums, behind the K-partials
  constexpr int NCH = tile_elems / 64;
  const bool gn = a.groups > 0;
  // issue every global load of the epilogue now; they land while the statistics are reduced
  float gm[EPT], be[EPT], tb[EPT], rs_[EPT];
  int64_t yoff[EPT];
  bool live[EPT];
#pragma unroll
  for (int k = 0; k < EPT; ++k) {
    const int e = tid + NT * k;
    const int l = e & (a.lout - 1);
    const int c = n0 + ((e >> a.log2_lout) & (a.ct - 1));
    const int b = b0 + (e >> (a.log2_lout + a.log2_ct));
    live[k] = e < tile_elems && b < batch && c < a.cout && l < a.lout_valid;
    gm[k] = 1.f; be[k] = 0.f; tb[k] = 0.f; rs_[k] = 0.f; yoff[k] = 0;
    if (live[k]) {
      if (gn) { gm[k] = a.io.gamma[c]; be[k] = a.io.beta[c]; }
      if (a.io.tbias != nullptr) tb[k] = a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
      if (a.io.res != nullptr)
        rs_[k] = a.io.res[(int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)l * a.io.res_sl];
      yoff[k] = (int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)l * a.io.y_sl;
    }
  }
  if (gn) {
    // two-pass mean / variance: wave shuffle, then the pair's n/64 wave partials through LDS
    const int cpp = n >> 6;  // 64-element chunks per pair (n is a multiple of 64: checked on the host)
    const float inv_n = 1.0f / (float)(a.cg * a.lout_valid);       // statistics over the real positions only
    float mean[EPT], rstd[EPT];
    bool inr[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) inr[k] = !ragged || ((tid + NT * k) & (a.lout - 1)) < a.lout_valid;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const float s = wave_sum(inr[k] ? v[k] : 0.f);
      const int ch = wave + NW * k;
      if (lane == 0 && ch < NCH) red[ch] = s;
    }
    __syncthreads();
    ADX_TSTAMP(6);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int ch = min(wave + NW * k, NCH - 1);
      const int base = (ch / cpp) * cpp;
      float s = 0.f;
      for (int i = 0; i < cpp; ++i) s += red[base + i];
      mean[k] = s * inv_n;
      const float d = inr[k] ? v[k] - mean[k] : 0.f;
      const float q = wave_sum(d * d);
      if (lane == 0 && wave + NW * k < NCH) red[NCH + wave + NW * k] = q;
    }
    __syncthreads();
    ADX_TSTAMP(7);
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      const int ch = min(wave + NW * k, NCH - 1);
      const int base = (ch / cpp) * cpp;
      float q = 0.f;
      for (int i = 0; i < cpp; ++i) q += red[NCH + base + i];
      rstd[k] = 1.0f / sqrtf(q * inv_n + a.eps);
      if (a.io.stats != nullptr && lane == 0 && wave + NW * k < NCH && (ch % cpp) == 0) {
        const int e0 = (wave + NW * k) * 64;  // first element of this (sample, group) pair
        const int b = b0 + (e0 >> (a.log2_lout + a.log2_ct));
        const int g = (n0 + ((e0 >> a.log2_lout) & (a.ct - 1))) / a.cg;
        if (b < batch) {
          a.io.stats[((int64_t)b * a.groups + g) * 2] = mean[k];
          a.io.stats[((int64_t)b * a.groups + g) * 2 + 1] = rstd[k];
        }
      }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (live[k]) {
        float o = (v[k] - mean[k]) * (rstd[k] * gm[k]) + be[k];
        o = mish_f(o);
        if (a.io.tbias != nullptr) o += tb[k];
        if (a.io.res != nullptr) o += rs_[k];
        a.io.y[yoff[k]] = o;
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
      if (live[k]) {
        float o = v[k];
        if (a.io.tbias != nullptr) o += tb[k];
        if (a.io.res != nullptr) o += rs_[k];
        a.io.y[yoff[k]] = o;
      }
    }
  }
  ADX_TSTAMP(8);
}

}  // namespace adx
