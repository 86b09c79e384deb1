// Backward kernels of the temporal stack (training step T1, train.py:242-251).
//
//   gn_mish_bwd   gradient through [+time bias] -> Mish -> GroupNorm(8) of one Conv1dBlock:
//                 dy -> dc (gradient w.r.t. conv+bias), d gamma, d beta, d conv-bias, d time-bias
//   tconv_wgrad   weight gradient of every temporal conv: dW[co][ci][tap] = sum_{b,l} dc[b][co][l] x[b][ci][pos(l,tap)]
//                 as an MFMA GEMM with K = (sample, position), split over workgroups and atomically reduced
//   bias_grad     db[c] = sum_{b,l} dc[b][c][l] for the convs without GroupNorm
// The data gradient needs no kernel of its own: it is adx_tconv_forward with the same weight read
// through w_layout / w_flip (include/adx.h).
#include "tconv.h"

namespace adx {

// d/dx [x tanh(softplus(x))], same single-exp formulation as mish_f
__device__ __forceinline__ float mish_grad(float x) {
  if (x > 20.f) return 1.f;
  const float e = expf(x);
  const float n = e * (e + 2.f);
  const float t = n / (n + 2.f);             // tanh(softplus(x))
  const float sg = e / (1.f + e);            // sigmoid(x) = d softplus / dx
  return t + x * (1.f - t * t) * sg;
}

struct GnBwdArgs {
  const float* dy; int64_t dy_sb, dy_sc, dy_sl;
  const float* pre;      // [B][C][L]
  const float* stats;    // [B][G][2]
  const float* gamma; const float* beta;
  float* dc;             // [B][C][L]
  float* dgamma; float* dbeta; float* dbias;   // [C], accumulated atomically (caller zeroes)
  float* dtb; int64_t dtb_stride;              // [B][...] written (may be null)
  int B, C, L, log2_L, G, cg;
  int Lv;                // real length (<= L): positions >= Lv do not exist (adx_tconv_desc::lout_valid)
};

// sum over aligned groups of `width` lanes (width = power of two <= 64); every lane gets its group's sum
__device__ __forceinline__ float seg_sum(float v, int width) {
  for (int off = 1; off < width; off <<= 1) v += __shfl_xor(v, off, 64);
  return v;
}

constexpr int kGnEmax = 8;  // cg * L <= 512 elements per (sample, group)

__global__ void __launch_bounds__(256) gn_mish_bwd_kernel(const GnBwdArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int pair = blockIdx.x * 4 + wave;
  if (pair >= a.B * a.G) return;
  const int b = pair / a.G, g = pair - b * a.G;
  const int n = a.cg << a.log2_L;
  const float mean = a.stats[(int64_t)pair * 2], rstd = a.stats[(int64_t)pair * 2 + 1];
  const float inv_n = 1.0f / (float)(a.cg * a.Lv);
  float dz[kGnEmax], xh[kGnEmax], gm[kGnEmax], dyv[kGnEmax];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < kGnEmax; ++k) {
    const int e = lane + 64 * k;
    dz[k] = 0.f; xh[k] = 0.f; gm[k] = 0.f; dyv[k] = 0.f;
    if (e < n && (e & (a.L - 1)) < a.Lv) {
      const int c = g * a.cg + (e >> a.log2_L), l = e & (a.L - 1);
      const float p = a.pre[((int64_t)b * a.C + c) * a.L + l];
      const float d = a.dy[(int64_t)b * a.dy_sb + (int64_t)c * a.dy_sc + (int64_t)l * a.dy_sl];
      gm[k] = a.gamma[c];
      xh[k] = (p - mean) * rstd;
      const float u = xh[k] * gm[k] + a.beta[c];
      dyv[k] = d;
      dz[k] = d * mish_grad(u);
      s1 += dz[k] * gm[k];
      s2 += dz[k] * gm[k] * xh[k];
    }
  }
  const float m1 = wave_sum(s1) * inv_n, m2 = wave_sum(s2) * inv_n;
  const int seg = a.L < 64 ? a.L : 64;   // lanes that share a channel inside one 64-element chunk
#pragma unroll
  for (int k = 0; k < kGnEmax; ++k) {
    const int e = lane + 64 * k;
    if (64 * k >= n) break;              // wave-uniform
    const int c = g * a.cg + (e >> a.log2_L), l = e & (a.L - 1);
    const bool ok = e < n && l < a.Lv;
    const float dcv = ok ? rstd * (dz[k] * gm[k] - m1 - xh[k] * m2) : 0.f;
    if (e < n) a.dc[((int64_t)b * a.C + c) * a.L + l] = dcv;      // zero at positions that do not exist: the sums downstream stay unmasked
    // per-channel partial sums over the positions held by this chunk
    const float sg = seg_sum(ok ? dz[k] * xh[k] : 0.f, seg);
    const float sb = seg_sum(ok ? dz[k] : 0.f, seg);
    const float sc = seg_sum(dcv, seg);
    const float st = seg_sum(ok ? dyv[k] : 0.f, seg);
    if (e < n && (lane & (seg - 1)) == 0) {
      atomicAdd(a.dgamma + c, sg);
      atomicAdd(a.dbeta + c, sb);
      if (a.dbias != nullptr) atomicAdd(a.dbias + c, sc);
      if (a.dtb != nullptr) {
        if (a.L <= 64) a.dtb[(int64_t)b * a.dtb_stride + c] = st;      // the chunk covers every position of c
        else atomicAdd(a.dtb + (int64_t)b * a.dtb_stride + c, st);
      }
    }
  }
}

int gn_mish_backward(const GnBwdArgs& a, hipStream_t s) {
  ADX_REQUIRE(a.dy && a.pre && a.stats && a.gamma && a.beta && a.dc && a.dgamma && a.dbeta, "gn_mish_backward: null tensor");
  ADX_REQUIRE(a.cg * a.L <= 64 * kGnEmax && (a.cg * a.L) % 64 == 0, "gn_mish_backward: group of %d elements unsupported",
              a.cg * a.L);
  gn_mish_bwd_kernel<<<dim3(ceil_div(a.B * a.G, 4)), dim3(256), 0, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int gn_mish_backward_raw(const float* dy, int64_t sb, int64_t sc, int64_t sl, const float* pre, const float* stats,
                         const float* gamma, const float* beta, float* dc, float* dgamma, float* dbeta, float* dbias,
                         float* dtb, int64_t dtb_stride, int B, int C, int L, int groups, hipStream_t s, int L_valid) {
  GnBwdArgs a;
  a.dy = dy; a.dy_sb = sb; a.dy_sc = sc; a.dy_sl = sl;
  a.pre = pre; a.stats = stats; a.gamma = gamma; a.beta = beta; a.dc = dc;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.dtb = dtb; a.dtb_stride = dtb_stride;
  a.B = B; a.C = C; a.L = L; a.G = groups; a.cg = C / groups;
  a.Lv = L_valid > 0 ? L_valid : L;
  a.log2_L = 0;
  while ((1 << a.log2_L) < L) ++a.log2_L;
  ADX_REQUIRE((1 << a.log2_L) == L && a.Lv <= L, "gn_mish_backward: L must be a power of two (real length in L_valid)");
  return gn_mish_backward(a, s);
}

// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bias_grad_kernel(const float* __restrict__ dc, int64_t sb, int64_t sc, int64_t sl,
                                                         float* __restrict__ db, int B, int C, int L, int Lv) {
  // one workgroup per channel
  const int c = blockIdx.x, tid = threadIdx.x;
  float s = 0.f;
  for (int i = tid; i < B * L; i += 256) {
    const int b = i / L, l = i - b * L;
    if (l < Lv) s += dc[(int64_t)b * sb + (int64_t)c * sc + (int64_t)l * sl];
  }
  __shared__ float red[4];
  s = wave_sum(s);
  if ((tid & 63) == 0) red[tid >> 6] = s;
  __syncthreads();
  if (tid == 0) db[c] = (red[0] + red[1]) + (red[2] + red[3]);
}

int bias_grad(const float* dc, int64_t sb, int64_t sc, int64_t sl, float* db, int B, int C, int L, hipStream_t s, int L_valid) {
  ADX_REQUIRE(dc && db && B >= 1 && C >= 1 && L >= 1, "bias_grad: bad argument");
  bias_grad_kernel<<<dim3(C), dim3(256), 0, s>>>(dc, sb, sc, sl, db, B, C, L, L_valid > 0 ? L_valid : L);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

__global__ void __launch_bounds__(256) add_strided_kernel(float* __restrict__ dst, const float* __restrict__ src,
                                                           int64_t sb, int64_t sc, int64_t sl, int B, int C, int L, int Lv) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= B * C * L) return;
  const int l = idx % L, c = (idx / L) % C, b = idx / (L * C);
  if (l < Lv) dst[idx] += src[(int64_t)b * sb + (int64_t)c * sc + (int64_t)l * sl];
}

// dst dense [B][C][L]; positions >= L_valid (0 = L) of src are not read (a strided view may not have them)
int add_strided(float* dst, const float* src, int64_t sb, int64_t sc, int64_t sl, int B, int C, int L, hipStream_t s, int L_valid) {
  add_strided_kernel<<<dim3(ceil_div(B * C * L, 256)), dim3(256), 0, s>>>(dst, src, sb, sc, sl, B, C, L, L_valid > 0 ? L_valid : L);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient.  GEMM view: rows i = input channel (16 per workgroup), cols j = output channel
// (16 * NFO per workgroup), one accumulator set per tap, K = flattened (sample, output position).
//   A (16 x 4): x[b][ci][pos(l, tap)] from the same zero-padded LDS tile the forward conv stages;
//   B (4 x 16): dc[b][co][l] straight from global memory (contiguous 4-float runs along l).
// Each workgroup reduces `nb` samples; `nsplit` workgroups share one weight tile and combine with
// float atomics (the caller zeroes dW), so even the 64-channel layers fill the chip.
struct WgradArgs {
  const float* x0; int64_t x0_sb, x0_sc, x0_sl;
  const float* x1; int64_t x1_sb, x1_sc, x1_sl;
  const float* dc;        // [B][cout][lout] dense
  float* dw;              // kind 0: [cout][cin][taps]; kind 1 (conv-transpose weight): [cin][cout][taps] with roles swapped by the host
  int c0, cin, cout, lin, lout, log2_lout, taps, stride, pad;
  int lin_valid, lout_valid;    // real lengths: x positions >= lin_valid read as zero, dc positions >= lout_valid likewise
  int batch, nb, nsplit, n_ci_tiles, n_co_tiles;
  int sbt, lp, pl, rs;    // samples per staged super-tile, per-sample pitch, left pad, LDS row stride
  int64_t dw_so, dw_si;   // strides of dw for (co, ci); tap stride is 1
};

template <int NFO, int TAPS>
__global__ void __launch_bounds__(256) tconv_wgrad_kernel(const WgradArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int ci_t = bid % a.n_ci_tiles; bid /= a.n_ci_tiles;
  const int co_t = bid % a.n_co_tiles; bid /= a.n_co_tiles;
  const int split = bid;
  const int ci0 = ci_t * 16, co0 = co_t * 16 * NFO;
  const int bs = split * a.nb, be = min(bs + a.nb, a.batch);
  const int i16 = lane & 15, kq = lane >> 4;

  f32x4 acc[TAPS][NFO];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int f = 0; f < NFO; ++f) acc[t][f] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int KSW = 8;   // K-steps per wave per staged super-tile (host keeps sbt * lout <= 128 rows)
  for (int bb = bs; bb < be; bb += a.sbt) {
    const int nsamp = min(a.sbt, be - bb);
    const int mrows = nsamp << a.log2_lout;
    const int ksteps = (mrows + 3) >> 2;              // 4 flattened (sample, position) rows per MFMA
    // all dc operands of this wave for the super-tile: issued first, they land while x is staged
    float bv[KSW][NFO];
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const int m = 4 * (wave + 4 * j) + kq;
      const bool mok = m < mrows;
      const int sl_ = mok ? m >> a.log2_lout : 0, l = m & (a.lout - 1);
#pragma unroll
      for (int f = 0; f < NFO; ++f) {
        const int co = co0 + 16 * f + i16;
        bv[j][f] = (mok && co < a.cout && l < a.lout_valid) ? a.dc[((int64_t)(bb + sl_) * a.cout + co) * a.lout + l] : 0.f;
      }
    }
    if (bb > bs) __syncthreads();
    // stage x[bb .. bb+nsamp)[ci0 .. ci0+16) zero padded: [16][rs], sample s at column s*lp + pl
#pragma unroll 4
    for (int it = tid; it < 16 * a.rs; it += 256) {
      const int cl = it / a.rs, col = it - cl * a.rs;
      const int sl_ = col / a.lp, ip = col - sl_ * a.lp - a.pl;
      const int ci = ci0 + cl, b = bb + sl_;
      float v = 0.f;
      if (sl_ < nsamp && ip >= 0 && ip < a.lin_valid && ci < a.cin) {
        v = ci < a.c0 ? a.x0[(int64_t)b * a.x0_sb + (int64_t)ci * a.x0_sc + (int64_t)ip * a.x0_sl]
                      : a.x1[(int64_t)b * a.x1_sb + (int64_t)(ci - a.c0) * a.x1_sc + (int64_t)ip * a.x1_sl];
      }
      smem[it] = v;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < KSW; ++j) {
      const int ks = wave + 4 * j;
      if (ks >= ksteps) break;                          // wave-uniform
      const int m = 4 * ks + kq;                        // this lane's K row
      const bool mok = m < mrows;                       // ragged tail contributes zeros
      const int sl_ = mok ? m >> a.log2_lout : 0, l = m & (a.lout - 1);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int ip = l * a.stride + t - a.pad;        // halo columns are zero in LDS
        const float av = mok ? smem[i16 * a.rs + sl_ * a.lp + a.pl + ip] : 0.f;
#pragma unroll
        for (int f = 0; f < NFO; ++f) acc[t][f] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv[j][f], acc[t][f], 0, 0, 0);
      }
    }
  }
  // C layout: col = lane & 15 = output channel, row = 4*(lane>>4) + q = input channel.
  // The 4 waves are summed through LDS into the layout of dW ([co][ci][tap]: 16*TAPS contiguous floats per
  // output channel), so that every atomic wave-instruction adds to consecutive addresses.
  __syncthreads();
  constexpr int ROWF = 16 * TAPS;                  // floats per output channel in this tile
  float* red = smem;                               // [16 * NFO][ROWF]
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int f = 0; f < NFO; ++f)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float* dst = red + (16 * f + i16) * ROWF + (4 * kq + q) * TAPS + t;
            *dst = w == 0 ? acc[t][f][q] : *dst + acc[t][f][q];
          }
    }
    __syncthreads();
  }
  const int ci_n = min(16, a.cin - ci0);           // valid input channels of this tile
  for (int e = tid; e < 16 * NFO * ROWF; e += 256) {
    const int col = e / ROWF, r = e - col * ROWF;
    const int co = co0 + col;
    if (co < a.cout && r < ci_n * TAPS) {
      float* dst = a.dw + (int64_t)co * a.dw_so + (int64_t)ci0 * a.dw_si + r;
      if (a.nsplit == 1) *dst = red[e];
      else atomicAdd(dst, red[e]);
    }
  }
}

int tconv_wgrad(const adx_tconv_desc* d, const adx_tconv_io* io, const float* dc, float* dw, hipStream_t s, bool zero) {
  int rc = tconv_check(d);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(d->kind == 0, "tconv_wgrad: express a transposed conv's weight gradient as a strided conv with x and dy swapped");
  ADX_REQUIRE(io && io->x0 && dc && dw, "tconv_wgrad: null tensor");
  ADX_REQUIRE(d->c1 == 0 || io->x1 != nullptr, "tconv_wgrad: c1 > 0 but x1 is null");
  WgradArgs a;
  memset(&a, 0, sizeof(a));
  a.x0 = io->x0; a.x0_sb = io->x0_sb; a.x0_sc = io->x0_sc; a.x0_sl = io->x0_sl;
  a.x1 = io->x1; a.x1_sb = io->x1_sb; a.x1_sc = io->x1_sc; a.x1_sl = io->x1_sl;
  a.dc = dc; a.dw = dw;
  a.c0 = d->c0; a.cin = d->c0 + d->c1; a.cout = d->cout; a.lin = d->lin; a.lout = d->lout;
  a.lin_valid = d->lin_valid > 0 ? d->lin_valid : d->lin;
  a.lout_valid = d->lout_valid > 0 ? d->lout_valid : d->lout;
  a.log2_lout = 0;
  while ((1 << a.log2_lout) < d->lout) ++a.log2_lout;
  a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
  a.batch = io->batch;
  // (sample, position) rows are consumed 4 at a time; staged super-tiles hold whole K-steps when possible
  const int per4 = d->lout >= 4 ? 1 : 4 / d->lout;   // samples per MFMA K-step
  const int nfo = d->cout >= 64 ? 4 : (d->cout > 16 ? 2 : 1);
  a.n_ci_tiles = ceil_div(a.cin, 16);
  a.n_co_tiles = ceil_div(d->cout, 16 * nfo);
  const int tiles = a.n_ci_tiles * a.n_co_tiles;
  // split the batch so that ~256 workgroups exist, each with at least `per4` samples
  int nsplit = ceil_div(256, tiles);
  int nb = ceil_div(io->batch, nsplit);
  nb = round_up(nb < per4 ? per4 : nb, per4);
  nsplit = ceil_div(io->batch, nb);
  a.nb = nb; a.nsplit = nsplit;
  const int pr = (d->lout - 1) * d->stride + d->taps - 1 - d->pad - (d->lin - 1);
  a.pl = d->pad;
  a.lp = a.pl + d->lin + (pr > 0 ? pr : 0);
  int sbt = 128 / d->lout;                 // 128 (sample, position) rows = 8 K-steps per wave per super-tile
  if (sbt < per4) sbt = per4;
  sbt = sbt / per4 * per4;
  if (sbt > nb) sbt = nb;
  a.sbt = sbt;
  a.rs = sbt * a.lp + 1;                   // odd pitch: the 16 channel rows of an A fragment hit distinct banks
  ADX_REQUIRE((sbt * d->lout) % 4 == 0, "tconv_wgrad: staged tile not a multiple of 4 rows");
  a.dw_so = (int64_t)a.cin * d->taps; a.dw_si = d->taps;
  if (zero) ADX_CHECK_HIP(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->cout * a.cin * d->taps, s));   // else: pre-zeroed by the caller
  const size_t lds_stage = (size_t)16 * a.rs, lds_red = (size_t)16 * nfo * 16 * d->taps;
  const size_t lds = sizeof(float) * (lds_stage > lds_red ? lds_stage : lds_red);
  const dim3 grid((unsigned)(tiles * nsplit)), blk(256);
#define ADX_WG(NFO, TAPS) tconv_wgrad_kernel<NFO, TAPS><<<grid, blk, lds, s>>>(a)
#define ADX_WG_T(NFO)                                  \
  switch (d->taps) {                                   \
    case 1: ADX_WG(NFO, 1); break;                     \
    case 3: ADX_WG(NFO, 3); break;                     \
    case 4: ADX_WG(NFO, 4); break;                     \
    case 5: ADX_WG(NFO, 5); break;                     \
    default: set_error("tconv_wgrad: taps %d unsupported", d->taps); return ADX_ERR_INVALID; \
  }
  if (nfo == 4) { ADX_WG_T(4) } else if (nfo == 2) { ADX_WG_T(2) } else { ADX_WG_T(1) }
#undef ADX_WG_T
#undef ADX_WG
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

using namespace adx;

extern "C" {

int adx_gn_mish_backward(const float* dy, int64_t dy_sb, int64_t dy_sc, int64_t dy_sl, const float* pre,
                         const float* stats, const float* gamma, const float* beta, float* dc, float* dgamma,
                         float* dbeta, float* dbias, float* dtb, int64_t dtb_stride, int32_t B, int32_t C, int32_t L,
                         int32_t groups, adx_stream stream) {
  ADX_REQUIRE(B >= 1 && C >= 1 && L >= 1 && groups >= 1 && C % groups == 0, "adx_gn_mish_backward: bad shape");
  GnBwdArgs a;
  a.dy = dy; a.dy_sb = dy_sb; a.dy_sc = dy_sc; a.dy_sl = dy_sl;
  a.pre = pre; a.stats = stats; a.gamma = gamma; a.beta = beta; a.dc = dc;
  a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.dtb = dtb; a.dtb_stride = dtb_stride;
  a.B = B; a.C = C; a.L = L; a.G = groups; a.cg = C / groups;
  a.Lv = L;
  a.log2_L = 0;
  while ((1 << a.log2_L) < L) ++a.log2_L;
  ADX_REQUIRE((1 << a.log2_L) == L, "adx_gn_mish_backward: L must be a power of two");
  return gn_mish_backward(a, (hipStream_t)stream);
}

int adx_tconv_wgrad(const adx_tconv_desc* d, const adx_tconv_io* io, const float* dc, float* dw, adx_stream stream) {
  return tconv_wgrad(d, io, dc, dw, (hipStream_t)stream, true);
}

int adx_bias_grad(const float* dc, float* db, int32_t B, int32_t C, int32_t L, adx_stream stream) {
  return bias_grad(dc, (int64_t)C * L, L, 1, db, B, C, L, (hipStream_t)stream, 0);
}

}  // extern "C"
