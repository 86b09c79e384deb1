// TrajPredict (state head of classifier guidance) forward and input-gradient, one workgroup per
// sample, every activation of a layer resident in LDS.
//   reference: modeling/helpers.py:22-59 (Linear(3,64) + SinusoidalPosEmb + time_embed ->
//   2 x post-norm nn.TransformerEncoderLayer(d=64, 4 heads, ff=256, SiLU) -> LayerNorm -> Linear(64,4)),
//   used at modeling/temporal.py:238-241 and interact.py:158-160; its input gradient is what
//   GuidanceLoss asks autograd for (control/guidance.py:47-50).
//
// T = horizon-1 <= 31 rows x 64 features per sample is a 8 KB tile (32..63 rows: see kMaxTP), so this is latency-bound
// small-matrix work: plain fp32 FMA with the weight read coalesced (K-major copy for x @ W^T,
// PyTorch layout for the transposed products of the backward pass) and the activation operand
// broadcast from LDS, 8 rows per thread.  The backward kernel saves nothing from the forward:
// it keeps the three layer inputs and recomputes each layer's internals right before
// back-propagating through it (dropout is inactive: eval mode).
//
// The fused guidance kernel (adx_guided_output) additionally evaluates TargetGuidance
// (control/guidance_loss.py:10-22) per sample, back-propagates its gradient to the action and
// applies GuidanceLoss.forward's update + clip (control/guidance.py:51-59) in the same launch.
#include "adx_common.h"

namespace adx {

constexpr int E = 64;        // hidden_dim (must equal MODEL.DIM)
constexpr int NH = 4;        // heads
constexpr int DH = 16;       // head dim
constexpr int FF = 256;      // dim_feedforward
constexpr int kMaxTP = 64;   // padded rows: the kernels are instantiated for TP = 32 (T <= 31, everything in LDS) and TP = 64
                             // (T <= 63: QKV, the attention probabilities and the feed-forward tile live in a global scratch)
constexpr int NT = 1024;     // threads per workgroup (one workgroup per sample; LDS allows one per CU anyway: 4 waves per SIMD
                             // hide the latencies of the ~60 dependent phases a single wave per SIMD exposed)
constexpr int NL = 2;        // encoder layers
constexpr int IN_DIM = 3;

// packed parameter block (floats), produced by trajpred_pack_kernel
struct TPLayer {
  int w_in_t, w_in, b_in;        // in_proj  [192][64]: K-major copy, original, bias
  int w_out_t, w_out, b_out;     // out_proj [64][64]
  int w1_t, w1, b1;              // linear1  [256][64]
  int w2_t, w2, b2;              // linear2  [64][256]
  int g1, be1, g2, be2;          // norm1 / norm2
};
struct TPLayout {
  int w_ip, b_ip;                // input_proj [64][3], bias
  TPLayer layer[NL];
  int gf, bef;                   // final LayerNorm
  int w_op, b_op;                // output_proj [out][64], bias
  int freqs;                     // 32 sinusoidal frequencies (host-computed like helpers.py:66-69)
  int total;
  int out_dim;
};

static TPLayout make_layout(int out_dim) {
  TPLayout L;
  int o = 0;
  auto take = [&](int n) { int r = o; o += (n + 3) & ~3; return r; };
  L.w_ip = take(E * IN_DIM); L.b_ip = take(E);
  for (int l = 0; l < NL; ++l) {
    TPLayer& y = L.layer[l];
    y.w_in_t = take(3 * E * E); y.w_in = take(3 * E * E); y.b_in = take(3 * E);
    y.w_out_t = take(E * E); y.w_out = take(E * E); y.b_out = take(E);
    y.w1_t = take(FF * E); y.w1 = take(FF * E); y.b1 = take(FF);
    y.w2_t = take(E * FF); y.w2 = take(E * FF); y.b2 = take(E);
    y.g1 = take(E); y.be1 = take(E); y.g2 = take(E); y.be2 = take(E);
  }
  L.gf = take(E); L.bef = take(E);
  L.w_op = take(out_dim * E); L.b_op = take(out_dim);
  L.freqs = take(E / 2);
  L.total = o;
  L.out_dim = out_dim;
  return L;
}

__global__ void transpose_copy_kernel(const float* __restrict__ w, float* __restrict__ wt, float* __restrict__ wc,
                                      int rows, int cols) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * cols) return;
  const int r = idx / cols, c = idx - r * cols;
  const float v = w[idx];
  wc[idx] = v;
  wt[c * rows + r] = v;  // K-major: wt[k][n] = w[n][k]
}

// ---------------------------------------------------------------------------------------------
// small GEMMs on an LDS-resident [TP][K] tile
// out[t][n] = sum_k in[t][k] * wt[k][n] (+ bias[n]);  wt is K-major in global memory
// RT = rows per thread: chosen per call so that N * (TP / RT) items fill the workgroup (N = 64: RT 2; 192, 256: RT 8)
template <int TP, bool ACCUM, int RT>
__device__ __forceinline__ void mm_fwd(float* out, int ldo, const float* in, int ldi, const float* __restrict__ wt,
                                       const float* __restrict__ bias, int K, int N, int tid) {
  for (int item = tid; item < N * (TP / RT); item += NT) {
    const int n = item % N, tg = item / N;
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
    const float* ip = in + tg * RT * ldi;
    // four k per step: one 16-byte LDS read per row instead of four 4-byte ones (the loop was LDS-issue bound); the
    // products are still added in ascending k (K % 4 == 0 and every tile is 16-byte aligned: E, FF multiples of 64)
#pragma unroll 2
    for (int k = 0; k < K; k += 4) {
      const float w0 = wt[(size_t)k * N + n], w1 = wt[(size_t)(k + 1) * N + n], w2 = wt[(size_t)(k + 2) * N + n],
                  w3 = wt[(size_t)(k + 3) * N + n];
#pragma unroll
      for (int r = 0; r < RT; ++r) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(ip + r * ldi + k);
        acc[r] += x[0] * w0;
        acc[r] += x[1] * w1;
        acc[r] += x[2] * w2;
        acc[r] += x[3] * w3;
      }
    }
    const float b = bias != nullptr ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      float* o = out + (tg * RT + r) * ldo + n;
      *o = ACCUM ? *o + acc[r] + b : acc[r] + b;
    }
  }
}

__device__ __forceinline__ void layer_norm_rows(float* y, const float* x, int ld, const float* __restrict__ g,
                                                const float* __restrict__ b, float* xhat, float* rstd_out, int T,
                                                int tid) {
  // one wave per row, 64 features = 64 lanes
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < T; t += NT / 64) {
    const float v = x[t * ld + lane];
    const float mean = wave_sum(v) * (1.0f / E);
    const float d = v - mean;
    const float var = wave_sum(d * d) * (1.0f / E);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    const float xh = d * rstd;
    if (xhat != nullptr) xhat[t * ld + lane] = xh;
    if (rstd_out != nullptr && lane == 0) rstd_out[t] = rstd;
    y[t * ld + lane] = xh * g[lane] + b[lane];
  }
}

// dx = rstd * (dxh - mean(dxh) - xhat * mean(dxh * xhat)),  dxh = dy * gamma
__device__ __forceinline__ void layer_norm_bwd_rows(float* dx, const float* dy, const float* xhat, const float* rstd,
                                                    int ld, const float* __restrict__ g, int T, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < T; t += NT / 64) {
    const float dxh = dy[t * ld + lane] * g[lane];
    const float xh = xhat[t * ld + lane];
    const float m1 = wave_sum(dxh) * (1.0f / E);
    const float m2 = wave_sum(dxh * xh) * (1.0f / E);
    dx[t * ld + lane] = rstd[t] * (dxh - m1 - xh * m2);
  }
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float silu_grad(float x) {
  const float s = 1.0f / (1.0f + expf(-x));
  return s * (1.0f + x * (1.0f - s));
}

// parameter-gradient accumulation (training): gW[n][k] += sum_t A[t][n] * B[t][k], gb[n] += sum_t A[t][n].
// One workgroup per sample adds into the shared gradient image with float atomics.
template <typename FA>
__device__ __forceinline__ void outer_acc(float* gW, int N, int K, FA fa, const float* Bm, int ldb, int T, int tid) {
  for (int idx = tid; idx < N * K; idx += NT) {
    const int n = idx / K, k = idx - n * K;
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += fa(t, n) * Bm[t * ldb + k];
    atomicAdd(gW + idx, acc);
  }
}
template <typename FA>
__device__ __forceinline__ void colsum_acc(float* gb, int N, FA fa, int T, int tid) {
  for (int n = tid; n < N; n += NT) {
    float acc = 0.f;
    for (int t = 0; t < T; ++t) acc += fa(t, n);
    atomicAdd(gb + n, acc);
  }
}

// LDS plan (floats).  X: layer input; QKV; P: attention probs [NH][TP][TP]; O: attention output,
// later reused; Y: pre-norm sums / scratch; H1: norm1 output; F: feed-forward pre-activation;
// XH1/XH2: normalised values for the LayerNorm backward; small per-row vectors at the end.
template <int TP>
struct Lds {
  float* X; float* QKV; float* P; float* O; float* Y; float* H1; float* F; float* XH1; float* XH2; float* R1; float* R2; float* R3;
  float* INb;   // NL + 1 saved [TP][E] tiles (layer inputs + gradient scratch)
  __device__ __forceinline__ float* IN(int i) const { return INb + i * TP * E; }
};
// TP = 32: everything in LDS (148 KB).  TP = 64: the nine [TP][E] tiles and the row vectors (145 KB); QKV, P and F (176 KB per
// sample) in global memory, where one workgroup's writes reach its own later reads through __syncthreads() like LDS ones.
template <int TP> constexpr int big_floats() { return TP * 3 * E + NH * TP * TP + TP * FF; }
template <int TP> constexpr int lds_floats() { return TP * E * 6 + 3 * TP + (NL + 1) * TP * E + (TP == 32 ? big_floats<TP>() : 0); }

template <int TP>
__device__ __forceinline__ Lds<TP> carve(float* s, float* big) {
  Lds<TP> l;
  if (TP == 32) { big = s; s += big_floats<TP>(); }
  l.QKV = big; big += TP * 3 * E;
  l.P = big; big += NH * TP * TP;
  l.F = big;
  l.X = s; s += TP * E;
  l.O = s; s += TP * E;
  l.Y = s; s += TP * E;
  l.H1 = s; s += TP * E;
  l.XH1 = s; s += TP * E;
  l.XH2 = s; s += TP * E;
  l.R1 = s; s += TP;
  l.R2 = s; s += TP;
  l.R3 = s; s += TP;
  l.INb = s;
  return l;
}

// x0 = input_proj(action) + pos_emb(arange(T)) + time_embed   (helpers.py:52-57)
template <int TP>
__device__ __forceinline__ void embed_rows(float* X, const float* __restrict__ act, int64_t act_stride,
                                           const float* __restrict__ te, const float* __restrict__ P,
                                           const TPLayout& L, int T, int tid) {
  for (int idx = tid; idx < TP * E; idx += NT) {
    const int t = idx >> 6, j = idx & 63;
    float v = 0.f;
    if (t < T) {
      const float* a = act + (int64_t)t * act_stride;
      v = P[L.b_ip + j];
#pragma unroll
      for (int i = 0; i < IN_DIM; ++i) v += a[i] * P[L.w_ip + j * IN_DIM + i];
      const int fi = j < 32 ? j : j - 32;
      const float arg = (float)t * P[L.freqs + fi];
      v += j < 32 ? sinf(arg) : cosf(arg);
      v += te[j];
    }
    X[idx] = v;
  }
}

// Train-mode dropout of nn.TransformerEncoderLayer (p = 0.1 in the reference: attention probabilities, the two
// residual branches, the feed-forward activation).  Masks are never stored: element idx of site s of sample b keeps
// its value iff lowbias32(base(seed, b, s) ^ idx) >= p * 2^32, so the forward pass, the recomputation inside the
// backward pass and the gradient formulas all regenerate the same mask.  (torch draws its masks from its own Philox
// stream; the two can only agree in distribution.)
struct Drop {
  uint32_t base;      // per (seed, sample) key
  uint32_t thresh;    // p * 2^32; 0 = dropout off
  float scale;        // 1 / (1 - p)
};
__device__ __forceinline__ uint32_t lowbias32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_site(const Drop& d, int site) { return lowbias32(d.base ^ ((uint32_t)site * 0x85EBCA6BU)); }
// multiplier of element idx at a site: 0 or 1/(1-p)
__device__ __forceinline__ float drop_mul(const Drop& d, uint32_t site_key, uint32_t idx) {
  if (d.thresh == 0u) return 1.f;
  return lowbias32(site_key ^ idx) >= d.thresh ? d.scale : 0.f;
}
__device__ __forceinline__ Drop make_drop(uint32_t seed_lo, uint32_t seed_hi, uint32_t thresh, float scale, int b) {
  Drop d;
  d.base = lowbias32(lowbias32(seed_lo ^ ((uint32_t)b * 0x9E3779B9U)) ^ seed_hi);
  d.thresh = thresh;
  d.scale = scale;
  return d;
}

// one encoder layer, forward; leaves every intermediate the backward needs in LDS
template <int TP>
__device__ __forceinline__ void layer_forward(const Lds<TP>& l, const float* __restrict__ P, const TPLayer& y, int T,
                                              int tid, const Drop& dr, int li) {
  const uint32_t k_att = drop_site(dr, 4 * li + 0), k_d1 = drop_site(dr, 4 * li + 1), k_in = drop_site(dr, 4 * li + 2),
                 k_d2 = drop_site(dr, 4 * li + 3);
  mm_fwd<TP, false, 8>(l.QKV, 3 * E, l.X, E, P + y.w_in_t, P + y.b_in, E, 3 * E, tid);
  __syncthreads();
  // scores + softmax: one thread per (head, query row)
  for (int idx = tid; idx < NH * TP; idx += NT) {
    const int h = idx / TP, t = idx - h * TP;
    float* p = l.P + (h * TP + t) * TP;
    if (t < T) {
      const float* q = l.QKV + t * 3 * E + h * DH;
      float mx = -INFINITY;
      for (int s = 0; s < T; ++s) {
        const float* k = l.QKV + s * 3 * E + E + h * DH;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < DH; ++i) d += q[i] * k[i];
        d *= 0.25f;  // 1 / sqrt(16)
        p[s] = d;
        mx = fmaxf(mx, d);
      }
      float sum = 0.f;
      for (int s = 0; s < T; ++s) {
        const float e = expf(p[s] - mx);
        p[s] = e;
        sum += e;
      }
      const float inv = 1.0f / sum;
      for (int s = 0; s < T; ++s) p[s] *= inv;
    }
  }
  __syncthreads();
  // O = P @ V
  for (int idx = tid; idx < TP * E; idx += NT) {
    const int t = idx >> 6, c = idx & 63, h = c >> 4;
    float acc = 0.f;
    if (t < T) {
      const float* p = l.P + (h * TP + t) * TP;     // softmax probabilities stay unmasked in LDS
      for (int s = 0; s < T; ++s)
        acc += p[s] * drop_mul(dr, k_att, (uint32_t)((h * TP + t) * TP + s)) * l.QKV[s * 3 * E + 2 * E + c];
    }
    l.O[idx] = acc;
  }
  __syncthreads();
  // Y = X + dropout1(out_proj(O))   (H1 is free until the LayerNorm below writes it)
  mm_fwd<TP, false, 2>(l.H1, E, l.O, E, P + y.w_out_t, P + y.b_out, E, E, tid);
  __syncthreads();
  for (int idx = tid; idx < TP * E; idx += NT) l.Y[idx] = l.X[idx] + l.H1[idx] * drop_mul(dr, k_d1, (uint32_t)idx);
  __syncthreads();
  layer_norm_rows(l.H1, l.Y, E, P + y.g1, P + y.be1, l.XH1, l.R1, T, tid);
  __syncthreads();
  mm_fwd<TP, false, 8>(l.F, FF, l.H1, E, P + y.w1_t, P + y.b1, E, FF, tid);
  __syncthreads();
  // Y = H1 + linear2(silu(F))
  for (int idx = tid; idx < TP * E; idx += NT) l.Y[idx] = l.H1[idx];
  __syncthreads();
  {
    // silu(F) @ W2^T, K = 256.  The activation (an exp and a division) is evaluated ONCE per element -- 64 columns of
    // silu(F) * mask at a time into X (dead since Y = X + sa was formed; norm2 rewrites it below) -- not once per (element, output column) inside the product loop,
    // which made this the longest phase of the layer.  Thread = (output column n, row group tg), accumulators live
    // across the four slabs; the products are added in ascending k as before.
    constexpr int RT = TP * E / NT;                // rows per thread: E * (TP / RT) items = one per thread
    static_assert(RT >= 1 && TP % RT == 0 && E * (TP / RT) == NT, "one (column, row group) item per thread");
    const int n = tid % E, tg = tid / E;
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
    const float* wt = P + y.w2_t;
    for (int k0 = 0; k0 < FF; k0 += E) {
      for (int idx = tid; idx < TP * E; idx += NT) {
        const int t = idx >> 6, j = idx & 63;
        l.X[idx] = silu_f(l.F[t * FF + k0 + j]) * drop_mul(dr, k_in, (uint32_t)(t * FF + k0 + j));
      }
      __syncthreads();
      const float* ip = l.X + tg * RT * E;
#pragma unroll 2
      for (int k = 0; k < E; k += 4) {
        const float w0 = wt[(size_t)(k0 + k) * E + n], w1 = wt[(size_t)(k0 + k + 1) * E + n],
                    w2 = wt[(size_t)(k0 + k + 2) * E + n], w3 = wt[(size_t)(k0 + k + 3) * E + n];
#pragma unroll
        for (int r = 0; r < RT; ++r) {
          const f32x4 x = *reinterpret_cast<const f32x4*>(ip + r * E + k);
          acc[r] += x[0] * w0;
          acc[r] += x[1] * w1;
          acc[r] += x[2] * w2;
          acc[r] += x[3] * w3;
        }
      }
      __syncthreads();
    }
    const float b = P[y.b2 + n];
#pragma unroll
    for (int r = 0; r < RT; ++r)
      l.Y[(tg * RT + r) * E + n] += (acc[r] + b) * drop_mul(dr, k_d2, (uint32_t)((tg * RT + r) * E + n));
  }
  __syncthreads();
  layer_norm_rows(l.X, l.Y, E, P + y.g2, P + y.be2, l.XH2, l.R2, T, tid);  // X becomes the layer output
  __syncthreads();
}

// back-propagate through one layer: D (in/out, [TP][E]) holds d(layer output) on entry and
// d(layer input) on exit.  Requires layer_forward() to have just run on this layer's input.
// G != nullptr (training): parameter gradients are accumulated into the gradient image G, which has the packed
// buffer's layout (PyTorch-layout slots); Xin = this layer's input tile.
template <int TP>
__device__ __forceinline__ void layer_backward(const Lds<TP>& l, float* D, const float* __restrict__ P, const TPLayer& y,
                                               int T, int tid, float* G, const float* Xin, const Drop& dr, int li) {
  const uint32_t k_att = drop_site(dr, 4 * li + 0), k_d1 = drop_site(dr, 4 * li + 1), k_in = drop_site(dr, 4 * li + 2),
                 k_d2 = drop_site(dr, 4 * li + 3);
  if (G != nullptr) {  // norm2 affine
    for (int i = tid; i < E; i += NT) {
      float sg = 0.f, sb = 0.f;
      for (int t = 0; t < T; ++t) { sg += D[t * E + i] * l.XH2[t * E + i]; sb += D[t * E + i]; }
      atomicAdd(G + y.g2 + i, sg);
      atomicAdd(G + y.be2 + i, sb);
    }
  }
  // norm2
  layer_norm_bwd_rows(l.Y, D, l.XH2, l.R2, E, P + y.g2, T, tid);   // Y = d(y2): the residual path's gradient ...
  __syncthreads();
  // ... and D (free until norm1's backward) = d(ff) = dropout2's mask on it
  for (int idx = tid; idx < TP * E; idx += NT) D[idx] = l.Y[idx] * drop_mul(dr, k_d2, (uint32_t)idx);
  __syncthreads();
  if (G != nullptr) {  // linear2: ff = W2 (mask * silu(F)) + b2, before F is overwritten
    for (int idx = tid; idx < E * FF; idx += NT) {
      const int j = idx / FF, n = idx - j * FF;
      float acc = 0.f;
      for (int t = 0; t < T; ++t)
        acc += D[t * E + j] * silu_f(l.F[t * FF + n]) * drop_mul(dr, k_in, (uint32_t)(t * FF + n));
      atomicAdd(G + y.w2 + idx, acc);
    }
    colsum_acc(G + y.b2, E, [&](int t, int n) { return D[t * E + n]; }, T, tid);
    __syncthreads();     // the loop below overwrites F, which the weight-gradient sums above still read
  }
  // ds = dff @ W2  ([T][256]); dF = ds * silu'(F), stored over F
  constexpr int RT = 8;          // FF * (TP / 8) = 1024 items
  for (int item = tid; item < FF * (TP / RT); item += NT) {
    const int n = item % FF, tg = item / FF;
    float acc[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) acc[r] = 0.f;
    const float* w = P + y.w2;  // [64][256]: row j, column n
    for (int j = 0; j < E; ++j) {
      const float wv = w[(size_t)j * FF + n];
#pragma unroll
      for (int r = 0; r < RT; ++r) acc[r] += D[(tg * RT + r) * E + j] * wv;
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
      float* f = l.F + (tg * RT + r) * FF + n;
      *f = acc[r] * drop_mul(dr, k_in, (uint32_t)((tg * RT + r) * FF + n)) * silu_grad(*f);
    }
  }
  __syncthreads();
  if (G != nullptr) {  // linear1: F = W1 H1 + b1 (F now holds dF)
    outer_acc(G + y.w1, FF, E, [&](int t, int n) { return l.F[t * FF + n]; }, l.H1, E, T, tid);
    colsum_acc(G + y.b1, FF, [&](int t, int n) { return l.F[t * FF + n]; }, T, tid);
  }
  // dH1 = dy2 + dF @ W1   (W1 is [256][64]: K = 256 rows, N = 64 contiguous)
  mm_fwd<TP, true, 2>(l.Y, E, l.F, FF, P + y.w1, nullptr, FF, E, tid);
  __syncthreads();
  if (G != nullptr) {  // norm1 affine
    for (int i = tid; i < E; i += NT) {
      float sg = 0.f, sb = 0.f;
      for (int t = 0; t < T; ++t) { sg += l.Y[t * E + i] * l.XH1[t * E + i]; sb += l.Y[t * E + i]; }
      atomicAdd(G + y.g1 + i, sg);
      atomicAdd(G + y.be1 + i, sb);
    }
  }
  // norm1: D = d(y1)
  layer_norm_bwd_rows(D, l.Y, l.XH1, l.R1, E, P + y.g1, T, tid);
  __syncthreads();
  // d(sa) = dropout1's mask on d(y1), kept in XH2 (dead since norm2's backward); D stays the residual path's gradient
  for (int idx = tid; idx < TP * E; idx += NT) l.XH2[idx] = D[idx] * drop_mul(dr, k_d1, (uint32_t)idx);
  __syncthreads();
  if (G != nullptr) {  // out_proj: sa = Wout O + bout
    outer_acc(G + y.w_out, E, E, [&](int t, int n) { return l.XH2[t * E + n]; }, l.O, E, T, tid);
    colsum_acc(G + y.b_out, E, [&](int t, int n) { return l.XH2[t * E + n]; }, T, tid);
  }
  // dO = dsa @ Wout  (Wout [64][64], row j = output feature)
  mm_fwd<TP, false, 2>(l.Y, E, l.XH2, E, P + y.w_out, nullptr, E, E, tid);    // Y = dO
  __syncthreads();
  // dP[h][t][s] = sum_d dO[t][hd] V[s][hd];  dS = P * (dP - sum_s dP P), written into F (dead by now)
  for (int idx = tid; idx < NH * TP; idx += NT) {
    const int h = idx / TP, t = idx - h * TP;
    if (t < T) {
      const float* p = l.P + (h * TP + t) * TP;
      float* dS = l.F + (h * TP + t) * TP;
      const float* dO = l.Y + t * E + h * DH;
      float dot = 0.f;
      for (int s = 0; s < T; ++s) {
        const float* v = l.QKV + s * 3 * E + 2 * E + h * DH;
        float d = 0.f;
#pragma unroll
        for (int i = 0; i < DH; ++i) d += dO[i] * v[i];
        d *= drop_mul(dr, k_att, (uint32_t)((h * TP + t) * TP + s));     // through the attention dropout
        dS[s] = d;
        dot += d * p[s];
      }
      for (int s = 0; s < T; ++s) dS[s] = p[s] * (dS[s] - dot);
    }
  }
  __syncthreads();
  // dV[s][c] = sum_t P[h][t][s] dO[t][c];  dQ[t][c] = sum_s dS[t][s] K[s][c] / 4;  dK[s][c] = sum_t dS[t][s] Q[t][c] / 4
  // written into O (dq), H1 (dk), XH1 (dv) -- all dead at this point
  for (int idx = tid; idx < TP * E; idx += NT) {
    const int t = idx >> 6, c = idx & 63, h = c >> 4;
    float dq = 0.f, dk = 0.f, dv = 0.f;
    if (t < T) {
      const float* dS = l.F + (h * TP + t) * TP;
      for (int s = 0; s < T; ++s) {
        dq += dS[s] * l.QKV[s * 3 * E + E + c];
        dk += l.F[(h * TP + s) * TP + t] * l.QKV[s * 3 * E + c];
        dv += l.P[(h * TP + s) * TP + t] * drop_mul(dr, k_att, (uint32_t)((h * TP + s) * TP + t)) * l.Y[s * E + c];
      }
    }
    l.O[idx] = dq * 0.25f;
    l.H1[idx] = dk * 0.25f;
    l.XH1[idx] = dv;
  }
  __syncthreads();
  if (G != nullptr) {  // in_proj: [q k v] = Win Xin + bin
    outer_acc(G + y.w_in, E, E, [&](int t, int n) { return l.O[t * E + n]; }, Xin, E, T, tid);
    outer_acc(G + y.w_in + E * E, E, E, [&](int t, int n) { return l.H1[t * E + n]; }, Xin, E, T, tid);
    outer_acc(G + y.w_in + 2 * E * E, E, E, [&](int t, int n) { return l.XH1[t * E + n]; }, Xin, E, T, tid);
    colsum_acc(G + y.b_in, E, [&](int t, int n) { return l.O[t * E + n]; }, T, tid);
    colsum_acc(G + y.b_in + E, E, [&](int t, int n) { return l.H1[t * E + n]; }, T, tid);
    colsum_acc(G + y.b_in + 2 * E, E, [&](int t, int n) { return l.XH1[t * E + n]; }, T, tid);
  }
  // dX = d(y1) + [dq dk dv] @ Win   (Win [192][64]); D already holds d(y1)
  mm_fwd<TP, true, 2>(D, E, l.O, E, P + y.w_in, nullptr, E, E, tid);
  mm_fwd<TP, true, 2>(D, E, l.H1, E, P + y.w_in + E * E, nullptr, E, E, tid);
  mm_fwd<TP, true, 2>(D, E, l.XH1, E, P + y.w_in + 2 * E * E, nullptr, E, E, tid);
  __syncthreads();
}

struct TrajArgs {
  const float* P;            // packed parameters
  TPLayout L;
  const float* action;       // [B][T(+1)][3] rows with stride act_stride
  int64_t act_sb, act_st;
  const float* te;           // [B][64]
  int B, T;
  // forward
  float* out; int64_t out_sb, out_st;         // [B][T][out_dim]
  // backward
  const float* gout; int64_t gout_sb, gout_st; // [B][T][out_dim]
  float* gact; int64_t gact_sb, gact_st;       // [B][T][3]
  // fused guidance
  const float* target;       // [B][2]
  float* xg;                 // [B][T+1][out_dim+3] guided output (state | action)
  float grad_scale, scale;   // model_std, GUIDANCE.CLASSIFIER_SCALE
  float* loss;               // [B] or null
  // training: gradient image with the packed buffer's layout (atomically accumulated) and d(time_embed) [B][64]
  float* G; float* dte;
  float* big;                // TP = 64: [B][big_floats<64>()] scratch for QKV / P / F (unused at TP = 32)
  // training: dropout (thresh = p * 2^32, 0 = off)
  uint32_t drop_thresh, seed_lo, seed_hi; float drop_scale;
};

// head: LayerNorm -> Linear(64, out_dim)
// The normalised values and 1/std of the final norm go to their own places (the IN(NL) tile, which the backward pass
// then turns into d(layer output) in place, and R3): the last layer's intermediates stay intact, so its backward needs
// no recomputation.
template <int TP>
__device__ __forceinline__ void head_forward(const Lds<TP>& l, const float* __restrict__ P, const TPLayout& L, int T,
                                             int tid) {
  layer_norm_rows(l.Y, l.X, E, P + L.gf, P + L.bef, l.IN(NL), l.R3, T, tid);
  __syncthreads();
}

template <int TP>
__global__ void __launch_bounds__(NT) trajpred_forward_kernel(const TrajArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds<TP> l = carve<TP>(smem, a.big + (size_t)blockIdx.x * big_floats<TP>());
  const int b = blockIdx.x, tid = threadIdx.x;
  embed_rows<TP>(l.X, a.action + (int64_t)b * a.act_sb, a.act_st, a.te + (int64_t)b * E, a.P, a.L, a.T, tid);
  __syncthreads();
  const Drop dr = make_drop(a.seed_lo, a.seed_hi, a.drop_thresh, a.drop_scale, b);
  for (int li = 0; li < NL; ++li) layer_forward<TP>(l, a.P, a.L.layer[li], a.T, tid, dr, li);
  head_forward<TP>(l, a.P, a.L, a.T, tid);
  const int od = a.L.out_dim;
  for (int idx = tid; idx < a.T * od; idx += NT) {
    const int t = idx / od, j = idx - t * od;
    float acc = a.P[a.L.b_op + j];
    for (int i = 0; i < E; ++i) acc += l.Y[t * E + i] * a.P[a.L.w_op + j * E + i];
    a.out[(int64_t)b * a.out_sb + (int64_t)t * a.out_st + j] = acc;
  }
}

// forward with every layer's input kept (IN(li)); on return the LAST layer's intermediates and the head's normalised
// output (l.Y), normalised values (IN(NL)) and 1/std (R3) are in LDS
template <int TP>
__device__ __forceinline__ void forward_keep(const Lds<TP>& l, const TrajArgs& a, int b, int tid, const Drop& dr) {
  const int T = a.T;
  embed_rows<TP>(l.X, a.action + (int64_t)b * a.act_sb, a.act_st, a.te + (int64_t)b * E, a.P, a.L, T, tid);
  __syncthreads();
  for (int li = 0; li < NL; ++li) {
    for (int idx = tid; idx < TP * E; idx += NT) l.IN(li)[idx] = l.X[idx];
    __syncthreads();
    layer_forward<TP>(l, a.P, a.L.layer[li], T, tid, dr, li);
  }
  head_forward<TP>(l, a.P, a.L, T, tid);
}

// shared by the backward and the fused guidance kernels, after forward_keep(): back-propagation of d(out) [T][out_dim]
// (given by `dout(t, j)`) down to d(action) [T][3] in l.O.  The last layer is back-propagated from the intermediates the
// forward left; the layers below are recomputed from their saved inputs first.
template <int TP, typename DOut>
__device__ __forceinline__ void backward_core(const Lds<TP>& l, const TrajArgs& a, int b, int tid, const Drop& dr, DOut dout) {
  const float* P = a.P;
  const TPLayout& L = a.L;
  const int T = a.T, od = L.out_dim;
  float* G = a.G;
  if (G != nullptr) {  // output_proj: out = Wop Ynorm + bop
    for (int idx = tid; idx < od * E; idx += NT) {
      const int j = idx / E, i = idx - j * E;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += dout(t, j) * l.Y[t * E + i];
      atomicAdd(G + L.w_op + idx, acc);
    }
    for (int j = tid; j < od; j += NT) {
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += dout(t, j);
      atomicAdd(G + L.b_op + j, acc);
    }
  }
  // d(normed) = dout @ Wop  -> X (the last layer's output: consumed by the head already)
  for (int idx = tid; idx < TP * E; idx += NT) {
    const int t = idx >> 6, i = idx & 63;
    float acc = 0.f;
    if (t < T)
      for (int j = 0; j < od; ++j) acc += dout(t, j) * P[L.w_op + j * E + i];
    l.X[idx] = acc;
  }
  __syncthreads();
  float* D = l.IN(NL);          // holds the final norm's normalised values; becomes d(last layer output) in place
  if (G != nullptr) {  // final LayerNorm affine
    for (int i = tid; i < E; i += NT) {
      float sg = 0.f, sb = 0.f;
      for (int t = 0; t < T; ++t) { sg += l.X[t * E + i] * D[t * E + i]; sb += l.X[t * E + i]; }
      atomicAdd(G + L.gf + i, sg);
      atomicAdd(G + L.bef + i, sb);
    }
    __syncthreads();
  }
  layer_norm_bwd_rows(D, l.X, D, l.R3, E, P + L.gf, T, tid);    // each lane reads its own element before writing it
  for (int idx = tid + 0; idx < TP * E; idx += NT)
    if ((idx >> 6) >= T) D[idx] = 0.f;
  __syncthreads();
  for (int li = NL - 1; li >= 0; --li) {
    if (li != NL - 1) {
      for (int idx = tid; idx < TP * E; idx += NT) l.X[idx] = l.IN(li)[idx];
      __syncthreads();
      layer_forward<TP>(l, P, L.layer[li], T, tid, dr, li);     // recompute this layer's internals (same masks)
    }
    layer_backward<TP>(l, D, P, L.layer[li], T, tid, G, l.IN(li), dr, li);
    for (int idx = tid; idx < TP * E; idx += NT)
      if ((idx >> 6) >= T) D[idx] = 0.f;
    __syncthreads();
  }
  if (G != nullptr) {  // input_proj: x0 = Wip a + bip (+ pos + time_embed)
    const float* act = a.action + (int64_t)b * a.act_sb;
    for (int idx = tid; idx < E * IN_DIM; idx += NT) {
      const int j = idx / IN_DIM, i = idx - j * IN_DIM;
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += D[t * E + j] * act[(int64_t)t * a.act_st + i];
      atomicAdd(G + L.w_ip + idx, acc);
    }
    for (int j = tid; j < E; j += NT) {
      float acc = 0.f;
      for (int t = 0; t < T; ++t) acc += D[t * E + j];
      atomicAdd(G + L.b_ip + j, acc);
      if (a.dte != nullptr) a.dte[(int64_t)b * E + j] = acc;   // time_embed is added to every row
    }
  }
  // d(action)[t][i] = sum_j D[t][j] Wip[j][i]
  for (int idx = tid; idx < TP * IN_DIM; idx += NT) {
    const int t = idx / IN_DIM, i = idx - t * IN_DIM;
    float acc = 0.f;
    if (t < T)
      for (int j = 0; j < E; ++j) acc += D[t * E + j] * P[L.w_ip + j * IN_DIM + i];
    l.O[idx] = acc;
  }
  __syncthreads();
}

template <int TP>
__global__ void __launch_bounds__(NT) trajpred_backward_kernel(const TrajArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds<TP> l = carve<TP>(smem, a.big + (size_t)blockIdx.x * big_floats<TP>());
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* g = a.gout + (int64_t)b * a.gout_sb;
  const int64_t gst = a.gout_st;
  const Drop dr = make_drop(a.seed_lo, a.seed_hi, a.drop_thresh, a.drop_scale, b);
  forward_keep<TP>(l, a, b, tid, dr);
  backward_core<TP>(l, a, b, tid, dr, [&](int t, int j) { return g[(int64_t)t * gst + j]; });
  if (a.gact != nullptr)
    for (int idx = tid; idx < a.T * IN_DIM; idx += NT) {
      const int t = idx / IN_DIM, i = idx - t * IN_DIM;
      a.gact[(int64_t)b * a.gact_sb + (int64_t)t * a.gact_st + i] = l.O[idx];
    }
}

// Fused classifier guidance for one sample (GUIDANCE.STEP = 1):
//   x = cat([0; state_pred(action[:-1])], action);  choose h* by TargetGuidance's rule;
//   g_x = 2 (x[h*, :2] - target) at (h*, :2);  g_a = d(state)/d(action)^T g_x[1:, :4];
//   x[:, :4] -= scale/15 * std * g_x[:, :4];  x[:, 4:] -= scale * std * g_a;  clip(-1, 1)
template <int TP>
__global__ void __launch_bounds__(NT) guided_output_kernel(const TrajArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Lds<TP> l = carve<TP>(smem, a.big + (size_t)blockIdx.x * big_floats<TP>());
  __shared__ float st[TP + 1][4];
  __shared__ int hstar;
  __shared__ float gxy[2];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int T = a.T, H = T + 1, od = a.L.out_dim;
  const float* act = a.action + (int64_t)b * a.act_sb;
  // ONE forward (layer inputs kept) serves both the state rows and the gradient below
  const Drop dr_off{0u, 0u, 1.f};      // guidance runs the state head in eval mode
  forward_keep<TP>(l, a, b, tid, dr_off);
  for (int idx = tid; idx < H * od; idx += NT) {
    const int h = idx / od, j = idx - h * od;
    float acc = 0.f;
    if (h > 0) {
      acc = a.P[a.L.b_op + j];
      for (int i = 0; i < E; ++i) acc += l.Y[(h - 1) * E + i] * a.P[a.L.w_op + j * E + i];
    }
    st[h][j] = acc;
  }
  __syncthreads();
  if (tid == 0) {
    const float tx = a.target[2 * b], ty = a.target[2 * b + 1];
    const float x0 = st[0][0], y0 = st[0][1];
    const float t2a = sqrtf((tx - x0) * (tx - x0) + (ty - y0) * (ty - y0));
    const float fx = st[H - 1][0] - x0, fy = st[H - 1][1] - y0;
    const float f2a = sqrtf(fx * fx + fy * fy);
    int best = 0;
    if (!(f2a < t2a)) {
      float bd = INFINITY;
      for (int h = 0; h < H; ++h) {
        const float dx = st[h][0] - tx, dy = st[h][1] - ty;
        const float d = dx * dx + dy * dy;
        if (d < bd) { bd = d; best = h; }   // first minimum, like torch.argmin
      }
    }
    hstar = best;
    const float dx = st[best][0] - tx, dy = st[best][1] - ty;
    gxy[0] = 2.f * dx;
    gxy[1] = 2.f * dy;
    if (a.loss != nullptr) a.loss[b] = dx * dx + dy * dy;
  }
  __syncthreads();
  const int hs = hstar;
  const float g0 = gxy[0], g1 = gxy[1];
  // gradient w.r.t. the action through the state path (row h* of x is state row h*-1; row 0 is the dummy zero)
  if (hs > 0) {
    backward_core<TP>(l, a, b, tid, dr_off, [&](int t, int j) { return (t == hs - 1 && j < 2) ? (j == 0 ? g0 : g1) : 0.f; });
  } else {
    for (int idx = tid; idx < TP * IN_DIM; idx += NT) l.O[idx] = 0.f;
    __syncthreads();
  }
  const int dims = od + IN_DIM;
  const float s_state = a.scale / 15.f * a.grad_scale, s_act = a.scale * a.grad_scale;
  float* xo = a.xg + (int64_t)b * H * dims;
  for (int idx = tid; idx < H * dims; idx += NT) {
    const int h = idx / dims, j = idx - h * dims;
    float v;
    if (j < od) {
      v = st[h][j];
      if (h == hs && j < 2) v -= s_state * (j == 0 ? g0 : g1);
    } else {
      v = act[(int64_t)h * a.act_st + (j - od)];
      if (h < T) v -= s_act * l.O[h * IN_DIM + (j - od)];
    }
    xo[idx] = v < -1.f ? -1.f : (v > 1.f ? 1.f : v);
  }
}

}  // namespace adx

using namespace adx;

struct adx_trajpred {
  TPLayout L;
  bool packed = false;
  float* big = nullptr;        // T >= 32: per-sample scratch of the 64-row kernels (adx_trajpred_set_scratch; caller-owned)
  size_t big_bytes = 0;
};

extern "C" {

int adx_trajpred_create(int32_t out_dim, adx_trajpred** out) {
  ADX_REQUIRE(out != nullptr && out_dim >= 2 && out_dim <= 4, "adx_trajpred_create: out_dim %d must be 2..4", out_dim);
  adx_trajpred* t = new adx_trajpred();
  t->L = make_layout(out_dim);
  *out = t;
  return ADX_OK;
}
void adx_trajpred_destroy(adx_trajpred* t) { delete t; }

// Sequences of 32..63 rows run the 64-row kernels, which keep QKV, the attention probabilities and the feed-forward tile
// of every sample in global memory: the caller lends that scratch (no initialisation needed; it must stay alive and
// unshared while launches that use it are in flight).  0 bytes for T <= 31.
size_t adx_trajpred_scratch_bytes(const adx_trajpred* t, int32_t batch, int32_t T) {
  return (t && T >= 32 && batch > 0) ? (size_t)batch * big_floats<64>() * sizeof(float) : 0;
}
int adx_trajpred_set_scratch(adx_trajpred* t, void* scratch, size_t bytes) {
  ADX_REQUIRE(t != nullptr && (scratch != nullptr || bytes == 0), "adx_trajpred_set_scratch: null argument");
  t->big = (float*)scratch;
  t->big_bytes = bytes;
  return ADX_OK;
}
int adx_trajpred_num_params(const adx_trajpred* t) { return t ? 2 + NL * 12 + 2 + 2 : 0; }
size_t adx_trajpred_packed_bytes(const adx_trajpred* t) { return t ? (size_t)t->L.total * sizeof(float) : 0; }

// params: state_pred.* in named_parameters() order (input_proj w,b; per layer in_proj_weight,
// in_proj_bias, out_proj w,b, linear1 w,b, linear2 w,b, norm1 w,b, norm2 w,b; final norm w,b;
// output_proj w,b); freqs = exp(-i ln(1e4)/31), i < 32.
int adx_trajpred_pack(adx_trajpred* t, const float* const* P, int32_t n, const float* freqs, void* packed,
                      adx_stream stream) {
  ADX_REQUIRE(t && P && freqs && packed, "adx_trajpred_pack: null argument");
  ADX_REQUIRE(n == adx_trajpred_num_params(t), "adx_trajpred_pack: expected %d tensors, got %d",
              adx_trajpred_num_params(t), n);
  hipStream_t s = (hipStream_t)stream;
  float* base = (float*)packed;
  const TPLayout& L = t->L;
  auto cp = [&](int off, const float* src, int cnt) -> int {
    ADX_CHECK_HIP(hipMemcpyAsync(base + off, src, cnt * sizeof(float), hipMemcpyDeviceToDevice, s));
    return ADX_OK;
  };
  auto tr = [&](int off_t, int off_c, const float* src, int rows, int cols) -> int {
    transpose_copy_kernel<<<dim3(ceil_div(rows * cols, 256)), dim3(256), 0, s>>>(src, base + off_t, base + off_c, rows, cols);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  };
  int i = 0, rc = ADX_OK;
  rc = cp(L.w_ip, P[i++], E * IN_DIM); if (rc) return rc;
  rc = cp(L.b_ip, P[i++], E); if (rc) return rc;
  for (int l = 0; l < NL; ++l) {
    const TPLayer& y = L.layer[l];
    rc = tr(y.w_in_t, y.w_in, P[i++], 3 * E, E); if (rc) return rc;
    rc = cp(y.b_in, P[i++], 3 * E); if (rc) return rc;
    rc = tr(y.w_out_t, y.w_out, P[i++], E, E); if (rc) return rc;
    rc = cp(y.b_out, P[i++], E); if (rc) return rc;
    rc = tr(y.w1_t, y.w1, P[i++], FF, E); if (rc) return rc;
    rc = cp(y.b1, P[i++], FF); if (rc) return rc;
    rc = tr(y.w2_t, y.w2, P[i++], E, FF); if (rc) return rc;
    rc = cp(y.b2, P[i++], E); if (rc) return rc;
    rc = cp(y.g1, P[i++], E); if (rc) return rc;
    rc = cp(y.be1, P[i++], E); if (rc) return rc;
    rc = cp(y.g2, P[i++], E); if (rc) return rc;
    rc = cp(y.be2, P[i++], E); if (rc) return rc;
  }
  rc = cp(L.gf, P[i++], E); if (rc) return rc;
  rc = cp(L.bef, P[i++], E); if (rc) return rc;
  rc = cp(L.w_op, P[i++], L.out_dim * E); if (rc) return rc;
  rc = cp(L.b_op, P[i++], L.out_dim); if (rc) return rc;
  rc = cp(L.freqs, freqs, E / 2); if (rc) return rc;
  t->packed = true;
  return ADX_OK;
}

static int tp_dropout(TrajArgs* a, float p, uint64_t seed) {
  ADX_REQUIRE(p >= 0.f && p < 1.f, "trajpred: dropout probability %g outside [0, 1)", (double)p);
  a->drop_thresh = p > 0.f ? (uint32_t)((double)p * 4294967296.0) : 0u;
  a->drop_scale = 1.f / (1.f - p);
  a->seed_lo = (uint32_t)seed;
  a->seed_hi = (uint32_t)(seed >> 32);
  return ADX_OK;
}

static int tp_common(adx_trajpred* t, const void* packed, int batch, int T, TrajArgs* a) {
  ADX_REQUIRE(t && packed, "trajpred: null argument");
  if (!t->packed) {
    set_error("trajpred: weights were never packed (call adx_trajpred_pack first)");
    return ADX_ERR_STATE;
  }
  ADX_REQUIRE(batch >= 1 && T >= 1 && T < kMaxTP, "trajpred: batch %d / sequence length %d unsupported (T <= %d)", batch, T,
              kMaxTP - 1);
  memset(a, 0, sizeof(*a));
  a->P = (const float*)packed;
  a->L = t->L;
  a->B = batch;
  a->T = T;
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    const void* f32s[3] = {reinterpret_cast<const void*>(&trajpred_forward_kernel<32>),
                           reinterpret_cast<const void*>(&trajpred_backward_kernel<32>),
                           reinterpret_cast<const void*>(&guided_output_kernel<32>)};
    const void* f64s[3] = {reinterpret_cast<const void*>(&trajpred_forward_kernel<64>),
                           reinterpret_cast<const void*>(&trajpred_backward_kernel<64>),
                           reinterpret_cast<const void*>(&guided_output_kernel<64>)};
    for (const void* f : f32s)
      ADX_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_floats<32>() * sizeof(float))));
    for (const void* f : f64s)
      ADX_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_floats<64>() * sizeof(float))));
    once.commit();
  }
  if (T >= 32) {
    const size_t need = adx_trajpred_scratch_bytes(t, batch, T);
    if (t->big == nullptr || t->big_bytes < need) {
      set_error("trajpred: sequence length %d (32..63 rows) needs %zu bytes of scratch for %d samples, %zu were lent "
                "(adx_trajpred_scratch_bytes / adx_trajpred_set_scratch)", T, need, batch, t->big_bytes);
      return ADX_ERR_STATE;
    }
    a->big = t->big;
  }
  return ADX_OK;
}

// the 32-row instantiation when the sequence fits it, the 64-row one otherwise
#define ADX_TP_LAUNCH(kernel, a, batch, s)                                                                         \
  do {                                                                                                             \
    if ((a).T < 32)                                                                                                \
      kernel<32><<<dim3(batch), dim3(NT), lds_floats<32>() * sizeof(float), (s)>>>(a);                             \
    else                                                                                                           \
      kernel<64><<<dim3(batch), dim3(NT), lds_floats<64>() * sizeof(float), (s)>>>(a);                             \
    ADX_LAUNCH_CHECK();                                                                                            \
  } while (0)

int adx_trajpred_forward(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                         const float* time_embed, float* out, int32_t batch, int32_t T, adx_stream stream) {
  TrajArgs a;
  int rc = tp_common(t, packed, batch, T, &a);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(action && time_embed && out, "adx_trajpred_forward: null tensor");
  a.action = action; a.act_sb = act_sb; a.act_st = act_st; a.te = time_embed;
  a.out = out; a.out_sb = (int64_t)T * t->L.out_dim; a.out_st = t->L.out_dim;
  ADX_TP_LAUNCH(trajpred_forward_kernel, a, batch, (hipStream_t)stream);
  return ADX_OK;
}

// train-mode forward: the encoder layers' dropout (probability dropout_p, masks keyed by `seed`) is applied; the
// matching adx_trajpred_backward_params call must be given the same dropout_p and seed
int adx_trajpred_forward_train(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                               const float* time_embed, float* out, int32_t batch, int32_t T, float dropout_p,
                               uint64_t seed, adx_stream stream) {
  TrajArgs a;
  int rc = tp_common(t, packed, batch, T, &a);
  if (rc != ADX_OK) return rc;
  rc = tp_dropout(&a, dropout_p, seed);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(action && time_embed && out, "adx_trajpred_forward_train: null tensor");
  a.action = action; a.act_sb = act_sb; a.act_st = act_st; a.te = time_embed;
  a.out = out; a.out_sb = (int64_t)T * t->L.out_dim; a.out_st = t->L.out_dim;
  ADX_TP_LAUNCH(trajpred_forward_kernel, a, batch, (hipStream_t)stream);
  return ADX_OK;
}

int adx_trajpred_backward(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                          const float* time_embed, const float* grad_out, float* grad_action, int32_t batch, int32_t T,
                          adx_stream stream) {
  TrajArgs a;
  int rc = tp_common(t, packed, batch, T, &a);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(action && time_embed && grad_out && grad_action, "adx_trajpred_backward: null tensor");
  a.action = action; a.act_sb = act_sb; a.act_st = act_st; a.te = time_embed;
  a.gout = grad_out; a.gout_sb = (int64_t)T * t->L.out_dim; a.gout_st = t->L.out_dim;
  a.gact = grad_action; a.gact_sb = (int64_t)T * IN_DIM; a.gact_st = IN_DIM;
  ADX_TP_LAUNCH(trajpred_backward_kernel, a, batch, (hipStream_t)stream);
  return ADX_OK;
}

// Training: parameter gradients (gradient image in the packed layout, zeroed here, see
// adx_trajpred_param_offsets) and d(time_embed) [B][64]; grad_action may be NULL.
int adx_trajpred_backward_params(adx_trajpred* t, const void* packed, const float* action, int64_t act_sb, int64_t act_st,
                                 const float* time_embed, const float* grad_out, float* grad_action, void* grad_image,
                                 float* d_time_embed, int32_t batch, int32_t T, float dropout_p, uint64_t seed,
                                 adx_stream stream) {
  TrajArgs a;
  int rc = tp_common(t, packed, batch, T, &a);
  if (rc != ADX_OK) return rc;
  rc = tp_dropout(&a, dropout_p, seed);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(action && time_embed && grad_out && grad_image && d_time_embed, "adx_trajpred_backward_params: null tensor");
  ADX_CHECK_HIP(hipMemsetAsync(grad_image, 0, (size_t)t->L.total * sizeof(float), (hipStream_t)stream));
  a.action = action; a.act_sb = act_sb; a.act_st = act_st; a.te = time_embed;
  a.gout = grad_out; a.gout_sb = (int64_t)T * t->L.out_dim; a.gout_st = t->L.out_dim;
  a.gact = grad_action; a.gact_sb = (int64_t)T * IN_DIM; a.gact_st = IN_DIM;
  a.G = (float*)grad_image; a.dte = d_time_embed;
  ADX_TP_LAUNCH(trajpred_backward_kernel, a, batch, (hipStream_t)stream);
  return ADX_OK;
}

// float offset of every parameter (adx_trajpred_pack order) inside the packed / gradient image
int adx_trajpred_param_offsets(const adx_trajpred* t, int64_t* offsets, int32_t n) {
  ADX_REQUIRE(t && offsets && n == adx_trajpred_num_params(t), "adx_trajpred_param_offsets: bad argument");
  const TPLayout& L = t->L;
  int i = 0;
  offsets[i++] = L.w_ip; offsets[i++] = L.b_ip;
  for (int l = 0; l < NL; ++l) {
    const TPLayer& y = L.layer[l];
    offsets[i++] = y.w_in; offsets[i++] = y.b_in; offsets[i++] = y.w_out; offsets[i++] = y.b_out;
    offsets[i++] = y.w1; offsets[i++] = y.b1; offsets[i++] = y.w2; offsets[i++] = y.b2;
    offsets[i++] = y.g1; offsets[i++] = y.be1; offsets[i++] = y.g2; offsets[i++] = y.be2;
  }
  offsets[i++] = L.gf; offsets[i++] = L.bef; offsets[i++] = L.w_op; offsets[i++] = L.b_op;
  return ADX_OK;
}

int adx_guided_output(adx_trajpred* t, const void* packed, const float* action /* [B][T+1][3] */,
                      const float* time_embed, const float* target /* [B][2] */, float model_std, float scale,
                      float* x_guided /* [B][T+1][out_dim+3] */, float* loss /* [B] or NULL */, int32_t batch, int32_t T,
                      adx_stream stream) {
  TrajArgs a;
  int rc = tp_common(t, packed, batch, T, &a);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(action && time_embed && target && x_guided, "adx_guided_output: null tensor");
  a.action = action; a.act_sb = (int64_t)(T + 1) * IN_DIM; a.act_st = IN_DIM; a.te = time_embed;
  a.target = target; a.xg = x_guided; a.grad_scale = model_std; a.scale = scale; a.loss = loss;
  ADX_TP_LAUNCH(guided_output_kernel, a, batch, (hipStream_t)stream);
  return ADX_OK;
}

}  // extern "C"
