// ResNet-34 perception forward (eval mode) on the fp32 matrix cores of gfx950.
//   modeling/resnet.py:56-102 (BasicBlock), :163-296 (ResNet), fc replaced by Linear(512, dim)
//   at modeling/temporal.py:83-84.  This is 99.5 % of the FLOPs of one reference forward
//   (34 GFLOP per 3x256x900 image).
//
// conv2d = implicit GEMM, C[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (tap, cin):
//   * v_mfma_f32_32x32x2_f32: A = weights (32 couts x 2 k), B = activations (2 k x 32 pixels),
//     so a lane of the accumulator owns one pixel column -> NCHW stores are 128-byte rows and
//     the tensors keep PyTorch's layout end to end (no layout conversion of the camera image);
//   * one workgroup = 4 waves = 4 output rows x 32 output columns x 64 output channels; per
//     16-channel chunk the zero-padded input patch and the [tap][cin][64] weight slab are staged
//     in LDS (both read conflict-free: consecutive lanes = consecutive columns / channels);
//   * epilogue fuses eval-mode BatchNorm as y = acc * scale[c] + shift[c] (the same
//     alpha/beta form torch uses), the residual add and ReLU.
#include <stdlib.h>

#include <vector>

#include "adx_common.h"
#include "conv2d_internal.h"

namespace adx {

constexpr int kCoutT = 64;   // output channels per workgroup (2 MFMA row blocks)
constexpr size_t kMaxLds = 96 * 1024;
constexpr int g_conv_cc = 16;    // channels per LDS chunk of the 8-row kernel
constexpr int g_conv_rows = 2;   // rows per wave of the 3x3 stride-1 kernel

// Software pipeline (register double buffer): the global loads of chunk i+1 (input patch + weight
// slab) are issued before the MFMAs of chunk i and written to LDS after them, so HBM/L2 latency
// hides behind ~9k cycles of matrix work per 16-channel chunk.
// ROWS = output rows per wave (tile = 4*ROWS rows x 32 columns x 64 channels): ROWS = 2 reuses every
// weight fragment for two pixel rows (1.0 instead of 1.5 LDS reads per MFMA).
template <int STRIDE, int K, int ROWS, int CCH = 16>
__global__ void __launch_bounds__(256) conv2d_kernel(const Conv2dArgs a) {
  constexpr int TH = 4 * ROWS;
  constexpr int CC = (K == 7) ? 4 : CCH;                  // channels per chunk
  constexpr int PH = (TH - 1) * STRIDE + K;           // staged patch rows / columns
  constexpr int PW = (kTileW - 1) * STRIDE + K;
  constexpr int PLANE = PH * PW;
  constexpr int NP = CC * PLANE;                          // patch floats per chunk
  constexpr int PITEMS = (NP + 255) / 256;
  constexpr int NTAPS = K * K;
  constexpr int NW4 = NTAPS * CC * (kCoutT / 4);          // weight float4s per chunk
  constexpr int WITEMS = (NW4 + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                    // [CC][PH][PW]
  float* wl = smem + PITEMS * 256;                        // [NTAPS][CC][64]
  float* ss = wl + NTAPS * CC * kCoutT;                   // scale[64], shift[64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bid = blockIdx.x;
  const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int n = bid;
  const int oy0 = ty * TH, ox0 = tx * kTileW;
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
  const int cout0 = ct * kCoutT;
  const int l31 = lane & 31, khalf = lane >> 5;
  const size_t hw = (size_t)a.H * a.W;
  const float* xin = a.x + (size_t)n * a.Cin * hw;

  // per-thread gather offsets of its patch elements, identical for every chunk (-1 = zero padding)
  int goff[PITEMS];
#pragma unroll
  for (int k = 0; k < PITEMS; ++k) {
    const int e = tid + 256 * k;
    const int c = e / PLANE, rem = e - c * PLANE;
    const int py = rem / PW, px = rem - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = e < NP && c < a.Cin && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[k] = ok ? (int)(c * hw + (size_t)iy * a.W + ix) : -1;
  }
  if (tid < 2 * kCoutT) {
    const int c = cout0 + (tid & (kCoutT - 1));
    ss[tid] = a.scale == nullptr ? (tid < kCoutT ? 1.f : 0.f) : (tid < kCoutT ? a.scale[c] : a.shift[c]);
  }

  f32x16 acc[ROWS][2];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[r][0][i] = 0.f; acc[r][1][i] = 0.f; }
  float pv[PITEMS];
  f32x4 wv[WITEMS];
  auto load_chunk = [&](int c0) {
    const float* xc = xin + (size_t)c0 * hw;
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
      const float v = xc[goff[k] >= 0 ? goff[k] : 0];   // branch-free: out-of-image taps read element 0 and are zeroed
      pv[k] = goff[k] >= 0 ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < WITEMS; ++k) {
      const int e = tid + 256 * k;
      const int q = e & 15, row = e >> 4;             // row = tap * CC + c
      const int tap = row / CC, c = row - tap * CC;
      const int ec = e < NW4 ? e : 0;                    // clamp instead of branching; the store below is predicated
      const int rowc = ec >> 4, tapc = rowc / CC, cc_ = rowc - tapc * CC;
      (void)tap; (void)c;
      wv[k] = *reinterpret_cast<const f32x4*>(a.w + ((size_t)tapc * a.cin_pad + c0 + cc_) * a.Cout + cout0 + 4 * q);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) patch[tid + 256 * k] = pv[k];   // the patch area is padded to PITEMS * 256 floats
#pragma unroll
    for (int k = 0; k < WITEMS; ++k) {
      const int e = tid + 256 * k;
      if (e < NW4) *reinterpret_cast<f32x4*>(wl + 4 * e) = wv[k];
    }
  };

  load_chunk(0);
  for (int c0 = 0; c0 < a.cin_pad; c0 += CC) {
    if (c0 > 0) __syncthreads();          // everyone finished reading the previous chunk
    store_chunk();
    __syncthreads();
    if (c0 + CC < a.cin_pad) load_chunk(c0 + CC);   // in flight during the MFMAs below
#pragma unroll 1
    for (int kh = 0; kh < K; ++kh) {
#pragma unroll 1
      for (int kw = 0; kw < K; ++kw) {
        const float* pb = patch + (wave * ROWS * STRIDE + kh) * PW + l31 * STRIDE + kw + khalf * PLANE;
        const float* wa = wl + ((kh * K + kw) * CC + khalf) * kCoutT + l31;
#pragma unroll
        for (int ks = 0; ks < CC / 2; ++ks) {
          const float a0 = wa[2 * ks * kCoutT];
          const float a1 = wa[2 * ks * kCoutT + 32];
#pragma unroll
          for (int r = 0; r < ROWS; ++r) {
            const float b = pb[2 * ks * PLANE + r * STRIDE * PW];
            acc[r][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[r][0], 0, 0, 0);
            acc[r][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[r][1], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- epilogue: BN scale/shift, residual, ReLU; lane = pixel column, register = channel --------
  const int ox = ox0 + l31;
  const size_t img = (size_t)n * a.Cout * a.OH * a.OW;
  const size_t plane_o = (size_t)a.OH * a.OW;
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int oy = oy0 + wave * ROWS + rr;
    if (oy >= a.OH || ox >= a.OW) continue;
    const size_t pix = (size_t)oy * a.OW + ox;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float rv[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        rv[r] = a.res != nullptr ? a.res[img + (size_t)(cout0 + cl) * plane_o + pix] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
        float v = acc[rr][half][r];
        v = v * ss[cl] + ss[kCoutT + cl];
        v += rv[r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        a.y[img + (size_t)(cout0 + cl) * plane_o + pix] = v;
      }
    }
  }
}

// [Cout][Cin][KH][KW] -> [KH*KW][cin_pad][Cout]
__global__ void conv2d_pack_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int taps,
                                   int cin_pad, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int co = idx % Cout;
  const int ci = (idx / Cout) % cin_pad;
  const int tap = idx / ((size_t)Cout * cin_pad);
  p[idx] = ci < Cin ? w[((size_t)co * Cin + ci) * taps + tap] : 0.f;
}

// data-gradient image of the same weight: K = original cout, N = original cin, taps flipped:
// p[tap'][co][ci] = w[co][ci][taps-1-tap']
__global__ void conv2d_pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ p, int Cout, int Cin, int taps,
                                         size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int ci = idx % Cin;
  const int co = (idx / Cin) % Cout;
  const int tap = idx / ((size_t)Cin * Cout);
  p[idx] = w[((size_t)co * Cin + ci) * taps + (taps - 1 - tap)];
}

int conv2d_pack_spec(const ConvSpec& c, const float* w, float* packed, int dgrad, hipStream_t s) {
  if (conv2d_hs_eligible(c)) return conv2d_hs_pack(c, w, packed, dgrad, s);
  // dgrad: the forward weight is [c.cin][c.cout][k][k]
  return dgrad ? conv2d_pack_raw(w, packed, c.cin, c.cout, c.k, c.cin, 1, s)
               : conv2d_pack_raw(w, packed, c.cout, c.cin, c.k, c.cin_pad, 0, s);
}

int conv2d_pack_raw(const float* w, float* packed, int cout, int cin, int k, int cin_pad, int dgrad, hipStream_t s) {
  if (dgrad) {
    const size_t total = (size_t)k * k * cout * cin;
    conv2d_pack_dgrad_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w, packed, cout, cin, k * k, total);
  } else {
    const size_t total = (size_t)k * k * cin_pad * cout;
    conv2d_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w, packed, cout, cin, k * k, cin_pad, total);
  }
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// eval-mode BatchNorm2d as y = x * scale + shift (eps = 1e-5)
__global__ void bn_fold_kernel(const float* g, const float* b, const float* mean, const float* var, float* scale,
                               float* shift, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float s = g[c] / sqrtf(var[c] + 1e-5f);
  scale[c] = s;
  shift[c] = b[c] - mean[c] * s;
}

// MaxPool2d(kernel 3, stride 2, padding 1), modeling/resnet.py:197.  One wave per output row (blockIdx.x = plane),
// lanes along the row.  Even widths: a lane loads columns (2 ox, 2 ox + 1) as one 8-byte word -- every load
// instruction of a wave covers one contiguous 512-byte span, each input row is fetched exactly once -- and takes
// column 2 ox - 1 from its left neighbour's word (only the first lane of a 64-column chunk loads it itself).
__device__ __forceinline__ float pool_max(float m, float v) { return (v > m || v != v) ? v : m; }   // NaN propagates like torch

__global__ void __launch_bounds__(256) maxpool_kernel(const float* __restrict__ x, float* __restrict__ y, int planes,
                                                       int H, int W, int OH, int OW) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int oy = blockIdx.y * 4 + wave;
  const int pl = blockIdx.x;
  if (oy >= OH) return;
  const float* src = x + (size_t)pl * H * W;
  float* dst = y + ((size_t)pl * OH + oy) * OW;
  if ((W & 1) == 0 && ((size_t)x & 7) == 0) {
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    for (int ox0 = lane; ox0 - lane < OW; ox0 += 256) {   // wave-uniform trip count: the shuffles need every lane
      float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int iy = oy * 2 - 1 + dy;
        if (iy < 0 || iy >= H) continue;               // wave-uniform
        const float* row = src + (size_t)iy * W;
        f32x2 v[4];
        float left[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int ox = ox0 + 64 * q;
          const bool in = 2 * ox + 1 < W;              // W even: both columns exist or neither
          v[q] = in ? *reinterpret_cast<const f32x2*>(row + 2 * ox) : f32x2{-INFINITY, -INFINITY};
          left[q] = (lane == 0 && ox > 0 && 2 * ox - 1 < W) ? row[2 * ox - 1] : -INFINITY;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float nb = __shfl_up(v[q][1], 1, 64);
          const float l = lane == 0 ? left[q] : nb;
          m[q] = pool_max(pool_max(pool_max(m[q], l), v[q][0]), v[q][1]);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (ox0 + 64 * q < OW) dst[ox0 + 64 * q] = m[q];
    }
    return;
  }
  // generic widths: four 64-column chunks per pass, 36 independent loads in flight per lane
  for (int ox0 = lane; ox0 < OW; ox0 += 256) {
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
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
          const bool ok = rok && ix >= 0 && ix < W;
          const float v = ok ? row[ix] : -INFINITY;
          m[q] = pool_max(m[q], v);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (ox0 + 64 * q < OW) dst[ox0 + 64 * q] = m[q];
  }
}

// AdaptiveAvgPool2d(1) + flatten + fc, modeling/resnet.py:288-290; one 16-wave workgroup per image, four channel
// planes in flight per wave (the means are plain sequential-lane sums: deterministic)
__global__ void __launch_bounds__(1024) avgpool_fc_kernel(const float* __restrict__ x, const float* __restrict__ fw,
                                                           const float* __restrict__ fb, float* __restrict__ out,
                                                           int C, int HW, int out_dim) {
  __shared__ float pooled[512];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* src = x + (size_t)n * C * HW;
  const float inv = 1.0f / (float)HW;
  for (int c0 = wave * 4; c0 < C; c0 += 64) {
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < HW; i += 64) {
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += (c0 + q < C) ? src[(size_t)(c0 + q) * HW + i] : 0.f;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float t = wave_sum(s[q]);
      if (lane == 0 && c0 + q < C) pooled[c0 + q] = t * inv;
    }
  }
  __syncthreads();
  for (int j = wave; j < out_dim; j += 16) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += fw[(size_t)j * C + c] * pooled[c];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)n * out_dim + j] = s + fb[j];
  }
}

int maxpool_launch(const float* x, float* y, int planes, int H, int W, int OH, int OW, hipStream_t s) {
  maxpool_kernel<<<dim3(planes, ceil_div(OH, 4)), dim3(256), 0, s>>>(x, y, planes, H, W, OH, OW);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// the same on a map in the cell layout (conv2d_hs.hip: per image [C / 8][hi, lo][HW] cells of eight fp16 channels): one cell
// group per wave at a time, value = hi + lo / 2^11, the same lane-strided partial sums and wave reduction per channel
__global__ void __launch_bounds__(1024) avgpool_fc_cells_kernel(const uint4* __restrict__ x, const float* __restrict__ fw,
                                                                 const float* __restrict__ fb, float* __restrict__ out,
                                                                 int C, int HW, int out_dim) {
  typedef _Float16 h8 __attribute__((ext_vector_type(8)));
  __shared__ float pooled[512];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const uint4* src = x + (size_t)n * (C / 8) * 2 * HW;
  const float inv = 1.0f / (float)HW;
  for (int g = wave; g < C / 8; g += 16) {
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = lane; i < HW; i += 64) {
      const h8 hi = __builtin_bit_cast(h8, src[(size_t)(2 * g) * HW + i]);
      const h8 lo = __builtin_bit_cast(h8, src[(size_t)(2 * g + 1) * HW + i]);
#pragma unroll
      for (int q = 0; q < 8; ++q) s[q] += (float)hi[q] + (float)lo[q] * (1.f / 2048.f);
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float t = wave_sum(s[q]);
      if (lane == 0) pooled[8 * g + q] = t * inv;
    }
  }
  __syncthreads();
  for (int j = wave; j < out_dim; j += 16) {
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += fw[(size_t)j * C + c] * pooled[c];
    s = wave_sum(s);
    if (lane == 0) out[(size_t)n * out_dim + j] = s + fb[j];
  }
}

int avgpool_fc_launch(const float* x, const float* fw, const float* fb, float* out, int batch, int C, int HW, int out_dim,
                      hipStream_t s, int x_cells) {
  ADX_REQUIRE(C <= 512, "avgpool_fc: at most 512 channels");
  if (x_cells) {
    ADX_REQUIRE(C % 8 == 0, "avgpool_fc: the cell layout holds channels in groups of eight");
    avgpool_fc_cells_kernel<<<dim3(batch), dim3(1024), 0, s>>>(reinterpret_cast<const uint4*>(x), fw, fb, out, C, HW, out_dim);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  avgpool_fc_kernel<<<dim3(batch), dim3(1024), 0, s>>>(x, fw, fb, out, C, HW, out_dim);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

namespace adx {

static size_t align64f(size_t v) { return (v + 63) / 64 * 64; }

static int conv_out(int h, int k, int s, int p) { return conv_out_dim(h, k, s, p); }

static int conv2d_launch(const ConvSpec& L, const float* base, const float* x, const float* res, float* y, int N, int H,
                         int W, int relu, hipStream_t s, int fmt = 0) {
  return conv2d_launch_raw(L, x, base + L.o_w, base + L.o_scale, base + L.o_shift, res, y, N, H, W, relu, s, nullptr, 0, nullptr, 0,
                           nullptr, fmt);
}

static thread_local float* t_split_scratch = nullptr;
static thread_local size_t t_split_floats = 0;
void conv2d_set_split_scratch(float* p, size_t floats) { t_split_scratch = p; t_split_floats = p != nullptr ? floats : 0; }

int conv2d_launch_raw(const ConvSpec& L, const float* x, const float* w, const float* scale, const float* shift,
                      const float* res, float* y, int N, int H, int W, int relu, hipStream_t s, const uint32_t* x_amax,
                      int x_amax_n, float* stats_part, size_t stats_floats, int* stats_p, int fmt, const BnBwdStats* bst) {
  if (stats_p != nullptr) *stats_p = 0;
  Conv2dArgs a;
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.res = res; a.y = y; a.x_amax = x_amax; a.x_amax_n = x_amax_n;
  a.w_ds = nullptr; a.scale_ds = nullptr; a.shift_ds = nullptr; a.y_ds = nullptr;
  a.N = N; a.Cin = L.cin; a.H = H; a.W = W; a.Cout = L.cout;
  a.OH = conv_out(H, L.k, L.stride, L.pad); a.OW = conv_out(W, L.k, L.stride, L.pad);
  a.KH = L.k; a.KW = L.k; a.stride = L.stride; a.pad = L.pad; a.relu = relu;
  a.cin_pad = L.cin_pad; a.cc = L.cc;
  a.x_u8 = nullptr;
  a.d2s_cin = 0; a.d2s_h = 0; a.d2s_w = 0; a.stem_seg_tiles = 0; a.stem_nseg = 1;
  a.ksplit = 1; a.cper = 0; a.part = t_split_scratch; a.part_stride = t_split_floats;   // part_stride: capacity until the launch fixes it
  a.stats_part = nullptr; a.stats_p = 0;
  a.bs_raw = a.bs_out = a.bs_mean = a.bs_rstd = a.bs_gamma = a.bs_beta = nullptr; a.bs_mask = 0; a.bs_bits = nullptr; a.res_bits = nullptr;
  if (bst != nullptr) {
    ADX_REQUIRE(bst->raw && bst->mean && bst->rstd && bst->gamma && bst->beta && (bst->mask == 2 || (bst->mask == 1 && (bst->out || bst->bits))),
                "conv2d: incomplete BatchNorm-backward statistics request");
    ADX_REQUIRE(scale == nullptr && relu == 0, "conv2d: BatchNorm-backward statistics belong to data-gradient launches");
    a.bs_raw = bst->raw; a.bs_out = bst->out; a.bs_mean = bst->mean; a.bs_rstd = bst->rstd; a.bs_gamma = bst->gamma;
    a.bs_beta = bst->beta; a.bs_mask = bst->mask; a.bs_bits = bst->mask == 1 ? bst->bits : nullptr;
    a.res_bits = bst->res_bits;
    ADX_REQUIRE(a.res_bits == nullptr || (res != nullptr && stats_part != nullptr &&
                                          conv2d_hs3x3_dgrad_stats(L, N, H, W, (fmt & kFmtXCells) != 0, stats_floats)),
                "conv2d: a masked residual belongs to a data-gradient launch with the statistics epilogue");
  }
  a.x_cells = (fmt & kFmtXCells) != 0; a.y_cells = (fmt & kFmtYCells) != 0; a.res_cells = (fmt & kFmtResCells) != 0 && res != nullptr;
  // cells: plain launches of the pipelined 3x3 kernel (the inference executor), or the INPUT of a training-forward launch
  const bool train_cells = fmt == kFmtXCells && stats_part != nullptr && stats_p != nullptr && bst == nullptr && x_amax == nullptr &&
                           conv2d_hs3x3_train_cells(L, N, H, W, stats_floats);
  // ... or the input of a data-gradient launch: a gradient written as cells under a known power-of-two scale
  const bool dgrad_cells = fmt == (kFmtXCells | kFmtXScaled) && L.dgrad && x_amax != nullptr && x_amax_n < 0 &&
                           conv2d_hs3x3_dgrad_cells(L, N, H, W);
  if (dgrad_cells) fmt = kFmtXCells;
  ADX_REQUIRE(fmt == 0 || train_cells || dgrad_cells || (stats_part == nullptr && conv2d_hs3x3_plain(L, N, H, W)),
              "conv2d: the cell layout belongs to plain launches of the pipelined 3x3 kernel (%d -> %d, k%d s%d)", L.cin, L.cout, L.k, L.stride);
  if (conv2d_hs_eligible(L)) {
    if (stats_part != nullptr && stats_p != nullptr && conv2d_hs_stats_tiles(L, a) > 0 &&
        (size_t)conv2d_hs_stats_tiles(L, a) * (L.cout * 2 + L.cout / 64) <= stats_floats) {     // (+ the data gradient's max |dz| per slab)
      a.stats_part = stats_part;
      a.stats_p = conv2d_hs_stats_tiles(L, a);
      *stats_p = a.stats_p;
    }
    return conv2d_hs_launch(L, a, s);
  }
  const int rows = (L.stride == 1 && L.k == 3 && g_conv_rows == 2) ? 2 : 1;   // 8-row tiles for the 3x3 stride-1 convs
  const int th = 4 * rows;
  a.tiles_x = ceil_div(a.OW, kTileW); a.tiles_y = ceil_div(a.OH, th); a.cout_tiles = L.cout / kCoutT;
  a.PH = (th - 1) * L.stride + L.k; a.PW = (kTileW - 1) * L.stride + L.k;
  a.PWp = a.PW;
  ADX_REQUIRE(L.cin % L.cc == 0 || L.cin < L.cc, "conv2d: cin %d must be < %d or a multiple of it", L.cin, L.cc);
  ADX_REQUIRE((size_t)L.cin * H * W < (1u << 31), "conv2d: image plane too large for 32-bit gather offsets");
  const int cch = (rows == 2 && g_conv_cc == 8) ? 8 : a.cc;   // channels staged per chunk
  const size_t np = (size_t)cch * a.PH * a.PW;
  const size_t lds = sizeof(float) * ((np + 255) / 256 * 256 + (size_t)L.k * L.k * cch * kCoutT + 2 * kCoutT);
  ADX_REQUIRE(lds <= kMaxLds, "conv2d: LDS %zu bytes too large", lds);
  const size_t grid = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y * N;
  ADX_REQUIRE(grid < (1u << 31), "conv2d: grid too large");
  static std::atomic<uint64_t> attr_set{0};  // dynamic LDS above 64 KB must be opted into once per kernel
  if (DeviceOnce once{attr_set}; once) {
    const void* fns[6] = {reinterpret_cast<const void*>(&conv2d_kernel<1, 3, 1>), reinterpret_cast<const void*>(&conv2d_kernel<2, 3, 1>),
                          reinterpret_cast<const void*>(&conv2d_kernel<2, 1, 1>), reinterpret_cast<const void*>(&conv2d_kernel<2, 7, 1>),
                          reinterpret_cast<const void*>(&conv2d_kernel<1, 1, 1>), reinterpret_cast<const void*>(&conv2d_kernel<1, 3, 2>)};
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_kernel<1, 3, 2, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    for (const void* f : fns) ADX_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxLds));
    once.commit();
  }
  const dim3 g((unsigned)grid), blk(256);
  if (L.stride == 1 && L.k == 3 && rows == 2 && cch == 8) conv2d_kernel<1, 3, 2, 8><<<g, blk, lds, s>>>(a);
  else if (L.stride == 1 && L.k == 3 && rows == 2) conv2d_kernel<1, 3, 2><<<g, blk, lds, s>>>(a);
  else if (L.stride == 1 && L.k == 3) conv2d_kernel<1, 3, 1><<<g, blk, lds, s>>>(a);
  else if (L.stride == 2 && L.k == 3) conv2d_kernel<2, 3, 1><<<g, blk, lds, s>>>(a);
  else if (L.stride == 2 && L.k == 1) conv2d_kernel<2, 1, 1><<<g, blk, lds, s>>>(a);
  else if (L.stride == 2 && L.k == 7) conv2d_kernel<2, 7, 1><<<g, blk, lds, s>>>(a);
  else if (L.stride == 1 && L.k == 1) conv2d_kernel<1, 1, 1><<<g, blk, lds, s>>>(a);
  else {
    set_error("conv2d: no kernel for k=%d stride=%d (ResNet-34 uses 3x3 s1, 3x3 s2, 1x1 s2, 7x7 s2)", L.k, L.stride);
    return ADX_ERR_INVALID;
  }
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx

using namespace adx;

static int spec_from_desc(const adx_conv2d_desc* d, ConvSpec* L) {
  ADX_REQUIRE(d != nullptr, "conv2d: null descriptor");
  ADX_REQUIRE(d->cin >= 1 && d->cout >= 64 && d->cout % kCoutT == 0, "conv2d: cout %d must be a multiple of %d", d->cout,
              kCoutT);
  ADX_REQUIRE(d->k >= 1 && d->k <= 7 && (d->stride == 1 || d->stride == 2) && d->pad >= 0 && d->pad <= 3,
              "conv2d: unsupported k/stride/pad %d/%d/%d", d->k, d->stride, d->pad);
  memset(L, 0, sizeof(*L));
  L->cin = d->cin; L->cout = d->cout; L->k = d->k; L->stride = d->stride; L->pad = d->pad;
  L->cc = d->cin >= 16 ? 16 : 4;
  L->cin_pad = round_up(d->cin, L->cc);
  return ADX_OK;
}

extern "C" {

size_t adx_conv2d_packed_bytes(const adx_conv2d_desc* d) {
  ConvSpec L;
  if (spec_from_desc(d, &L) != ADX_OK) return 0;
  return sizeof(float) * (conv2d_hs_eligible(L) ? conv2d_packed_floats(L) : (size_t)L.k * L.k * L.cin_pad * L.cout);
}

int adx_conv2d_pack(const adx_conv2d_desc* d, const float* w, float* packed, adx_stream stream) {
  ConvSpec L;
  int rc = spec_from_desc(d, &L);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(w && packed, "adx_conv2d_pack: null pointer");
  return conv2d_pack_spec(L, w, packed, 0, (hipStream_t)stream);
}

int adx_conv2d_forward(const adx_conv2d_desc* d, const float* x, const float* packed_w, const float* scale,
                       const float* shift, const float* res, float* y, int32_t n, int32_t h, int32_t w, int32_t relu,
                       adx_stream stream) {
  ConvSpec L;
  int rc = spec_from_desc(d, &L);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(x && packed_w && y, "adx_conv2d_forward: null tensor");
  ADX_REQUIRE((scale == nullptr) == (shift == nullptr), "adx_conv2d_forward: scale and shift go together");
  ADX_REQUIRE(n >= 1 && h + 2 * d->pad >= d->k && w + 2 * d->pad >= d->k, "adx_conv2d_forward: input too small");
  return conv2d_launch_raw(L, x, packed_w, scale, shift, res, y, n, h, w, relu, (hipStream_t)stream);
}

int adx_conv2d_cells_supported(const adx_conv2d_desc* d, int32_t n, int32_t h, int32_t w) {
  ConvSpec L;
  if (d == nullptr || spec_from_desc(d, &L) != ADX_OK) return 0;
  return conv2d_hs3x3_plain(L, n, h, w) ? 1 : 0;
}

int adx_conv2d_forward_cells(const adx_conv2d_desc* d, const void* x, const float* packed_w, const float* scale,
                             const float* shift, const void* res, void* y, int32_t n, int32_t h, int32_t w, int32_t relu,
                             int32_t fmt, adx_stream stream) {
  ConvSpec L;
  int rc = spec_from_desc(d, &L);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(x && packed_w && y, "adx_conv2d_forward_cells: null tensor");
  ADX_REQUIRE((scale == nullptr) == (shift == nullptr), "adx_conv2d_forward_cells: scale and shift go together");
  ADX_REQUIRE(fmt >= 0 && fmt <= 7 && (fmt & kFmtYCells), "adx_conv2d_forward_cells: fmt %d (the output is a cell tensor)", fmt);
  ADX_REQUIRE(conv2d_hs3x3_plain(L, n, h, w), "adx_conv2d_forward_cells: not a plain launch of the pipelined 3x3 kernel "
              "(adx_conv2d_cells_supported)");
  return conv2d_launch_raw(L, (const float*)x, packed_w, scale, shift, (const float*)res, (float*)y, n, h, w, relu,
                           (hipStream_t)stream, nullptr, 0, nullptr, 0, nullptr, fmt);
}

int adx_resnet_create(int32_t out_dim, adx_resnet** out) {
  ADX_REQUIRE(out != nullptr && out_dim >= 1 && out_dim <= 4096, "adx_resnet_create: bad argument");
  adx_resnet* r = new adx_resnet();
  r->out_dim = out_dim;
  int t = 0;
  size_t off = 0;
  auto add = [&](int cin, int cout, int k, int stride, int pad) {
    ConvSpec L{};
    L.cin = cin; L.cout = cout; L.k = k; L.stride = stride; L.pad = pad;
    L.t_w = t++; L.t_g = t++; L.t_b = t++; L.t_m = t++; L.t_v = t++;
    L.cc = cin >= 16 ? 16 : 4;
    L.cin_pad = round_up(cin, L.cc);
    L.fuse_with = -1;
    L.o_w = off; off = align64f(off + (conv2d_hs_eligible(L) ? conv2d_packed_floats(L) : (size_t)k * k * L.cin_pad * cout));
    L.o_scale = off; off = align64f(off + cout);
    L.o_shift = off; off = align64f(off + cout);
    r->convs.push_back(L);
  };
  add(3, 64, 7, 2, 3);
  const int planes[4] = {64, 128, 256, 512}, nblk[4] = {3, 4, 6, 3};
  int inpl = 64;
  for (int li = 0; li < 4; ++li)
    for (int bi = 0; bi < nblk[li]; ++bi) {
      const int stride = (li > 0 && bi == 0) ? 2 : 1;
      add(inpl, planes[li], 3, stride, 1);
      add(planes[li], planes[li], 3, 1, 1);
      const int ds = (stride != 1 || inpl != planes[li]) ? 1 : 0;
      if (ds) {
        add(inpl, planes[li], 1, stride, 0);
        r->convs.back().fuse_with = (int)r->convs.size() - 3;   // its block's conv1
      }
      r->block_has_ds.push_back(ds);
      inpl = planes[li];
    }
  r->t_fcw = t++; r->t_fcb = t++;
  r->n_tensors = t;
  r->o_fcw = off; off = align64f(off + (size_t)out_dim * 512);
  r->o_fcb = off; off = align64f(off + out_dim);
  r->packed_floats = off;
  *out = r;
  return ADX_OK;
}

void adx_resnet_destroy(adx_resnet* r) {
  if (r == nullptr) return;
  for (auto& st : r->side) if (st != nullptr) (void)hipStreamDestroy(st);
  if (r->ev_fork != nullptr) (void)hipEventDestroy(r->ev_fork);
  for (auto& ev : r->ev_join) if (ev != nullptr) (void)hipEventDestroy(ev);
  delete r;
}
int adx_resnet_num_tensors(const adx_resnet* r) { return r ? r->n_tensors : 0; }
size_t adx_resnet_packed_bytes(const adx_resnet* r) { return r ? r->packed_floats * sizeof(float) : 0; }

int adx_resnet_pack(adx_resnet* r, const float* const* T, int32_t n, void* packed, adx_stream stream) {
  ADX_REQUIRE(r && T && packed, "adx_resnet_pack: null argument");
  ADX_REQUIRE(n == r->n_tensors, "adx_resnet_pack: expected %d tensors, got %d", r->n_tensors, n);
  for (int i = 0; i < n; ++i) ADX_REQUIRE(T[i] != nullptr, "adx_resnet_pack: tensor %d is null", i);
  hipStream_t s = (hipStream_t)stream;
  float* base = (float*)packed;
  for (const ConvSpec& L : r->convs) {
    // a downsample conv that rides on its block's split-fp16 conv1 is packed in that kernel's layout
    const bool fused = L.fuse_with >= 0 && resnet_fuses_ds(r->convs[L.fuse_with], L);
    int rc = fused ? conv2d_hs_pack(L, T[L.t_w], base + L.o_w, 0, s) : conv2d_pack_spec(L, T[L.t_w], base + L.o_w, 0, s);
    if (rc != ADX_OK) return rc;
    bn_fold_kernel<<<dim3(ceil_div(L.cout, 256)), dim3(256), 0, s>>>(T[L.t_g], T[L.t_b], T[L.t_m], T[L.t_v],
                                                                     base + L.o_scale, base + L.o_shift, L.cout);
    ADX_LAUNCH_CHECK();
  }
  ADX_CHECK_HIP(hipMemcpyAsync(base + r->o_fcw, T[r->t_fcw], (size_t)r->out_dim * 512 * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
  ADX_CHECK_HIP(hipMemcpyAsync(base + r->o_fcb, T[r->t_fcb], (size_t)r->out_dim * sizeof(float),
                               hipMemcpyDeviceToDevice, s));
  r->packed_once = true;
  return ADX_OK;
}

// workspace: the stem output, then three rotating buffers sized for the post-maxpool map
static void resnet_dims(int h, int w, int* h1, int* w1, int* h2, int* w2) {
  *h1 = conv_out(h, 7, 2, 3); *w1 = conv_out(w, 7, 2, 3);
  *h2 = conv_out(*h1, 3, 2, 1); *w2 = conv_out(*w1, 3, 2, 1);
}

size_t adx_resnet_workspace_bytes(const adx_resnet* r, int32_t batch, int32_t h, int32_t w) {
  if (!r || batch < 1 || h < 32 || w < 32) return 0;
  int h1, w1, h2, w2;
  resnet_dims(h, w, &h1, &w1, &h2, &w2);
  const size_t stem = align64f((size_t)batch * 64 * h1 * w1);
  const size_t act = align64f((size_t)batch * 64 * h2 * w2);
  return (stem + 3 * act) * sizeof(float);
}

static int resnet_forward_impl(adx_resnet* r, const void* packed, void* workspace, const float* img,
                               const uint8_t* frames_u8, const float* mean, const float* stdv, int32_t batch, int32_t h,
                               int32_t w, float* feature, adx_stream stream);

int adx_resnet_forward(adx_resnet* r, const void* packed, void* workspace, const float* img, int32_t batch, int32_t h,
                       int32_t w, float* feature, adx_stream stream) {
  ADX_REQUIRE(img != nullptr, "adx_resnet_forward: null image");
  return resnet_forward_impl(r, packed, workspace, img, nullptr, nullptr, nullptr, batch, h, w, feature, stream);
}

int adx_resnet_forward_u8(adx_resnet* r, const void* packed, void* workspace, const uint8_t* frames_hwc, const float* mean,
                          const float* stdv, int32_t batch, int32_t h, int32_t w, float* feature, adx_stream stream) {
  ADX_REQUIRE(frames_hwc && mean && stdv, "adx_resnet_forward_u8: null frames / mean / std");
  ADX_REQUIRE(r != nullptr && !r->convs.empty() && conv2d_hs_eligible(r->convs[0]),
              "adx_resnet_forward_u8: the uint8 front-end lives in the split-fp16 stem kernel (unavailable with ADX_CONV_EXACT=1)");
  return resnet_forward_impl(r, packed, workspace, nullptr, frames_hwc, mean, stdv, batch, h, w, feature, stream);
}

// ADX_CHECK_RANGE=1 (debug): the split-fp16 kernels hold activations as fp16 hi / lo pairs, so a forward activation of
// |x| >= 65504 becomes inf and travels on silently ("Range").  In this mode every activation tensor of
// the perception pass is scanned right after the launch that wrote it and the call fails with the FIRST tensor that left
// the range (fp32 tensors: |x| >= 65504 or non-finite; cell tensors: a half with an all-ones exponent).  It synchronises
// the stream after every launch (no graph capture) and owns one device word.
__global__ void __launch_bounds__(256) range_scan_kernel(const uint32_t* __restrict__ v, size_t words, int cells, unsigned* flag) {
  bool bad = false;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < words; i += (size_t)gridDim.x * 256) {
    const uint32_t b = v[i];
    if (cells) bad |= ((b & 0x7C00u) == 0x7C00u) | ((b & 0x7C000000u) == 0x7C000000u);
    else bad |= (b & 0x7FFFFFFFu) >= 0x477FE000u;            // 65504.0f
  }
  if (bad) atomicOr(flag, 1u);
}

extern "C++" int adx::conv2d_range_check(const char* what, int index, const float* t, size_t floats, bool cells, hipStream_t s) {
  if (!debug_switches().check_range) return ADX_OK;
  static unsigned* flag = nullptr;
  if (flag == nullptr) ADX_CHECK_HIP(hipMalloc(&flag, sizeof(unsigned)));
  ADX_CHECK_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), s));
  range_scan_kernel<<<dim3(1024), dim3(256), 0, s>>>(reinterpret_cast<const uint32_t*>(t), floats, cells ? 1 : 0, flag);
  ADX_LAUNCH_CHECK();
  unsigned h = 0;
  ADX_CHECK_HIP(hipMemcpyAsync(&h, flag, sizeof(unsigned), hipMemcpyDeviceToHost, s));
  ADX_CHECK_HIP(hipStreamSynchronize(s));
  if (h != 0) {
    set_error("adx_resnet_forward: %s %d writes activations outside the fp16 range of the split kernels (|x| >= 65504 or "
              "non-finite); ADX_CONV_EXACT=1 runs the pass on the exact-fp32 kernels", what, index);
    return ADX_ERR_RANGE;
  }
  return ADX_OK;
}

static int resnet_forward_impl(adx_resnet* r, const void* packed, void* workspace, const float* img,
                               const uint8_t* frames_u8, const float* mean, const float* stdv, int32_t batch, int32_t h,
                               int32_t w, float* feature, adx_stream stream) {
  ADX_REQUIRE(r && packed && workspace && (img || frames_u8) && feature, "adx_resnet_forward: null argument");
  if (!r->packed_once) {
    set_error("adx_resnet_forward: weights were never packed (call adx_resnet_pack first)");
    return ADX_ERR_STATE;
  }
  ADX_REQUIRE(batch >= 1 && h >= 32 && w >= 32, "adx_resnet_forward: image %dx%d (batch %d) too small", h, w, batch);
  hipStream_t s = (hipStream_t)stream;
  const float* base = (const float*)packed;
  int h1, w1, h2, w2;
  resnet_dims(h, w, &h1, &w1, &h2, &w2);
  float* ws = (float*)workspace;
  float* stem = ws;
  const size_t act = align64f((size_t)batch * 64 * h2 * w2);
  float* buf[3];
  buf[0] = ws + align64f((size_t)batch * 64 * h1 * w1);
  buf[1] = buf[0] + act;
  buf[2] = buf[1] + act;

  struct ScratchScope {     // split-reduction scratch of the 3x3 convs of THIS call
    void set(float* p, size_t n) { conv2d_set_split_scratch(p, n); }
    ~ScratchScope() { conv2d_set_split_scratch(nullptr, 0); }
  } scratch_scope;
  const size_t nblocks = r->block_has_ds.size();

  struct Cursor { size_t ci; int cur, H, W; bool cells; };
  constexpr int fuse = 1;         // stem + max-pool as one launch (0: two launches, the round-1 form)

  // stem (+ pool) of images [n0, n0 + n)
  auto run_stem = [&](Cursor& st, int n0, int n, int nf, hipStream_t s) -> int {
    const ConvSpec& c0 = r->convs[0];
    const float* im = img != nullptr ? img + (size_t)n0 * 3 * h * w : nullptr;
    const uint8_t* fr = frames_u8 != nullptr ? frames_u8 + (size_t)n0 * 3 * h * w : nullptr;
    float* pooled = buf[0] + (size_t)n0 * 64 * h2 * w2;
    bool pooled_cells = false;
    int rc;
    if ((fuse || frames_u8 != nullptr) && conv2d_hs_eligible(c0)) {
      // (the unpooled stem map is never written on this path: its region is the split-reduction scratch of this sub-batch's convs)
      // the pooled map's readers are layer1's first block: conv1 (input) and conv2 (residual); cells if both read cells
      pooled_cells = nblocks > 0 && !r->block_has_ds[0] && r->convs[1].stride == 1 && conv2d_hs3x3_plain(r->convs[1], nf, h2, w2) &&
                     conv2d_hs3x3_plain(r->convs[2], nf, h2, w2);
      rc = conv2d_hs_stem_pool(c0, im, base + c0.o_w, base + c0.o_scale, base + c0.o_shift, pooled, n, h, w, s, fr, mean, stdv,
                               pooled_cells);
      if (rc != ADX_OK) return rc;
    } else {
      float* sm = stem + (size_t)n0 * 64 * h1 * w1;
      rc = conv2d_launch(c0, base, im, nullptr, sm, n, h, w, 1, s);
      if (rc != ADX_OK) return rc;
      rc = maxpool_launch(sm, pooled, n * 64, h1, w1, h2, w2, s);
      if (rc != ADX_OK) return rc;
    }
    st = Cursor{1, 0, h2, w2, pooled_cells};
    return conv2d_range_check("the stem (conv1 + bn1 + relu + maxpool), tensor", 0, pooled, (size_t)n * 64 * h2 * w2, pooled_cells, s);
  };

  // Activation formats (conv2d_hs.hip: XCELLS).  Where a layer's 3x3 stride-1 convs run as plain launches of the pipelined
  // kernel, everything from the first one's output to the last one's is a CELL tensor (same bytes, same buffers): the next
  // layer's fused stride-2 launch reads AND writes cells (both of its outputs), the fused stem + pool launch writes them and
  // the average pool reads them.  What stays fp32 NCHW: all of a layer whose launches split their reduction (small batches)
  // -- and the pooled stem map then.
  // One BasicBlock for images [n0, n0 + n); nf = the batch the format decisions are made for.
  auto run_block = [&](size_t b, Cursor& st, int n0, int n, int nf, hipStream_t s) -> int {
    size_t ci = st.ci;
    const int cur = st.cur, H = st.H, W = st.W;
    const bool cur_cells = st.cells;
    const ConvSpec& c1 = r->convs[ci++];
    const ConvSpec& c2 = r->convs[ci++];
    const int mid = (cur + 1) % 3, outb = (cur + 2) % 3;
    const int OH = conv_out(H, 3, c1.stride, 1), OW = conv_out(W, 3, c1.stride, 1);
    // a sub-batch owns the SAME region of every rotating buffer at every layer (its first image's slot of the largest map, the
    // pooled one; the images of a layer are packed from there): sub-batches run on streams of their own and are at different
    // layers at the same time, so a region that moved with the layer's per-image size would overlap another sub-batch's
    const size_t off_in = (size_t)n0 * 64 * h2 * w2, off_out = off_in;
    const float* xin = buf[cur] + off_in;
    const float* identity = xin;
    bool id_cells = cur_cells, mid_cells = false;
    const bool c2_plain = conv2d_hs3x3_plain(c2, nf, OH, OW);
    int rc;
    if (r->block_has_ds[b] && resnet_fuses_ds(c1, r->convs[ci + 0])) {
      const ConvSpec& ds = r->convs[ci++];
      mid_cells = c2_plain;            // conv2 reads conv1's output and the downsample's (its residual): cells if it can
      rc = conv2d_hs_launch_block_s2(c1, ds, xin, base + c1.o_w, base + c1.o_scale, base + c1.o_shift, buf[mid] + off_out,
                                     base + ds.o_w, base + ds.o_scale, base + ds.o_shift, buf[outb] + off_out, n, H, W, s, cur_cells,
                                     mid_cells);
      if (rc != ADX_OK) return rc;
      identity = buf[outb] + off_out;
      id_cells = mid_cells;
    } else {
      const bool c1_plain = c1.stride == 1 && conv2d_hs3x3_plain(c1, nf, H, W);
      ADX_REQUIRE(!cur_cells || c1_plain, "adx_resnet_forward: internal error (cell-layout input of a launch that cannot read it)");
      mid_cells = c1_plain && c2_plain && !r->block_has_ds[b];
      rc = conv2d_launch(c1, base, xin, nullptr, buf[mid] + off_out, n, H, W, 1, s,            // conv1 + bn1 + relu
                         (cur_cells ? kFmtXCells : 0) | (mid_cells ? kFmtYCells : 0));
      if (rc != ADX_OK) return rc;
      if (r->block_has_ds[b]) {
        const ConvSpec& ds = r->convs[ci++];
        ADX_REQUIRE(!cur_cells, "adx_resnet_forward: internal error (cell-layout input of the downsample conv)");
        rc = conv2d_launch(ds, base, xin, nullptr, buf[outb] + off_out, n, H, W, 0, s);  // downsample conv + bn
        if (rc != ADX_OK) return rc;
        identity = buf[outb] + off_out;
        id_cells = false;
      }
    }
    // conv2 + bn2 + identity + relu.  With a downsample the identity lives in buf[outb] and the result
    // overwrites buf[cur] (the block input is dead by then); otherwise the result goes to buf[outb].
    float* dst = (r->block_has_ds[b] ? buf[cur] : buf[outb]) + off_out;
    // the block output is a cell tensor when conv2 is a plain launch and whoever reads it reads cells: the next block's plain
    // conv1 + conv2 (as input and as residual), the next layer's fused stride-2 launch, or the average pool
    bool out_cells = false;
    if (c2_plain) {
      if (b + 1 == nblocks) {
        out_cells = true;
      } else if (r->block_has_ds[b + 1]) {
        out_cells = resnet_fuses_ds(r->convs[ci], r->convs[ci + 2]);
      } else {
        out_cells = r->convs[ci].stride == 1 && conv2d_hs3x3_plain(r->convs[ci], nf, OH, OW) &&
                    conv2d_hs3x3_plain(r->convs[ci + 1], nf, OH, OW);
      }
    }
    ADX_REQUIRE(out_cells || !(mid_cells || id_cells),
                "adx_resnet_forward: internal error (a conv with cell-layout operands whose reader wants fp32)");
    rc = conv2d_range_check("conv1 + bn1 + relu of BasicBlock", (int)b, buf[mid] + off_out, (size_t)n * c1.cout * OH * OW, mid_cells, s);
    if (rc != ADX_OK) return rc;
    rc = conv2d_launch(c2, base, buf[mid] + off_out, identity, dst, n, OH, OW, 1, s,
                       (mid_cells ? kFmtXCells : 0) | (out_cells ? kFmtYCells : 0) | (id_cells ? kFmtResCells : 0));
    if (rc != ADX_OK) return rc;
    rc = conv2d_range_check("the output of BasicBlock", (int)b, dst, (size_t)n * c2.cout * OH * OW, out_cells, s);
    if (rc != ADX_OK) return rc;
    st = Cursor{ci, r->block_has_ds[b] ? cur : outb, OH, OW, out_cells};
    return ADX_OK;
  };

  // (Running stem, layer1 and layer2's first block in batch chunks of 16, so that a 59 MB activation is still in the 256 MB
  // Infinity Cache when the next launch reads it, was measured: layer1's convs already find most of their input there at
  // B = 64 -- inside a pass they take 0.20 ms where the same launch repeated on fixed buffers takes 0.25 -- and the chunked
  // stem launch has too few workgroups: +-1 % end to end for 2..4 chunks, -13 % for 8.  Not kept.)
  // Sub-batches on streams of their own (round 5).  A launch of the pipelined 3x3 kernels is one workgroup per CU and tile, and at
  // B = 64 the tile counts are 1.8 (256 channels) / 3.6 (128) / 0.94 (512) times the 256 CUs: the last round of every launch leaves
  // 6-10 % of the chip idle and the next launch cannot start before it ends.  Two half batches are two independent chains of
  // launches: while one's last workgroups run, the other's fill the idle CUs (a second workgroup of these kernels does not fit
  // a CU, so the co-scheduled kernel gets exactly the idle ones).  Same kernels, same per-image arithmetic: the features are bit
  // for bit those of the single-stream pass wherever the tile modes agree.  Weights are read once per sub-batch instead of once.
  constexpr int kMaxSub = adx_resnet::kMaxSub;
  int nsub = 1;
  if (batch >= 32 && !debug_switches().check_range && fuse && conv2d_hs_eligible(r->convs[0])) {
    nsub = std::min(debug_switches().resnet_streams, kMaxSub);
    while (nsub > 1 && batch / nsub < 16) --nsub;
  }
  if (nsub > 1) {
    // not inside a stream capture: a replayed graph with the forked branches measured 2 % SLOWER per tick than the single chain
    // (profiles/README.md, round 5), and streams are not created while capturing
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) nsub = 1;
  }
  if (nsub > 1) {
    int dev = 0;
    ADX_CHECK_HIP(hipGetDevice(&dev));
    if (r->side_device != dev || r->side[nsub - 2] == nullptr) {
      {
        if (r->side_device != dev) {
          // the handle moved to another device (module.to(other_gpu)): streams AND events belong to the device they were
          // created on -- recording an old event on a new stream fails -- so all of them are recreated here
          for (auto& st : r->side) if (st != nullptr) { (void)hipStreamDestroy(st); st = nullptr; }
          if (r->ev_fork != nullptr) { (void)hipEventDestroy(r->ev_fork); r->ev_fork = nullptr; }
          for (auto& ev : r->ev_join) if (ev != nullptr) { (void)hipEventDestroy(ev); ev = nullptr; }
          r->side_device = dev;
        }
        if (r->ev_fork == nullptr) ADX_CHECK_HIP(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));
        for (int k = 0; k < nsub - 1; ++k) {
          if (r->side[k] == nullptr) ADX_CHECK_HIP(hipStreamCreateWithFlags(&r->side[k], hipStreamNonBlocking));
          if (r->ev_join[k] == nullptr) ADX_CHECK_HIP(hipEventCreateWithFlags(&r->ev_join[k], hipEventDisableTiming));
        }
      }
    }
  }
  hipStream_t streams[kMaxSub];
  int n0s[kMaxSub], ns[kMaxSub];
  Cursor st[kMaxSub]{};
  for (int k = 0, at = 0; k < nsub; ++k) {
    streams[k] = k == 0 ? s : r->side[k - 1];
    n0s[k] = at;
    ns[k] = batch / nsub + (k < batch % nsub ? 1 : 0);
    at += ns[k];
  }
  const size_t scratch_per = ((size_t)batch * 64 * h1 * w1 / nsub) & ~(size_t)63;
  // (only where the fused stem + pool launch leaves the stem map's region unwritten; each sub-batch gets its own slice)
  const bool stem_free = (fuse || frames_u8 != nullptr) && conv2d_hs_eligible(r->convs[0]);
  auto lend = [&](int k) { if (stem_free) scratch_scope.set(stem + (size_t)k * scratch_per, scratch_per); };
  int rc = ADX_OK;
  // The stem and the first layer (the blocks in front of the first downsample block) run as ONE chain on the whole batch, the fork
  // comes behind them: their launches are thousands of two-per-CU workgroups whose last round hardly matters, and split they
  // only stream their operands twice (-1.6 % per faithful step, profiles/README.md).  The hand-off format does not depend on the
  // batch: a downsample block's fused stride-2 launch reads cells or fp32 as it finds them.  ADX_RESNET_SPLIT_FROM=<block> shortens it.
  size_t first_split = 0;
  if (nsub > 1) {
    while (first_split < nblocks && !r->block_has_ds[first_split]) ++first_split;
    // (only blocks of the FIRST layer: their maps have the pooled map's per-image size, so the whole-batch tensors they leave are
    // laid out exactly like the sub-batches' regions; the override can only shorten the prefix)
    if (debug_switches().resnet_split_from >= 0) first_split = std::min((size_t)debug_switches().resnet_split_from, first_split);
  }
  if (first_split > 0) {
    Cursor whole{};
    lend(0);
    rc = run_stem(whole, 0, batch, batch, s);
    for (size_t b = 0; b < first_split && rc == ADX_OK; ++b) rc = run_block(b, whole, 0, batch, batch, s);
    for (int k = 0; k < nsub; ++k) st[k] = whole;
  }
  if (nsub > 1) {
    ADX_CHECK_HIP(hipEventRecord(r->ev_fork, s));
    for (int k = 1; k < nsub; ++k) ADX_CHECK_HIP(hipStreamWaitEvent(streams[k], r->ev_fork, 0));
  }
  // launches are issued layer by layer, alternating between the sub-batches, so that every stream's queue has work early
  for (int k = 0; k < nsub && rc == ADX_OK && first_split == 0; ++k) { lend(k); rc = run_stem(st[k], n0s[k], ns[k], ns[k], streams[k]); }
  for (size_t b = first_split; b < nblocks && rc == ADX_OK; ++b)
    for (int k = 0; k < nsub && rc == ADX_OK; ++k) { lend(k); rc = run_block(b, st[k], n0s[k], ns[k], ns[k], streams[k]); }
  for (int k = 0; k < nsub && rc == ADX_OK; ++k) {
    const size_t off = (size_t)n0s[k] * 64 * h2 * w2;          // the sub-batch's region (run_block)
    rc = avgpool_fc_launch(buf[st[k].cur] + off, base + r->o_fcw, base + r->o_fcb, feature + (size_t)n0s[k] * r->out_dim, ns[k], 512,
                           st[k].H * st[k].W, r->out_dim, streams[k], st[k].cells);
  }
  // join even after an error: a side stream must not be left forked from a capturing stream
  for (int k = 1; k < nsub; ++k) {
    if (hipEventRecord(r->ev_join[k - 1], streams[k]) == hipSuccess) (void)hipStreamWaitEvent(s, r->ev_join[k - 1], 0);
  }
  return rc;
}

}  // extern "C"
