// Weight gradient of the 3x3 (stride 1 and 2) convolutions on the fp16 matrix cores (hi/lo split operands, see
// conv2d_hs.hip for the arithmetic):
//     dW[co][ci][kh][kw] = sum over (n, oy, ox) of dy[n][co][oy][ox] * x[n][ci][S oy + kh - 1][S ox + kw - 1]
// (torch.nn.grad.conv2d_weight of modeling/resnet.py's conv3x3).  Per tap this is a GEMM with M = co, N = ci and
// the PIXELS as the reduction axis, so both operands are needed k-major while NCHW (and the forward kernel's LDS
// image) is channel-major per pixel.  gfx950's transposing LDS read does the turn for free: activations and
// gradients are staged exactly like the forward patch -- 16-byte cells of 8 channels per pixel, split into an fp16
// hi and a scaled lo plane -- in [32-channel panel][pixel][32] images, and ds_read_b64_tr_b16 hands every lane the
// 4 consecutive PIXELS of its channel.  A tap's kw shift is then a whole-cell (64-byte) offset, never a misaligned
// read.
//
// One workgroup = 12 waves owns a 64 (co) x 64 (ci) x 9 (taps) block of dW: wave (cb, nb, kh) accumulates the three
// kw taps of kernel row kh for output-channel panel cb and input-channel panel nb (3 x 2 accumulators of 16
// registers).  It walks its share of (image, 64- or 32-pixel column segment) units row by row with a rolling
// window: per output row one new dy row and one new x row are fetched (buffer loads, range check = zero padding),
// split and stored while the previous row is multiplied; x rows live in a ring of four, dy rows in two buffers,
// one barrier per row.  The block is reduced across workgroups with coalesced float atomics through an LDS
// transpose ([co][ci][9] order = dW's memory order).  dy is far below fp16's normal range: its max|.| comes from
// the kernel that produced it and moves it into range by an exact power of two.
#include <stdlib.h>

#include <algorithm>

#include "adx_common.h"
#include "conv2d_internal.h"
#include "conv2d_hs_common.h"

namespace adx {

typedef __fp16 h4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) h4 lds_h4;

struct WgradHsArgs {
  const float* x;        // [N][Cin][H][W], or (XC) the same bytes as a cell tensor [N][Cin/8][hi, lo][H][W][8 halves]
  const float* dy;       // [N][Cout][OH][OW], or (XA) a cell tensor already multiplied by the power of two dy_amax points at
  float* dw;             // [Cout][Cin][3][3], zeroed by the caller (conv2d_wgrad)
  const uint32_t* dy_amax;
  int dy_amax_n;         // < 0: dy_amax points at two floats {xs, 1 / xs} (XA: the scale dy's cells were written with)
  int N, Cin, Cout, H, W, OH, OW;
  int segs, units, units_per_wg, n_ci_tiles, n_co_tiles;
  float* part;           // deterministic mode: [splits][Cout][Cin][9] partial sums (split s of a tile writes its own copy), else null
  size_t part_stride;    // Cout * Cin * 9
};

constexpr float kWLo = 2048.f;
constexpr int kWThreads = 768;

template <int SECOND>
__device__ __forceinline__ f16x8 tr_pair(const unsigned char* p) {
  // 8 consecutive k (pixels) of this lane's channel: two transposing reads of 4 pixel rows each (64 B per staged
  // pixel; SECOND = byte distance of 4 k-steps: 256 for stride 1, 512 when every other staged pixel is a k)
  const u32x2 a = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(p)));
  const u32x2 b = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_h4*)(p + SECOND)));
  u32x4 v;
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
  return __builtin_bit_cast(f16x8, v);
}

// XC: x arrives already split -- the cell tensor the training forward keeps between a block's convs (resnet_train.hip:
// bn_apply_cells_kernel; the same hi / lo halves this kernel's staging would compute from the fp32 map): two 16-byte loads per
// staged cell instead of eight dwords and the conversion
// XA: the same for dy -- a gradient the BatchNorm-backward apply pass wrote as cells under a scale it chose from a bound
// (resnet_train.hip: bn_bwd_apply_groups_kernel)
// (With both operands as cells a staged row is a copy; LDS-DMA loads for it -- a wave writes 16 pixels x the 4 cell groups of a panel,
// issued at the top of an iteration, awaited before its barrier -- were built and measured 0.3 ms per training step SLOWER than the
// register path, where the same loads in conv2d_hs16.hip gain 0.3: profiles/README.md, round 5.  Removed.)
template <int NPX, int S, bool XC = false, bool XA = false>
__global__ void __launch_bounds__(kWThreads) conv2d_wgrad_hs_kernel(const WgradHsArgs a) {
  constexpr int NPXB = S * NPX + 2;               // x row segment with its halo columns
  constexpr int NSLOT = S == 1 ? 4 : 6;           // x rows kept: 3 in use + S arriving
  constexpr int KS = NPX / 16;                    // MFMA k-steps per row segment
  constexpr int A_PLANE = 2 * NPX * 64;           // bytes: [co panel][pixel][32 halves]
  constexpr int A_BUF = 2 * A_PLANE;              // hi + lo
  constexpr int B_PLANE = 2 * NPXB * 64;          // bytes: [ci panel][pixel + halo][32 halves]
  constexpr int B_SLOT = 2 * B_PLANE;
  constexpr int NA = NPX * 8, NB = NPXB * 8;      // 16-byte cells per staged dy / x row
  constexpr int PIT = (NA + S * NB + kWThreads - 1) / kWThreads;
  static_assert(NA % 64 == 0, "a wave stages either dy cells or x cells");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lA = smem;                       // 2 x A_BUF  (dy row r -> buffer r & 1)
  unsigned char* lB = smem + 2 * A_BUF;           // NSLOT x B_SLOT (x row r -> slot r % NSLOT)
  uint32_t* red = reinterpret_cast<uint32_t*>(lB + NSLOT * B_SLOT);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave / 6, nb = (wave / 3) % 2, kh = wave % 3;
  const int tiles = a.n_co_tiles * a.n_ci_tiles;
  const int tile = blockIdx.x % tiles, split = blockIdx.x / tiles;
  const int co0 = (tile / a.n_ci_tiles) * 64, ci0 = (tile % a.n_ci_tiles) * 64;
  const size_t hw = (size_t)a.H * a.W;
  const uint32_t plane_bytes = (uint32_t)(hw * sizeof(float));
  constexpr uint32_t kOutside = 0xC0000000u;

  // dy's dynamic range: scale so that max|dy| lands in [2^14, 2^15), undone exactly in the epilogue
  float xs = 1.f, xs_inv = 1.f;
  if (a.dy_amax != nullptr && a.dy_amax_n < 0) {
    xs_inv = reinterpret_cast<const float*>(a.dy_amax)[1];
  } else if (a.dy_amax != nullptr) {
    uint32_t b = 0;
    for (int i = tid; i < a.dy_amax_n; i += kWThreads) b = a.dy_amax[i] > b ? a.dy_amax[i] : b;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
      b = o > b ? o : b;
    }
    if (lane == 0) red[wave] = b;
    __syncthreads();
    b = 0;
#pragma unroll
    for (int w = 0; w < kWThreads / 64; ++w) b = red[w] > b ? red[w] : b;
    const int e = (int)((b >> 23) & 0xFF);
    if (e != 0 && e != 255) {
      int sh = 127 + 14 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      xs = __builtin_bit_cast(float, (uint32_t)(127 + sh) << 23);
      xs_inv = __builtin_bit_cast(float, (uint32_t)(127 - sh) << 23);
    }
  }

  f32x16 am[3], al[3];   // per kw tap: hi*hi sums, cross-term sums (scaled by 2^11)
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int i = 0; i < 16; ++i) { am[k][i] = 0.f; al[k][i] = 0.f; }

  // transposing reads: lane (16-lane group g, member 4q + p) supplies pixel row q, channels 4p..4p+3 of channel
  // group g & 1; the group's k half is g >> 1 (MFMA lanes 32..63 hold k = 8..15)
  const int gi = lane >> 4, li = lane & 15;
  const int tr_px = 8 * (gi >> 1) + (li >> 2), tr_ch = ((gi & 1) * 16 + 4 * (li & 3)) * 2;
  const int aoff = cb * NPX * 64 + tr_px * 64 + tr_ch;
  const int boff = nb * NPXB * 64 + tr_px * S * 64 + tr_ch;

  // staging cells of this thread (the same for every row): kind, LDS byte offset inside a buffer / slot, column
  int sto[PIT], colv[PIT], rsel[PIT];
  uint32_t chan[PIT];
  bool isA[PIT];
  const uint32_t dplane_bytes = (uint32_t)((size_t)a.OH * a.OW * sizeof(float));
#pragma unroll
  for (int k = 0; k < PIT; ++k) {
    const int e = tid + kWThreads * k;
    isA[k] = e < NA;
    rsel[k] = 0;
    if (isA[k]) {
      const int px = e % NPX, cg = e / NPX;
      sto[k] = ((cg >> 2) * NPX + px) * 64 + (cg & 3) * 16;
      colv[k] = px;
      chan[k] = (uint32_t)cg * 8u * dplane_bytes;
    } else {
      const int e2 = e - NA;
      const int r = e2 / NB, e3 = e2 - r * NB;
      const int pp = e3 % NPXB, cg = e3 / NPXB;
      sto[k] = r < S ? ((cg >> 2) * NPXB + pp) * 64 + (cg & 3) * 16 : -1;
      rsel[k] = r;
      colv[k] = pp - 1;
      chan[k] = (uint32_t)cg * 8u * plane_bytes;
    }
  }

  const int u0 = split * a.units_per_wg;
  const int u1 = u0 + a.units_per_wg < a.units ? u0 + a.units_per_wg : a.units;
  float pv[PIT][8];
  for (int u = u0; u < u1; ++u) {
    const int n = u / a.segs, ox0 = (u - n * a.segs) * NPX;
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy + ((size_t)n * a.Cout + co0) * a.OH * a.OW), 0, (int)(64 * dplane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + ((size_t)n * a.Cin + ci0) * hw), 0, (int)(64 * plane_bytes), 0x00020000);
    // iteration t: fetch dy row t+1 and the S x rows that output row t+1 adds to the window, multiply output row t,
    // publish the fetched rows
    for (int t = (S == 1 ? -2 : -1); t < a.OH; ++t) {
      const int xrow0 = S * (t + 1) + (S == 1 ? 1 : 0);
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        const bool ka = __builtin_amdgcn_readfirstlane((int)isA[k]) != 0;     // wave-uniform by construction
        if (ka) {
          const int row = t + 1, col = ox0 + colv[k];
          const bool ok = row >= 0 && row < a.OH && col < a.OW;
          if constexpr (XA) {
            const uint32_t vo = ok ? chan[k] + (uint32_t)(row * a.OW + col) * 16u : kOutside;
            const u32x4 h4 = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 0, 0);
            const u32x4 l4 = __builtin_amdgcn_raw_buffer_load_b128(rdy, vo, 4 * dplane_bytes, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pv[k][j] = u2f(h4[j]); pv[k][4 + j] = u2f(l4[j]); }
          } else {
            const uint32_t vo = ok ? chan[k] + (uint32_t)(row * a.OW + col) * 4u : kOutside;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              pv[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rdy, vo, j * dplane_bytes, 0));
          }
        } else {
          const int row = xrow0 + rsel[k], col = S * ox0 + colv[k];
          const bool ok = sto[k] >= 0 && row >= 0 && row < a.H && col >= 0 && col < a.W;
          if constexpr (XC) {
            const uint32_t vo = ok ? chan[k] + (uint32_t)(row * a.W + col) * 16u : kOutside;
            const u32x4 h4 = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 0, 0);
            const u32x4 l4 = __builtin_amdgcn_raw_buffer_load_b128(rx, vo, 4 * plane_bytes, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { pv[k][j] = u2f(h4[j]); pv[k][4 + j] = u2f(l4[j]); }
          } else {
            const uint32_t vo = ok ? chan[k] + (uint32_t)(row * a.W + col) * 4u : kOutside;
#pragma unroll
            for (int j = 0; j < 8; ++j)
              pv[k][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rx, vo, j * plane_bytes, 0));
          }
        }
      }
      // kernel row 0 of output row 0 would read x row -1 (zero padding): nothing to add
      if (t >= 0 && !(kh == 0 && t == 0)) {
        const unsigned char* Ab = lA + (t & 1) * A_BUF + aoff;
        const unsigned char* Bb = lB + ((S * t + kh - 1 + NSLOT) % NSLOT) * B_SLOT + boff;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const f16x8 Ahi = tr_pair<256>(Ab + ks * 1024), Alo = tr_pair<256>(Ab + A_PLANE + ks * 1024);
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const f16x8 Bhi = tr_pair<256 * S>(Bb + ks * 1024 * S + kw * 64);
            const f16x8 Blo = tr_pair<256 * S>(Bb + B_PLANE + ks * 1024 * S + kw * 64);
            am[kw] = hs_mfma(Ahi, Bhi, am[kw]);
            al[kw] = hs_mfma(Ahi, Blo, al[kw]);
            al[kw] = hs_mfma(Alo, Bhi, al[kw]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < PIT; ++k) {
        if (sto[k] < 0) continue;
        const bool ka = isA[k];
        unsigned char* d = ka ? lA + ((t + 1) & 1) * A_BUF + sto[k] : lB + ((xrow0 + rsel[k]) % NSLOT) * B_SLOT + sto[k];
        const bool kau = __builtin_amdgcn_readfirstlane((int)ka) != 0;
        if ((XC && !kau) || (XA && kau)) {       // a cell: the halves as they were loaded
          u32x4 h4, l4;
#pragma unroll
          for (int j = 0; j < 4; ++j) { h4[j] = f2u(pv[k][j]); l4[j] = f2u(pv[k][4 + j]); }
          *reinterpret_cast<u32x4*>(d) = h4;
          *reinterpret_cast<u32x4*>(d + (kau ? A_PLANE : B_PLANE)) = l4;
          continue;
        }
        f16x8 h, l;
        const float sc = ka ? xs : 1.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = pv[k][j] * sc;
          const _Float16 hj = (_Float16)v;
          h[j] = hj;
          l[j] = (_Float16)((v - (float)hj) * kWLo);
        }
        *reinterpret_cast<u32x4*>(d) = __builtin_bit_cast(u32x4, h);
        *reinterpret_cast<u32x4*>(d + (ka ? A_PLANE : B_PLANE)) = __builtin_bit_cast(u32x4, l);
      }
      __syncthreads();
    }
  }

  // ---- reduce across workgroups: [32 co][64 ci][9] floats per co panel through LDS, then coalesced atomics ----
  float* tbuf = reinterpret_cast<float*>(smem);
  const int l31 = lane & 31, khalf = lane >> 5;
#pragma unroll 1
  for (int pb = 0; pb < 2; ++pb) {
    if (cb == pb) {
#pragma unroll
      for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co_l = (r & 3) + 8 * (r >> 2) + 4 * khalf;
          tbuf[(co_l * 64 + nb * 32 + l31) * 9 + kh * 3 + kw] = (am[kw][r] + al[kw][r] * (1.f / kWLo)) * xs_inv;
        }
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * 576; idx += kWThreads) {
      const int co_l = idx / 576, rem = idx - co_l * 576;
      const size_t at = ((size_t)(co0 + pb * 32 + co_l) * a.Cin + ci0) * 9 + rem;
      if (a.part != nullptr) a.part[(size_t)split * a.part_stride + at] = tbuf[idx];      // reduced in index order by wgrad_parts_reduce_kernel
      else atomicAdd(a.dw + at, tbuf[idx]);
    }
    __syncthreads();
  }
}

// deterministic mode: dw[i] += part[0][i] + part[1][i] + ... in index order (four elements per thread, eight loads in flight)
__global__ void __launch_bounds__(256) wgrad_parts_reduce_kernel(const float* __restrict__ part, size_t stride, int nparts,
                                                                 float* __restrict__ dw, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
  for (int p0 = 1; p0 < nparts; p0 += 8) {
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(part + (size_t)(p0 + j < nparts ? p0 + j : 0) * stride)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (p0 + j < nparts) v += t[j];
  }
  reinterpret_cast<f32x4*>(dw)[i] += v;
}

static thread_local float* t_wgrad_parts = nullptr;
static thread_local size_t t_wgrad_parts_floats = 0;
void conv2d_wgrad_set_partials(float* p, size_t floats) { t_wgrad_parts = p; t_wgrad_parts_floats = p != nullptr ? floats : 0; }
size_t conv2d_wgrad_partials_floats() { return debug_switches().wgrad_deterministic ? (size_t)256 * 64 * 64 * 9 : 0; }

bool conv2d_wgrad_hs_eligible(int Cin, int Cout, int k, int stride, int pad) {
  const bool exact = debug_switches().conv_exact || debug_switches().wgrad_exact;
  return !exact && k == 3 && (stride == 1 || stride == 2) && pad == 1 && Cin % 64 == 0 && Cout % 64 == 0;
}

template <int NPX, int S, bool XC, bool XA = false>
static int wgrad_hs_launch(WgradHsArgs a, hipStream_t s) {
  constexpr int NSLOT = S == 1 ? 4 : 6;
  constexpr size_t lds = (size_t)2 * (2 * 2 * NPX * 64) + (size_t)NSLOT * (2 * 2 * (S * NPX + 2) * 64) + 64;
  constexpr size_t need = std::max(lds, (size_t)32 * 576 * sizeof(float));
  static_assert(need <= 160 * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_hs_kernel<NPX, S, XC, XA>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)need));
    once.commit();
  }
  a.segs = ceil_div(a.OW, NPX);
  a.units = a.N * a.segs;
  const int tiles = a.n_co_tiles * a.n_ci_tiles;
  int splits = std::max(1, 256 / tiles);          // one 12-wave workgroup per CU
  if (splits > a.units) splits = a.units;
  a.units_per_wg = ceil_div(a.units, splits);
  splits = ceil_div(a.units, a.units_per_wg);
  a.part = nullptr; a.part_stride = 0;
  if (debug_switches().wgrad_deterministic) {
    // ADX_WGRAD_DETERMINISTIC=1: every (tile, split) workgroup leaves its 64 x 64 x 9 block in a copy of dw of its split, and one
    // more launch adds the copies up in index order (splits * Cout * Cin * 9 <= 256 * 64 * 64 * 9 floats: tiles * splits <= 256)
    a.part_stride = (size_t)a.Cout * a.Cin * 9;
    ADX_REQUIRE(t_wgrad_parts != nullptr && (size_t)splits * a.part_stride <= t_wgrad_parts_floats && (a.part_stride & 3) == 0 &&
                    (reinterpret_cast<uintptr_t>(a.dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(t_wgrad_parts) & 15) == 0,
                "conv2d_wgrad_hs: ADX_WGRAD_DETERMINISTIC=1 needs the partial-sum scratch (%zu floats; the caller lent %zu)",
                (size_t)splits * a.part_stride, t_wgrad_parts_floats);
    a.part = t_wgrad_parts;
  }
  conv2d_wgrad_hs_kernel<NPX, S, XC, XA><<<dim3((unsigned)(tiles * splits)), dim3(kWThreads), need, s>>>(a);
  ADX_LAUNCH_CHECK();
  if (a.part != nullptr) {
    const size_t total4 = a.part_stride / 4;
    wgrad_parts_reduce_kernel<<<dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s>>>(a.part, a.part_stride, splits, a.dw, total4);
    ADX_LAUNCH_CHECK();
  }
  return ADX_OK;
}

int conv2d_wgrad_hs(const float* x, const float* dy, float* dw, int N, int Cin, int H, int W, int Cout, int stride,
                    const uint32_t* dy_amax, int dy_amax_n, hipStream_t s, bool x_cells, bool dy_cells) {
  ADX_REQUIRE(x && dy && dw, "conv2d_wgrad_hs: null tensor");
  ADX_REQUIRE((size_t)64 * H * W * sizeof(float) < 0xC0000000u, "conv2d_wgrad_hs: image too large for 32-bit offsets");
  WgradHsArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.dy_amax = dy_amax; a.dy_amax_n = dy_amax_n;
  a.N = N; a.Cin = Cin; a.Cout = Cout; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, 3, stride, 1); a.OW = conv_out_dim(W, 3, stride, 1);
  a.n_ci_tiles = Cin / 64; a.n_co_tiles = Cout / 64;
  ADX_REQUIRE(!dy_cells || (stride == 1 && dy_amax != nullptr && dy_amax_n < 0), "conv2d_wgrad_hs: a cell-layout dy comes with its scale (stride 1)");
  if (dy_cells) {
    const bool narrow = ceil_div(a.OW, 32) * 32 - a.OW < ceil_div(a.OW, 64) * 64 - a.OW;
    if (x_cells) return narrow ? wgrad_hs_launch<32, 1, true, true>(a, s) : wgrad_hs_launch<64, 1, true, true>(a, s);
    return narrow ? wgrad_hs_launch<32, 1, false, true>(a, s) : wgrad_hs_launch<64, 1, false, true>(a, s);
  }
  if (stride == 2) return x_cells ? wgrad_hs_launch<32, 2, true>(a, s) : wgrad_hs_launch<32, 2, false>(a, s);
  // rows of <= 32 (or 33..48 -> two 32-pixel segments waste less than one 64) pixels use the narrow variant
  const int waste64 = ceil_div(a.OW, 64) * 64 - a.OW, waste32 = ceil_div(a.OW, 32) * 32 - a.OW;
  if (x_cells) return waste32 < waste64 ? wgrad_hs_launch<32, 1, true>(a, s) : wgrad_hs_launch<64, 1, true>(a, s);
  return waste32 < waste64 ? wgrad_hs_launch<32, 1, false>(a, s) : wgrad_hs_launch<64, 1, false>(a, s);
}

}  // namespace adx
