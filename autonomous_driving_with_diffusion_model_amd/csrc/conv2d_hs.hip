// fp32-equivalent conv2d on the fp16 matrix cores of gfx950 ("hs" = hi/lo split).
//
// gfx950 has no xf32/TF32 and its fp32 MFMA runs at the vector rate (157 TFLOP/s, 1/16 of the fp16/bf16
// rate), so an fp32-in/fp32-out convolution that wants the matrix cores' real throughput has to feed them
// 16-bit operands.  Every fp32 operand x (activation or weight) is split EXACTLY-ish into two fp16 numbers
//     hi = fp16(x)                     (round to nearest even, 11 significant bits)
//     lo = fp16((x - hi) * 2^11)       (the next 11 bits, scaled so it never falls into fp16 subnormals)
// so that x = hi + lo * 2^-11 up to a relative error of 2^-22 .. 2^-23, and a product is evaluated as
//     x * w  ~=  hi_x * hi_w  +  2^-11 * (hi_x * lo_w + lo_x * hi_w)
// (the dropped lo*lo term is 2^-22 relative).  fp16 x fp16 products are exact in the MFMA's fp32 datapath
// and both sums are accumulated in fp32 (two accumulators: `main` and `lo`, combined once in the epilogue),
// so the result carries a per-product error of ~7e-8 relative -- below the 1.6e-7 that an fp32 accumulation
// of the same length has on its own (measured: tests/test_gpu_ops.py::test_conv2d_hs_*).  It costs
// 3 x v_mfma_f32_32x32x16_f16 (16 k-values each, 32 cycles) where the exact-fp32 path needs
// 8 x v_mfma_f32_32x32x2_f32 (64 cycles each): 5.3x less matrix time, the same 4 bytes per operand in LDS.
// Range: |x| must stay below 65504 (fp16 max); larger values become inf/NaN loudly.  The ResNet's
// normalised image and batch-normalised activations are O(1..100).
//
// Tile: one workgroup = 4 waves = 8 output rows x 32 output columns x 64 output channels; wave w owns rows
// 2w, 2w+1.  Per 16-channel chunk the input patch is split while it is staged (once per staged element) into
// LDS as [k-half][plane][pixel] 16-byte cells of 8 channels -- exactly the B fragment of one lane -- and the
// weight slab arrives pre-split from adx_*_pack as [tap][plane][k-half][cout] cells (the A fragment), so every
// operand read is one conflict-free ds_read_b128.  Global loads of the next stage are issued before the MFMAs
// of the current one and land in the other LDS buffer after them (one barrier per stage).  Epilogue as conv2d.hip: BN scale/shift, residual, ReLU, NCHW stores.
#include <stdlib.h>

#include "adx_common.h"
#include "conv2d_internal.h"

namespace adx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kHsCout = 64;          // output channels per workgroup
constexpr int kHsCC = 16;            // channels per chunk = K of one MFMA
constexpr float kLoScale = 2048.f;   // 2^11

__device__ __forceinline__ void split8(const float* v, u32x4& hi, u32x4& lo) {
  f16x8 h, l;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 hj = (_Float16)v[j];
    h[j] = hj;
    l[j] = (_Float16)((v[j] - (float)hj) * kLoScale);
  }
  hi = __builtin_bit_cast(u32x4, h);
  lo = __builtin_bit_cast(u32x4, l);
}

template <int STRIDE, int K>
__global__ void __launch_bounds__(256, 2) conv2d_hs_kernel(const Conv2dArgs a) {
  constexpr int TH = 8;
  constexpr int PH = (TH - 1) * STRIDE + K;
  constexpr int PW = (kTileW - 1) * STRIDE + K;
  constexpr int PLANE = PH * PW;                  // pixels of the staged patch
  constexpr int NITEM = 2 * PLANE;                // (k-half, pixel) cells per chunk
  constexpr int PIT = (NITEM + 255) / 256;
  constexpr int NTAPS = K * K;
  constexpr int NW = NTAPS * 256;                 // 16-byte weight cells per chunk: [tap][plane][k-half][64]
  constexpr int WST = K * 256;                    // weight cells of one stage (= one kernel row kh)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* patch = reinterpret_cast<u32x4*>(smem_raw);   // 2 x [k-half][plane][PLANE]
  u32x4* wl = patch + 8 * PLANE;                       // 2 x [kw][plane][k-half][64]
  float* ss = reinterpret_cast<float*>(wl + 2 * WST);  // scale[64], shift[64]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> tile: consecutive ids go to different XCDs (round robin), so give each XCD one contiguous
  // eighth of the tile space; neighbouring tiles (shared halos, shared weight slabs) then meet in one L2
  int bid = blockIdx.x;
  {
    const int per = gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int n = bid;
  const int oy0 = ty * TH, ox0 = tx * kTileW;
  const int iy0 = oy0 * STRIDE - a.pad, ix0 = ox0 * STRIDE - a.pad;
  const int cout0 = ct * kHsCout;
  const int l31 = lane & 31, khalf = lane >> 5;
  const size_t hw = (size_t)a.H * a.W;
  const float* xin = a.x + (size_t)n * a.Cin * hw;
  const int nchunks = a.cin_pad / kHsCC;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w) + (size_t)ct * nchunks * NW;

  int goff[PIT];     // gather offset of cell k's first channel (-1: outside the image -> zeros)
#pragma unroll
  for (int k = 0; k < PIT; ++k) {
    const int e = tid + 256 * k;
    const int hg = e >= PLANE ? 1 : 0;
    const int p = e - hg * PLANE;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = e < NITEM && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[k] = ok ? (int)(hg * 8 * hw + (size_t)iy * a.W + ix) : -1;
  }
  if (tid < 2 * kHsCout) {
    const int c = cout0 + (tid & (kHsCout - 1));
    ss[tid] = a.scale == nullptr ? (tid < kHsCout ? 1.f : 0.f) : (tid < kHsCout ? a.scale[c] : a.shift[c]);
  }

  f32x16 accm[2][2], accl[2][2];   // [row][cout half]: hi*hi sums, cross-term sums (scaled by 2^11)
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) { accm[r][m][i] = 0.f; accl[r][m][i] = 0.f; }

  // Pipeline stage = (chunk, kernel row kh): weights are double-buffered per stage, the patch per chunk, so one
  // barrier per stage is enough and only K weight cells + PIT patch cells per thread are ever in registers.
  float pv[PIT][8];
  u32x4 wv[K];
  auto load_p = [&](int chunk) {
    const float* xc = xin + (size_t)chunk * kHsCC * hw;
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      const int g = goff[k] >= 0 ? goff[k] : 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = xc[(size_t)j * hw + g];
        pv[k][j] = goff[k] >= 0 ? v : 0.f;
      }
    }
  };
  auto store_p = [&](int buf) {
    u32x4* pd = patch + buf * 4 * PLANE;
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      const int e = tid + 256 * k;
      if (PIT * 256 == NITEM || e < NITEM) {
        const int hg = e >= PLANE ? 1 : 0;
        const int p = e - hg * PLANE;
        u32x4 hi, lo;
        split8(pv[k], hi, lo);
        pd[(hg * 2 + 0) * PLANE + p] = hi;
        pd[(hg * 2 + 1) * PLANE + p] = lo;
      }
    }
  };
  auto load_w = [&](int stage) {
    const u32x4* ws = wsrc + (size_t)stage * WST;
#pragma unroll
    for (int k = 0; k < K; ++k) wv[k] = ws[tid + 256 * k];
  };
  auto store_w = [&](int buf) {
#pragma unroll
    for (int k = 0; k < K; ++k) wl[buf * WST + tid + 256 * k] = wv[k];
  };

  const int pb_lane = khalf * 2 * PLANE + (wave * 2 * STRIDE) * PW + l31 * STRIDE;
  const int wa_lane = khalf * 64 + l31;
  const int nstages = nchunks * K;

  load_w(0);
  load_p(0);
  store_w(0);
  store_p(0);
  __syncthreads();
  int stage = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const u32x4* pb0 = patch + (chunk & 1) * 4 * PLANE + pb_lane;
#pragma unroll
    for (int kh = 0; kh < K; ++kh, ++stage) {
      const u32x4* wa0 = wl + (stage & 1) * WST + wa_lane;
      if (stage + 1 < nstages) load_w(stage + 1);
      if (kh == 0 && chunk + 1 < nchunks) load_p(chunk + 1);
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        f16x8 A[2][2], B[2][2];   // [plane][cout half], [plane][row]
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int m = 0; m < 2; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa0[(kw * 2 + pl) * 128 + m * 32]);
#pragma unroll
          for (int r = 0; r < 2; ++r) B[pl][r] = __builtin_bit_cast(f16x8, pb0[pl * PLANE + (r * STRIDE + kh) * PW + kw]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            accm[r][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], B[0][r], accm[r][m], 0, 0, 0);
            accl[r][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[0][m], B[1][r], accl[r][m], 0, 0, 0);
            accl[r][m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[1][m], B[0][r], accl[r][m], 0, 0, 0);
          }
      }
      if (stage + 1 < nstages) store_w((stage + 1) & 1);
      if (kh == K - 1 && chunk + 1 < nchunks) store_p((chunk + 1) & 1);
      __syncthreads();
    }
  }

  // ---- epilogue: combine, BN scale/shift, residual, ReLU; lane = pixel column, register = channel ----
  const int ox = ox0 + l31;
  const size_t img = (size_t)n * a.Cout * a.OH * a.OW;
  const size_t plane_o = (size_t)a.OH * a.OW;
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int oy = oy0 + wave * 2 + rr;
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
        float v = accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale);
        v = v * ss[cl] + ss[kHsCout + cl];
        v += rv[r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        a.y[img + (size_t)(cout0 + cl) * plane_o + pix] = v;
      }
    }
  }
}

// fp32 [M][Kc][taps] (forward: M = cout, Kc = cin) or its data-gradient view (dgrad: M = original cin,
// Kc = original cout, taps flipped) -> [M/64][Kc/16][tap][plane][k-half][64][8] fp16
__global__ void conv2d_hs_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ p, int M, int Kc, int Kreal,
                                      int taps, int dgrad, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;   // over [M/64][Kc/16][tap][k-half][64][8]
  if (idx >= total) return;
  const int j = idx & 7;
  const int ml = (idx >> 3) & 63;
  const int h = (idx >> 9) & 1;
  size_t rest = idx >> 10;
  const int tap = rest % taps; rest /= taps;
  const int nchunks = Kc / kHsCC;
  const int chunk = rest % nchunks;
  const int ct = rest / nchunks;
  const int m = ct * 64 + ml, kc = chunk * kHsCC + h * 8 + j;
  float v = 0.f;
  if (kc < Kreal) v = dgrad ? w[((size_t)kc * M + m) * taps + (taps - 1 - tap)] : w[((size_t)m * Kreal + kc) * taps + tap];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * kLoScale);
  const size_t cell = ((((size_t)(ct * nchunks + chunk) * taps + tap) * 2 + 0) * 2 + h) * 64 + ml;
  p[cell * 8 + j] = hi;
  p[(cell + 128) * 8 + j] = lo;     // plane 1 is 2 * 64 cells further
}

bool conv2d_hs_eligible(const ConvSpec& L) {
  static int exact = -1;
  if (exact < 0) {
    const char* e = getenv("ADX_CONV_EXACT");    // ADX_CONV_EXACT=1: keep every conv on the exact-fp32 MFMA kernels
    exact = (e != nullptr && e[0] == '1') ? 1 : 0;
  }
  if (exact) return false;
  return L.k == 3 && L.stride == 1 && L.cin % kHsCC == 0 && L.cin_pad == L.cin && L.cout % kHsCout == 0;
}

int conv2d_hs_pack(const ConvSpec& c, const float* w, void* packed, int dgrad, hipStream_t s) {
  const size_t total = (size_t)c.cout * c.cin_pad * c.k * c.k;
  conv2d_hs_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
      w, (_Float16*)packed, c.cout, c.cin_pad, c.cin, c.k * c.k, dgrad, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

template <int STRIDE, int K>
static int hs_launch_t(Conv2dArgs a, hipStream_t s) {
  constexpr int PH = 7 * STRIDE + K, PW = (kTileW - 1) * STRIDE + K;
  constexpr size_t lds = (size_t)2 * 64 * PH * PW + (size_t)2 * K * 256 * 16 + 2 * kHsCout * sizeof(float);
  static_assert(lds <= 80 * 1024, "two workgroups per CU need <= 80 KB each");
  static bool attr = false;
  if (!attr) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs_kernel<STRIDE, K>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr = true;
  }
  a.tiles_x = ceil_div(a.OW, kTileW); a.tiles_y = ceil_div(a.OH, 8); a.cout_tiles = a.Cout / kHsCout;
  const size_t grid = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y * a.N;
  ADX_REQUIRE(grid < (1u << 31), "conv2d_hs: grid too large");
  conv2d_hs_kernel<STRIDE, K><<<dim3((unsigned)grid), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int conv2d_hs_launch(const ConvSpec& L, Conv2dArgs a, hipStream_t s) {
  ADX_REQUIRE((size_t)L.cin * a.H * a.W < (1u << 31), "conv2d_hs: image plane too large for 32-bit gather offsets");
  if (L.k == 3 && L.stride == 1) return hs_launch_t<1, 3>(a, s);
  set_error("conv2d_hs: no kernel for k=%d stride=%d", L.k, L.stride);
  return ADX_ERR_INVALID;
}

}  // namespace adx
