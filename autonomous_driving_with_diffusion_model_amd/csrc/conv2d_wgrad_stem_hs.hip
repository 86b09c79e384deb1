// Weight gradient of the stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels: modeling/resnet.py:172) on the fp16 matrix
// cores, hi/lo split operands (conv2d_hs.hip has the arithmetic):
//     dW[co][ci][kh][kw] = sum over (n, oy, ox) of dy[n][co][oy][ox] * x[n][ci][2 oy + kh - 3][2 ox + kw - 3].
// A GEMM with M = 64 output channels, N = 147 (kh, ci, kw) combos (160: five 32-column tiles) and the output PIXELS as the
// reduction axis -- 69 GFLOP at B = 64, 3 x 256 x 900, as much as any 3x3 layer, which the exact-fp32 MFMA kernel
// (resnet_train.hip: conv2d_wgrad_kernel<2, 7, 5, 3>) took 1.0 ms for.
//
// One workgroup of 10 waves walks units of (image, 96-column segment, chunk of output rows) row by row.  Per output row it
// stages, split into fp16 hi / lo planes:
//   * the dy row segment as [64 co][96 pixels]: a lane's A fragment (its channel, 8 consecutive pixels) is one 16-byte read;
//   * the TWO input rows that enter the 7-row window, each as 21 (ci, kw) images R[p] = x[ci][iy][2 p + kw - 3] of the segment's
//     96 output pixels -- the stride and the tap shift are resolved while staging (a thread loads 16 consecutive floats and keeps
//     every other one), so a lane's B fragment (its (kh, ci, kw) column, 8 consecutive pixels) is again one aligned 16-byte
//     read, at any kw.  Input rows live in a ring of nine slots (seven in use, two arriving).
// Rows pitch 208 bytes: the 16 lanes of a read phase hit 16 x 4 distinct banks.  Wave (nt, kp) owns N tile nt for both M tiles
// and k-steps 3 kp .. 3 kp + 2 of the row's six: 6 MFMAs per 6 operand reads.  The fetch of the next row is issued before the
// multiplication of the current one; one barrier per row.  Per workgroup ONE pass of float atomics at the end.  dy is far below
// fp16's normal range: the kernel that produced it leaves partial maxima, and an exact power of two moves it into range.
#include <algorithm>

#include "adx_common.h"
#include "conv2d_internal.h"
#include "conv2d_hs_common.h"

namespace adx {

namespace {


struct WgradStemArgs {
  const float* x;        // [N][3][H][W]
  const float* dy;       // [N][64][OH][OW]
  float* dw;             // [64][3][7][7], zeroed by the caller
  const uint32_t* dy_amax;
  int dy_amax_n;
  int N, H, W, OH, OW;
  int segs, row_chunks, rows_per_unit, units;
};

constexpr int kNT = 640;                 // 10 waves
constexpr int kNPX = 96, kKS = kNPX / 16;
constexpr int kPitch = 208;              // bytes per staged row of 96 halves (+16: bank spread, see above)
constexpr int kAPlane = 64 * kPitch, kABuf = 2 * kAPlane;
constexpr int kBRows = 21;               // (ci, kw) images per input row
constexpr int kBPlane = kBRows * kPitch, kBSlot = 2 * kBPlane;
constexpr int kRing = 9;
constexpr int kZeroRow = 2 * kABuf + kRing * kBSlot;      // an all-zero row for the 13 padding columns of the last N tile
constexpr int kLds = kZeroRow + kPitch + 64;
constexpr float kLo = 2048.f;
static_assert(64 * (kNPX / 4) == 2 * kNT + 256 && kNT == 640, "three staging items per thread (see the kernel)");

__device__ __forceinline__ void split_pair(float v0, float v1, float s, float s2, uint32_t& hi, uint32_t& lo) {
  const float a0 = v0 * s, a1 = v1 * s, w0 = v0 * s2, w1 = v1 * s2;
  hi = __builtin_bit_cast(uint32_t, f16x2{(_Float16)a0, (_Float16)a1});
  float r0, r1;     // 2^11 (v - hi): hi rides as an fp16 operand, exact
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hi), "s"(-kLo), "v"(w0));
  asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hi), "s"(-kLo), "v"(w1));
  lo = __builtin_bit_cast(uint32_t, f16x2{(_Float16)r0, (_Float16)r1});
}

__global__ void __launch_bounds__(kNT) conv2d_wgrad_stem_hs_kernel(const WgradStemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* lA = smem;                       // 2 x kABuf (dy row oy -> buffer oy & 1), planes hi | lo
  unsigned char* lB = smem + 2 * kABuf;           // kRing x kBSlot (input row iy -> slot (iy + 18) % 9), planes hi | lo
  uint32_t* red = reinterpret_cast<uint32_t*>(smem + kZeroRow + kPitch);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = wave % 5, kp = wave / 5;
  const int l31 = lane & 31, khalf = lane >> 5;
  const int H = a.H, W = a.W, OH = a.OH, OW = a.OW;
  constexpr uint32_t kOutside = 0xC0000000u;

  for (int i = tid; i < kPitch / 4; i += kNT) reinterpret_cast<uint32_t*>(smem + kZeroRow)[i] = 0u;

  // dy's dynamic range: scale so that max|dy| lands in [2^14, 2^15), undone exactly at the end
  float xs = 1.f, xs_inv = 1.f;
  if (a.dy_amax != nullptr) {
    uint32_t b = 0;
    for (int i = tid; i < a.dy_amax_n; i += kNT) b = a.dy_amax[i] > b ? a.dy_amax[i] : b;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
      b = o > b ? o : b;
    }
    if (lane == 0) red[wave] = b;
    __syncthreads();
    b = 0;
#pragma unroll
    for (int w = 0; w < kNT / 64; ++w) b = red[w] > b ? red[w] : b;
    const int e = (int)((b >> 23) & 0xFF);
    if (e != 0 && e != 255) {
      int sh = 127 + 14 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      xs = __builtin_bit_cast(float, (uint32_t)(127 + sh) << 23);
      xs_inv = __builtin_bit_cast(float, (uint32_t)(127 - sh) << 23);
    }
  }

  // this lane's column of the GEMM: n = (kh * 3 + ci) * 7 + kw
  const int ncol = nt * 32 + l31;
  const bool nreal = ncol < 147;
  const int n_kh = nreal ? ncol / 21 : 0, n_cikw = nreal ? ncol % 21 : 0;
  const int frag_off = (kp * 3 * 16 + khalf * 8) * 2;        // byte offset of the first k-step's 8 pixels inside a row
  const int a_off0 = l31 * kPitch + frag_off;                 // M tile 0 (channel l31); tile 1: + 32 rows

  // Staging, per output row: every byte is fetched ONCE, by 16-byte loads whose lanes are neighbours in memory (the first
  // version gathered the x images element by element, one kw at a time: 4,900 L1 requests per row where 460 carry the data --
  // the launch was bound by them, profiles/README.md).
  //   * 1536 dy quads (co, 4 pixels): one load, split, one 8-byte LDS write per plane;
  //   * 300 x quads (new row r, ci, 4 columns): one load, split ONCE, and every value goes into the images of the kw that can
  //     read it -- column c is pixel p = (c - kw + 3) / 2 of image kw for the three or four kw of its parity; the two values of
  //     a quad with the same parity are neighbours there (14 small LDS writes per plane, immediate offsets).  Writes that fall
  //     a few pixels outside an image's 96 land in the pitch's slack (8 halves behind every row, which is also "before" the
  //     next one).
  // Three items per thread: two dy quads, then a dy quad (waves 0-3) or an x quad (waves 4-9: one (row, ci) of 50 quads per
  // wave, so a quad's right neighbour is the next lane); wave-uniform kinds.
  constexpr int kA4 = kNPX / 4;                                      // dy quads per channel row
  const bool third_is_a = wave < 4;                                   // uniform
  const bool valid2 = third_is_a || lane < 50;
  int a_co[3], a_px[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int idx = tid + kNT * k; a_co[k] = idx / kA4; a_px[k] = (idx % kA4) * 4; }
  const int b_r = wave >= 7 ? 1 : 0, b_ci = (wave + 2) % 3, b_j = lane;          // waves 4, 5, 6 -> ci 0, 1, 2 of row 0; 7, 8, 9 of row 1
  int sto[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) sto[k] = a_co[k] * kPitch + a_px[k] * 2;
  // x quad: byte offset of pixel 2 j of image (ci, kw = 0) inside a slot's plane; the images of one ci are kPitch apart
  const int stob = b_ci * 7 * kPitch + 4 * b_j;

  f32x16 am[2], al[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) { am[m][i] = 0.f; al[m][i] = 0.f; }

  // Two register sets: the fetch of row v + 1's additions is issued at the top of iteration v and converted at the end of
  // iteration v + 1 -- a whole iteration (the multiplication of a row and the conversion of another) of latency cover.
  float pv[2][3][4];
  const uint32_t rowA = (uint32_t)OW * 4u, rowB = (uint32_t)W * 4u;
  for (int u = blockIdx.x; u < a.units; u += gridDim.x) {
    const int n = u / (a.segs * a.row_chunks), rem = u - n * (a.segs * a.row_chunks);
    const int seg = rem / a.row_chunks, chunk = rem - seg * a.row_chunks;
    const int px0 = seg * kNPX, oy0 = chunk * a.rows_per_unit;
    const int oy1 = oy0 + a.rows_per_unit < OH ? oy0 + a.rows_per_unit : OH;
    const bool edge = px0 + kNPX > OW || 2 * px0 + 4 * 50 - 4 > W;        // uniform: the segment touches the right border
    const __amdgpu_buffer_rsrc_t rdy = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.dy + (size_t)n * 64 * OH * OW), 0, (int)((size_t)64 * OH * OW * sizeof(float)), 0x00020000);
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x + (size_t)n * 3 * H * W), 0, (int)((size_t)3 * H * W * sizeof(float)), 0x00020000);
    uint32_t base[3];
    int lim[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      base[k] = (uint32_t)((a_co[k] * OH * OW + px0 + a_px[k]) * 4);
      lim[k] = OW - (px0 + a_px[k]);                                        // values j < lim are real (right border)
    }
    const int c0 = 2 * px0 - 4 + 4 * b_j;                                   // first column of the x quad (-4: wholly left of the image)
    bool b_ok = false;
    if (!third_is_a) {
      b_ok = valid2 && c0 >= 0 && c0 < W;
      base[2] = (uint32_t)(((b_ci * H + b_r) * W + (c0 < 0 ? 0 : c0)) * 4);
      lim[2] = W - c0;
    }

    // target w = what output row w + 1 adds to the window: dy row w + 1, input rows 2 w + 4 and 2 w + 5
    auto fetch = [&](int w, int set) {
      const int row = w + 1, iy0 = 2 * w + 4;
      const bool okA = row >= oy0 && row < oy1, okB0 = iy0 >= 0 && iy0 < H, okB1 = iy0 + 1 >= 0 && iy0 + 1 < H;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        u32x4 t;
        if (k < 2 || third_is_a) {
          t = __builtin_amdgcn_raw_buffer_load_b128(rdy, okA && (k < 2 || valid2) ? base[k] + (uint32_t)row * rowA : kOutside, 0, 0);
        } else {
          const bool ok = b_ok && (b_r ? okB1 : okB0);
          t = __builtin_amdgcn_raw_buffer_load_b128(rx, ok ? base[2] + (uint32_t)iy0 * rowB : kOutside, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) pv[set][k][j] = u2f(t[j]);
      }
    };
    auto convert = [&](int w, int set) {
      unsigned char* const dA = lA + ((w + 1) & 1) * kABuf;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const bool isa = k < 2 || third_is_a;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = pv[set][k][j];
        if (edge) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v[j] = j < lim[k] ? v[j] : 0.f;
        }
        if (isa) {
          uint32_t h01, l01, h23, l23;
          split_pair(v[0], v[1], xs, xs * kLo, h01, l01);
          split_pair(v[2], v[3], xs, xs * kLo, h23, l23);
          typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
          *reinterpret_cast<u32x2*>(dA + sto[k]) = u32x2{h01, h23};
          *reinterpret_cast<u32x2*>(dA + kAPlane + sto[k]) = u32x2{l01, l23};
        } else if (valid2) {
          // columns c0, c0 + 2 (even elements) feed kw = 1, 3, 5 at pixels 2 j - (kw + 1) / 2 (+ 1); columns c0 + 1, c0 + 3 feed
          // kw = 0, 2, 4, 6 at pixels 2 j - kw / 2 (+ 1)
          uint32_t he, le, ho, lo_;
          split_pair(v[0], v[2], 1.f, kLo, he, le);
          split_pair(v[1], v[3], 1.f, kLo, ho, lo_);
          const int s0 = (2 * w + 4 + 18) % kRing, s1 = (2 * w + 5 + 18) % kRing;
          unsigned char* const d = lB + (b_r ? s1 : s0) * kBSlot + stob;
#pragma unroll
          for (int pl = 0; pl < 2; ++pl) {
            unsigned char* const dp = d + pl * kBPlane;
            const uint32_t e2 = pl ? le : he, o2 = pl ? lo_ : ho;
            // pixel pairs that start on an odd pixel of this quad start on an even one when taken (my second value, the next
            // quad's first value): the next lane's word by DPP, one v_alignbit -- every write is an aligned dword (misaligned
            // dword writes, which two adjacent 16-bit stores are merged into, cost ~25 clocks each here)
            const uint32_t en = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)e2, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
            const uint32_t on = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)o2, 0x130, 0xf, 0xf, true);
            const uint32_t ex = __builtin_amdgcn_alignbit(en, e2, 16), ox = __builtin_amdgcn_alignbit(on, o2, 16);
            // image kw at + kw * kPitch, pixel p at + 2 p bytes (stob already holds 2 * (2 j))
            *reinterpret_cast<uint32_t*>(dp + 3 * kPitch - 4) = e2;       // kw = 3: pixels 2 j - 2, 2 j - 1
            *reinterpret_cast<uint32_t*>(dp + 1 * kPitch) = ex;           // kw = 1: pixels 2 j, 2 j + 1 (columns c0 + 2, c0 + 4)
            *reinterpret_cast<uint32_t*>(dp + 5 * kPitch - 4) = ex;       // kw = 5: pixels 2 j - 2, 2 j - 1
            *reinterpret_cast<uint32_t*>(dp + 0 * kPitch) = o2;           // kw = 0: pixels 2 j, 2 j + 1
            *reinterpret_cast<uint32_t*>(dp + 4 * kPitch - 4) = o2;       // kw = 4: pixels 2 j - 2, 2 j - 1
            *reinterpret_cast<uint32_t*>(dp + 2 * kPitch) = ox;           // kw = 2: pixels 2 j, 2 j + 1 (columns c0 + 3, c0 + 5)
            *reinterpret_cast<uint32_t*>(dp + 6 * kPitch - 4) = ox;       // kw = 6: pixels 2 j - 2, 2 j - 1
          }
        }
      }
    };
    auto multiply = [&](int v) {
      const unsigned char* Ab = lA + (v & 1) * kABuf + a_off0;
      const int slot = (2 * v - 3 + n_kh + 18) % kRing;
      const unsigned char* Bb = nreal ? lB + slot * kBSlot + n_cikw * kPitch + frag_off : smem + kZeroRow + khalf * 16;
      const int bstep = nreal ? 32 : 0, bplane = nreal ? kBPlane : 0;
#pragma unroll
      for (int ks = 0; ks < 3; ++ks) {
        const f16x8 Bhi = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Bb + ks * bstep));
        const f16x8 Blo = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Bb + bplane + ks * bstep));
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const f16x8 Ahi = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Ab + m * 32 * kPitch + ks * 32));
          const f16x8 Alo = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(Ab + kAPlane + m * 32 * kPitch + ks * 32));
          am[m] = hs_mfma(Ahi, Bhi, am[m]);
          al[m] = hs_mfma(Ahi, Blo, al[m]);
          al[m] = hs_mfma(Alo, Bhi, al[m]);
        }
      }
    };
    // iteration v: fetch target v + 1, multiply output row v (v >= oy0: the four iterations before only fill the window),
    // convert target v (fetched one iteration ago), barrier
    const int vs = oy0 - 4;
    fetch(vs, 0);
    for (int v = vs; v < oy1; v += 2) {
      fetch(v + 1, 1);
      if (v >= oy0) multiply(v);
      convert(v, 0);
      __syncthreads();
      if (v + 1 < oy1) {
        fetch(v + 2, 0);
        if (v + 1 >= oy0) multiply(v + 1);
        convert(v + 1, 1);
        __syncthreads();
      }
    }
  }

  // ---- the two k halves of a tile meet in LDS, then [64 co][147] floats are added to dW with coalesced atomics ----
  float* tbuf = reinterpret_cast<float*>(smem);            // [64 co][160]
  __syncthreads();
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    if (kp == pass) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
          const float val = (am[m][r] + al[m][r] * (1.f / kLo)) * xs_inv;
          float* dst = tbuf + co * 160 + ncol;
          *dst = pass == 0 ? val : *dst + val;
        }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < 64 * 147; idx += kNT) {
    const int co = idx / 147, nn = idx - co * 147;
    const int kh = nn / 21, cikw = nn - kh * 21, ci = cikw / 7, kw = cikw - ci * 7;
    atomicAdd(a.dw + ((co * 3 + ci) * 7 + kh) * 7 + kw, tbuf[co * 160 + nn]);
  }
}

}  // namespace

bool conv2d_wgrad_stem_hs_eligible(int Cin, int Cout, int k, int stride, int pad) {
  const bool exact = debug_switches().conv_exact || debug_switches().wgrad_exact;
  return !exact && Cin == 3 && Cout == 64 && k == 7 && stride == 2 && pad == 3;
}

int conv2d_wgrad_stem_hs(const float* x, const float* dy, float* dw, int N, int H, int W, const uint32_t* dy_amax, int dy_amax_n,
                         hipStream_t s) {
  ADX_REQUIRE(x && dy && dw, "conv2d_wgrad_stem_hs: null tensor");
  WgradStemArgs a{};
  a.x = x; a.dy = dy; a.dw = dw; a.dy_amax = dy_amax; a.dy_amax_n = dy_amax_n;
  a.N = N; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, 7, 2, 3); a.OW = conv_out_dim(W, 7, 2, 3);
  ADX_REQUIRE((size_t)64 * a.OH * a.OW * sizeof(float) < 0x7FFFFFFFu && (size_t)3 * H * W * sizeof(float) < 0x7FFFFFFFu,
              "conv2d_wgrad_stem_hs: image too large for 31-bit offsets");
  a.segs = ceil_div(a.OW, kNPX);
  // units of ~32 output rows: enough of them to give every CU several (four window-filling iterations per unit)
  a.rows_per_unit = a.OH >= 64 ? 32 : a.OH;
  a.row_chunks = ceil_div(a.OH, a.rows_per_unit);
  a.units = N * a.segs * a.row_chunks;
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_wgrad_stem_hs_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLds));
    once.commit();
  }
  static_assert(kLds <= 160 * 1024 && (size_t)64 * 160 * sizeof(float) <= kLds, "LDS budget");
  const int grid = std::min(a.units, 256);
  conv2d_wgrad_stem_hs_kernel<<<dim3((unsigned)grid), dim3(kNT), kLds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
