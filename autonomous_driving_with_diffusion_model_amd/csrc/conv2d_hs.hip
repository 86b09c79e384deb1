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
// Range: |x| must stay below 65504 (fp16 max); larger values become inf/NaN loudly, and elements below 2^-14 keep
// an absolute (not relative) accuracy of 2^-36.  The ResNet's normalised image and batch-normalised activations
// are O(1..100).  Data gradients are not: the training executor hands their max|x| (Conv2dArgs::x_amax, computed
// by the kernel that produced them) and the kernel moves them into range by an exact power of two.
//
// Tile: one workgroup = 4 waves = 8 output rows x 32 output columns x 64 output channels; wave w owns rows
// 2w, 2w+1.  Per 16-channel chunk the input patch is split while it is staged (once per staged element) into
// LDS as [k-half][plane][pixel] 16-byte cells of 8 channels -- exactly the B fragment of one lane -- and the
// weight slab arrives pre-split from adx_*_pack as [tap][plane][k-half][cout] cells (the A fragment), so every
// operand read is one conflict-free ds_read_b128.  Global loads of the next stage are issued before the MFMAs
// of the current one and land in the other LDS buffer after them (one barrier per stage).  Epilogue as conv2d.hip: BN scale/shift, residual, ReLU, NCHW stores.
#include <type_traits>
#include <stdlib.h>

#include "adx_common.h"
#include "conv2d_internal.h"
#include "conv2d_hs_common.h"

namespace adx {

// Cell-layout store of one lane's 16 accumulator values of a 32-channel group (v[r]: channel (r & 3) + 8 (r >> 2) + 4 khalf
// of the lane's pixel; lane + 32 holds the other halves of the same cells).  Four v_permlane32_swap per pair of cells
// leave lanes 0-31 with cells 0 and 2 of the group and lanes 32-63 with cells 1 and 3, eight channels each (v[8 i + 0..7] =
// cell 2 i + khalf); then affine (sc / sh: LDS, indexed by the channel within the group's 64-channel slab), optional
// residual (res8[i]: the eight values of cell 2 i + khalf, or null), ReLU, split, two 16-byte stores per cell.
// so: SGPR byte offset of the group's first hi cell; cplane: bytes of one plane of cells; vcell: the lane's pixel * 16 +
// khalf * 2 * cplane, or the out-of-range offset.
template <bool AFFINE = true>
__device__ __forceinline__ void cells_store32(float (&v)[16], const float* sc, const float* sh, int cl0, const float (*res8)[8],
                                              bool relu, __amdgpu_buffer_rsrc_t yrsrc, uint32_t vcell, uint32_t so, uint32_t cplane) {
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const auto sw = __builtin_amdgcn_permlane32_swap(f2u(v[8 * i + j]), f2u(v[8 * i + 4 + j]), false, false);
      v[8 * i + j] = u2f(sw[0]);
      v[8 * i + 4 + j] = u2f(sw[1]);
    }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    float o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = AFFINE ? v[8 * i + j] * sc[cl0 + 16 * i + j] + sh[cl0 + 16 * i + j] : v[8 * i + j];
      if (res8 != nullptr) t += res8[i][j];
      o[j] = relu ? __builtin_fmaxf(t, 0.f) : t;
    }
    u32x4 hi, lo;
    split8(o, 1.f, hi, lo);
    const uint32_t sc2 = so + (uint32_t)(2 * i) * 2u * cplane;
    __builtin_amdgcn_raw_buffer_store_b128(hi, yrsrc, vcell, sc2, 0);
    __builtin_amdgcn_raw_buffer_store_b128(lo, yrsrc, vcell, sc2 + cplane, 0);
    // A VALU write to the data registers of a 16-byte buffer store in the next issue slot can reach the store (seen on
    // gfx950: one dword of ~1e-4 of the cells, run to run different); the compiler's hazard recogniser inserts the wait
    // state only for stores WITHOUT an SGPR offset, these have one.  The statement below keeps both data operands alive
    // across one wait state, wherever the scheduler puts it: nothing can write them before it.
    asm volatile("s_nop 0" ::"v"(hi), "v"(lo));
  }
}

// STRIDE/K: the convolution; ROWS: output rows per wave (tile = 4*ROWS rows x 32 columns x 64 channels);
// PBUF: LDS copies of the patch (2: one barrier per stage; 1: an extra barrier per chunk, for the large stride-2
// patches); DS: also evaluate the BasicBlock's 1x1 stride-2 downsample conv (modeling/resnet.py:223-232) on the
// centre tap's operand fragments -- same input pixels, its own weights / BN / output tensor.
template <int STRIDE, int K, int ROWS, int PBUF, bool DS, bool XCELLS = false, bool YCELLS = false>
__global__ void __launch_bounds__(256, 2) conv2d_hs_kernel(const Conv2dArgs a) {
  static_assert(!DS || (K == 3 && STRIDE == 2 && PBUF == 1), "the fused downsample rides on the 3x3 stride-2 conv");
  constexpr int TH = 4 * ROWS;
  constexpr int PH = (TH - 1) * STRIDE + K;
  constexpr int PW = (kTileW - 1) * STRIDE + K;
  constexpr int EVW = (PW + 1) / 2;               // stride 2: a patch row is stored as [even columns][odd columns]
  constexpr int PLANE = PH * PW;                  // pixels of the staged patch
  constexpr int NITEM = 2 * PLANE;                // (k-half, pixel) cells per chunk
  constexpr int PIT = (NITEM + 255) / 256;
  constexpr int NTAPS = K * K;
  constexpr int NW = NTAPS * 256;                 // 16-byte weight cells per chunk: [tap][plane][k-half][64]
  constexpr int WST = K * 256;                    // weight cells of one stage (= one kernel row kh)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* patch = reinterpret_cast<u32x4*>(smem_raw);   // PBUF x [k-half][plane][PLANE]
  u32x4* wl = patch + PBUF * 4 * PLANE;                // 2 x [kw][plane][k-half][64]
  u32x4* wds = wl + 2 * WST;                           // DS: [plane][k-half][64]
  float* ss = reinterpret_cast<float*>(wds + (DS ? 256 : 0));   // scale[64], shift[64] (+ the downsample's)
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
  const u32x4* wdsrc = DS ? reinterpret_cast<const u32x4*>(a.w_ds) + (size_t)ct * nchunks * 256 : nullptr;

  // gather offset (bytes, from the chunk's first channel plane) of cell k's first channel.  The patch is read with
  // buffer loads: SGPR descriptor + 32-bit VGPR offset + SGPR channel offset (no 64-bit address math, PIT offset
  // registers in all), and cells outside the image use an offset beyond the descriptor's extent, which the
  // hardware range check turns into zeros -- the convolution's zero padding.
  constexpr uint32_t kOutside = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xin), 0, (int)((size_t)a.Cin * hw * sizeof(float)), 0x00020000);
  const uint32_t plane_bytes = (uint32_t)(hw * sizeof(float));
  uint32_t goff[PIT];
  int pcell[PIT];        // LDS cell of item k inside its [k-half][plane] image
#pragma unroll
  for (int k = 0; k < PIT; ++k) {
    const int e = tid + 256 * k;
    const int hg = e >= PLANE ? 1 : 0;
    const int p = e - hg * PLANE;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool ok = e < NITEM && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
    goff[k] = !ok ? kOutside
                  : XCELLS ? (uint32_t)(hg * 8 * hw * sizeof(float) + ((size_t)iy * a.W + ix) * 16)   // cell layout (conv2d_hs3x3_kernel)
                           : (uint32_t)((hg * 8 * hw + (size_t)iy * a.W + ix) * sizeof(float));
    pcell[k] = hg * 2 * PLANE + (STRIDE == 2 ? py * PW + (px & 1) * EVW + (px >> 1) : p);
  }
  if (tid < 2 * kHsCout) {
    const int c = cout0 + (tid & (kHsCout - 1));
    ss[tid] = a.scale == nullptr ? (tid < kHsCout ? 1.f : 0.f) : (tid < kHsCout ? a.scale[c] : a.shift[c]);
    if (DS) ss[2 * kHsCout + tid] = a.scale_ds == nullptr ? (tid < kHsCout ? 1.f : 0.f) : (tid < kHsCout ? a.scale_ds[c] : a.shift_ds[c]);
  }

  // optional dynamic range: scale x so that max|x| lands in [2^14, 2^15), undone exactly in the epilogue
  float xs = 1.f, xs_inv = 1.f;
  if (a.x_amax != nullptr) {
    uint32_t* red = reinterpret_cast<uint32_t*>(ss + (DS ? 4 : 2) * kHsCout);
    uint32_t b = 0;
    for (int i = tid; i < a.x_amax_n; i += 256) b = a.x_amax[i] > b ? a.x_amax[i] : b;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
      b = o > b ? o : b;
    }
    if (lane == 0) red[wave] = b;
    __syncthreads();
    b = red[0] > red[1] ? red[0] : red[1];
    b = red[2] > b ? red[2] : b;
    b = red[3] > b ? red[3] : b;
    const int e = (int)((b >> 23) & 0xFF);
    if (e != 0 && e != 255) {
      int sh = 127 + 14 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      xs = __builtin_bit_cast(float, (uint32_t)(127 + sh) << 23);
      xs_inv = __builtin_bit_cast(float, (uint32_t)(127 - sh) << 23);
    }
  }

  constexpr int NDS = DS ? ROWS : 0;
  f32x16 accm[ROWS][2], accl[ROWS][2];   // [row][cout half]: hi*hi sums, cross-term sums (scaled by 2^11)
  f32x16 adm[NDS + 1][2], adl[NDS + 1][2];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        accm[r][m][i] = 0.f; accl[r][m][i] = 0.f;
        if (DS) { adm[r][m][i] = 0.f; adl[r][m][i] = 0.f; }
      }

  // Pipeline stage = (chunk, kernel row kh): weights are double-buffered per stage, the patch per chunk, so one
  // barrier per stage is enough and only K weight cells + PIT patch cells per thread are ever in registers.
  float pv[(PIT == K && PBUF == 2) ? 1 : PIT][8];
  u32x4 wv[K];
  u32x4 wdv;
  // the patch of the next chunk travels in K slices, one per stage (round k of the cell list <-> stage kh), so only
  // one slice (8 registers) is in flight at a time; with PIT != K everything goes with the first stage
  constexpr bool kSliced = (PIT == K) && PBUF == 2;
  auto load_p = [&](int chunk, int k0, int k1) {
    const uint32_t cbase = (uint32_t)chunk * kHsCC * plane_bytes;
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      if (XCELLS) {
        const u32x4 h4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[k], cbase, 0);
        const u32x4 l4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[k], cbase + 4 * plane_bytes, 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          pv[kSliced ? 0 : k][j] = u2f(h4[j]);
          pv[kSliced ? 0 : k][4 + j] = u2f(l4[j]);
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j)
          pv[kSliced ? 0 : k][j] =
              __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, goff[k], cbase + j * plane_bytes, 0));
      }
    }
    if (DS && k0 == 0) wdv = wdsrc[(size_t)chunk * 256 + tid];
  };
  auto store_p = [&](int buf, int k0, int k1) {
    u32x4* pd = patch + buf * 4 * PLANE;
#pragma unroll
    for (int k = k0; k < k1; ++k) {
      const int e = tid + 256 * k;
      if (PIT * 256 == NITEM || e < NITEM) {
        u32x4 hi, lo;
        if (XCELLS) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            hi[j] = __builtin_bit_cast(uint32_t, pv[kSliced ? 0 : k][j]);
            lo[j] = __builtin_bit_cast(uint32_t, pv[kSliced ? 0 : k][4 + j]);
          }
        } else {
          split8(pv[kSliced ? 0 : k], xs, hi, lo);
        }
        pd[pcell[k]] = hi;
        pd[pcell[k] + PLANE] = lo;
      }
    }
    if (DS && k0 == 0) wds[tid] = wdv;
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

  const int pb_lane = khalf * 2 * PLANE + (wave * ROWS * STRIDE) * PW + (STRIDE == 2 ? l31 : l31 * STRIDE);
  const int wa_lane = khalf * 64 + l31;
  const int nstages = nchunks * K;

  load_w(0);
  store_w(0);
  if (kSliced) {
#pragma unroll
    for (int k = 0; k < PIT; ++k) { load_p(0, k, k + 1); store_p(0, k, k + 1); }
  } else {
    load_p(0, 0, PIT);
    store_p(0, 0, PIT);
  }
  __syncthreads();
  struct Frags { f16x8 A[2][2], B[2][ROWS]; };   // [plane][cout half], [plane][row]
  int stage = 0;
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const u32x4* pb0 = patch + (PBUF == 2 ? (chunk & 1) * 4 * PLANE : 0) + pb_lane;
#pragma unroll
    for (int kh = 0; kh < K; ++kh, ++stage) {
      const u32x4* wa0 = wl + (stage & 1) * WST + wa_lane;
      if (stage + 1 < nstages) load_w(stage + 1);
      if (chunk + 1 < nchunks) {
        if (kSliced) load_p(chunk + 1, kh, kh + 1);
        else if (kh == 0) load_p(chunk + 1, 0, PIT);
      }
      auto fetch = [&](Frags& f, int kw) {
        const int col = STRIDE == 2 ? (kw & 1) * EVW + (kw >> 1) : kw;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int m = 0; m < 2; ++m) f.A[pl][m] = __builtin_bit_cast(f16x8, wa0[(kw * 2 + pl) * 128 + m * 32]);
#pragma unroll
          for (int r = 0; r < ROWS; ++r) f.B[pl][r] = __builtin_bit_cast(f16x8, pb0[pl * PLANE + (r * STRIDE + kh) * PW + col]);
        }
      };
      Frags fr[2];               // fragments of tap kw+1 are fetched under the MFMAs of tap kw
      fetch(fr[0], 0);
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        const Frags& f = fr[kw & 1];
        if (kw + 1 < K) fetch(fr[(kw + 1) & 1], kw + 1);
        __builtin_amdgcn_sched_barrier(0);     // the next tap's reads are issued before this tap's MFMAs, in this order
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            accm[r][m] = hs_mfma(f.A[0][m], f.B[0][r], accm[r][m]);
            accl[r][m] = hs_mfma(f.A[0][m], f.B[1][r], accl[r][m]);
            accl[r][m] = hs_mfma(f.A[1][m], f.B[0][r], accl[r][m]);
          }
        if (DS && kh == 1 && kw == 1) {   // x[2 oy][2 ox]: the 1x1 stride-2 conv's only tap
          f16x8 D[2][2];
#pragma unroll
          for (int pl = 0; pl < 2; ++pl)
#pragma unroll
            for (int m = 0; m < 2; ++m) D[pl][m] = __builtin_bit_cast(f16x8, wds[pl * 128 + wa_lane + m * 32]);
#pragma unroll
          for (int r = 0; r < ROWS; ++r)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
              adm[r][m] = hs_mfma(D[0][m], f.B[0][r], adm[r][m]);
              adl[r][m] = hs_mfma(D[0][m], f.B[1][r], adl[r][m]);
              adl[r][m] = hs_mfma(D[1][m], f.B[0][r], adl[r][m]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      if (stage + 1 < nstages) store_w((stage + 1) & 1);
      if (chunk + 1 < nchunks) {
        if (kSliced) {
          store_p((chunk + 1) & 1, kh, kh + 1);
        } else if (kh == K - 1) {
          if (PBUF == 1) __syncthreads();      // every wave is done with the only patch copy
          store_p(PBUF == 2 ? (chunk + 1) & 1 : 0, 0, PIT);
        }
      }
      __syncthreads();
    }
  }

  // ---- epilogue: combine, BN scale/shift, residual, ReLU; lane = pixel column, register = channel ----
  // All residual loads of the tile are issued up front (the staging registers are dead by now): one exposed
  // latency per tile instead of one per 16 values.
  // Buffer descriptors: one instruction per access (wave-uniform channel offset in an SGPR, the lane's pixel in one
  // 32-bit VGPR); lanes outside the map carry the out-of-range offset (loads return 0, stores are dropped); without a
  // residual the descriptor is empty and every load returns 0.
  const int ox = ox0 + l31;
  // plain store: channel cout0 + c of an OH x OW map.  Depth-to-space store (Conv2dArgs::d2s_cin): this workgroup's 64
  // channels belong to ONE parity class (d2s_cin is a multiple of 64), i.e. to channels ci0 .. ci0 + 63 of a d2s_h x d2s_w map
  // at the pixels (2 oy + py, 2 ox + px)
  const bool d2s = !DS && a.d2s_cin > 0;
  const int cls = d2s ? cout0 / a.d2s_cin : 0, py = cls >> 1, px = cls & 1;
  const int sH = d2s ? a.d2s_h : a.OH, sW = d2s ? a.d2s_w : a.OW, sC = d2s ? a.d2s_cin : a.Cout;
  const uint32_t plane_ob = (uint32_t)(sH * sW) * (uint32_t)sizeof(float);
  const size_t img = (size_t)n * sC * sH * sW;
  const int img_bytes = (int)(sC * plane_ob);
  constexpr uint32_t kOut = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(DS ? a.y_ds + img : a.y, 0, DS ? img_bytes : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res != nullptr ? a.res + img : a.y), 0, a.res != nullptr ? img_bytes : 0, 0x00020000);
  const uint32_t cbase_o = (uint32_t)(d2s ? cout0 - cls * a.d2s_cin : cout0) * plane_ob;
  if constexpr (YCELLS) {
    // both outputs as cell tensors (conv2d_hs3x3_kernel reads them: conv2 of the block takes y as its input and y_ds as its
    // residual); no residual and no depth-to-space store on this path (the host checks)
    const uint32_t cplane = (uint32_t)(a.OH * a.OW) * 16u;
    const uint32_t cell0 = (uint32_t)(cout0 >> 3) * 2u * cplane;
#pragma unroll
    for (int rr = 0; rr < ROWS; ++rr) {
      const int oy = oy0 + wave * ROWS + rr;
      const uint32_t vcell = (oy < a.OH && ox < a.OW) ? (uint32_t)(oy * a.OW + ox) * 16u + (uint32_t)khalf * 2u * cplane : kOut;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = (accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
        cells_store32(v, ss, ss + kHsCout, half * 32 + 8 * khalf, nullptr, a.relu != 0, yrsrc, vcell,
                      cell0 + (uint32_t)(half * 4) * 2u * cplane, cplane);
        if (DS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = (adm[rr][half][r] + adl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
          cells_store32(v, ss + 2 * kHsCout, ss + 3 * kHsCout, half * 32 + 8 * khalf, nullptr, false, drsrc, vcell,
                        cell0 + (uint32_t)(half * 4) * 2u * cplane, cplane);
        }
      }
    }
    return;
  }
  uint32_t voff[ROWS];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr) {
    const int oy = oy0 + wave * ROWS + rr;
    const int yy = d2s ? 2 * oy + py : oy, xx = d2s ? 2 * ox + px : ox;
    voff[rr] = (oy < a.OH && ox < a.OW && yy < sH && xx < sW) ? (uint32_t)(yy * sW + xx) * 4u + (uint32_t)(4 * khalf) * plane_ob : kOut;
  }
  float rv[ROWS][2][16];
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = half * 32 + (r & 3) + 8 * (r >> 2);     // + 4 * khalf, which rides in voff
        rv[rr][half][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, voff[rr], cbase_o + cu * plane_ob, 0));
      }
#pragma unroll
  for (int rr = 0; rr < ROWS; ++rr)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = half * 32 + (r & 3) + 8 * (r >> 2);
        const int cl = cu + 4 * khalf;
        float v = (accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
        v = v * ss[cl] + ss[kHsCout + cl];
        v += rv[rr][half][r];
        if (a.relu) v = v > 0.f ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrsrc, voff[rr], cbase_o + cu * plane_ob, 0);
        if (DS) {    // downsample branch: BN only (resnet.py:230-231), no ReLU, no residual
          const float d = (adm[rr][half][r] + adl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, d * ss[2 * kHsCout + cl] + ss[3 * kHsCout + cl]),
                                                drsrc, voff[rr], cbase_o + cu * plane_ob, 0);
        }
      }
}

// ---- 3x3 stride-1, deferred-store pipeline ---------------------------------------------------------------------
// Same LDS images and arithmetic as conv2d_hs_kernel<1,3,2,2,false>; what changes is WHEN staged data moves and how
// much of it a CU needs per MFMA.
//  * There the loads of stage s+1 are issued at the start of stage s and converted + written to LDS after its MFMAs,
//    so every stage ends with a serial tail (wait for HBM, ~60 VALU, 5 LDS writes, LDS drain, barrier).  Here data is
//    fetched TWO stages ahead into a second register set, and the set that arrived during the previous stage is split
//    and written between the MFMAs of this stage; a stage ends with just the LDS drain and the barrier.  Two chunks
//    (6 stages) are unrolled so that every register-set index is a constant, and the stage body has no branches:
//    with control flow the compiler's s_waitcnt bookkeeping merges states and every wait degrades to vmcnt(0).
//  * Measured on the 4-wave tile (SQ counters, tools/pmc_sq.sh; 512->512 layer): SQ_VALU_MFMA_BUSY 53 % at an
//    effective 1.98 GHz, i.e. ~1040 TFLOP/s of fp16 MFMA; ONE workgroup per CU is as fast as two, removing the LDS
//    fragment reads changes 7 %, and tiles that move 18-35 % fewer operand bytes per MFMA (MODE 1 / 2) change
//    nothing: the loop runs at the MFMA rate the chip sustains on random data (MI355X_MICROARCH.md quotes 1247
//    TFLOP/s for a tuned bf16 GEMM at 1.9-1.95 GHz), not at a rate set by operand delivery.  With all-zero operands
//    the same launches are 22-25 % faster (ZERO=1 tools/bench_conv.py): the limit is power, i.e. data-dependent
//    clock-down, and what shortens a launch is fewer MFMAs / instructions / bytes, not fewer stalls.  Tile modes:
//      MODE 0: 4 waves,  8 rows x 32 columns x  64 channels (two per CU)  19.5 KB per 36 MFMAs/wave
//      MODE 1: 8 waves, 16 rows x 32 columns x  64 channels               25.4 KB per 2 x 36   (-35 %)
//      MODE 2: 8 waves,  8 rows x 32 columns x 128 channels               31.9 KB per 2 x 36   (-18 %; 8-row maps)
#ifdef ADX_HS_TRACE
// diagnostic build (-DADX_HS_TRACE): every workgroup of conv2d_hs3x3_kernel leaves the shader clock at its phase
// boundaries and the hardware slot it ran on; tools/hs_trace.py reads them back through adx_hs_trace_read
__device__ unsigned long long g_hs_trace[8 * 16384];
__device__ __forceinline__ void hs_trace(int slot) {
  if (threadIdx.x == 0 && blockIdx.x < 16384) g_hs_trace[blockIdx.x * 8 + slot] = __builtin_readcyclecounter();
}
__device__ __forceinline__ void hs_trace_id() {
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    g_hs_trace[blockIdx.x * 8 + 7] = ((unsigned long long)xcc << 32) | hw;
  }
}
#define HS_TRACE(slot) hs_trace(slot)
#else
#define HS_TRACE(slot)
#endif

// XCELLS / YCELLS: the input / output tensor is in the CELL layout instead of fp32 NCHW -- per image [C / 8][plane: hi, lo][H][W]
// cells of 16 bytes = the 8 channels of one pixel already split into fp16 hi / lo, i.e. exactly what the staging writes to
// LDS.  Same bytes per tensor, but the consumer's staging is two 16-byte loads + two LDS writes per cell pair where the fp32
// layout costs eight 4-byte loads and ~48 VALU instructions, once per 64-channel output slab; the producer splits each
// element once.  The products are bit-identical either way (the halves are the ones the consumer would have computed); a
// residual read from cells (Conv2dArgs::res_cells) is hi + lo / 2^11, the tensor to 2^-23.  Inference executor only.
// STATS: 0 none; 1 the training forward's BatchNorm statistics of this conv's own output; 2 (data-gradient launches) the
// BACKWARD sums of the BatchNorm in front of this conv in forward order, whose incoming gradient this launch produces
// (Conv2dArgs::bs_*).
// VR: the column tiles run over the virtual row of the whole batch (below) although the OUTPUT is fp32 NCHW -- the training
// forward and data-gradient launches (round 5): a map 225 / 113 / 57 / 29 columns wide pays one padded MFMA column per image
// instead of the round-up to 32, like the cell launches; YCELLS implies it.
template <int MODE, int STATS = 0, bool XCELLS = false, bool YCELLS = false, bool VR = false>
__global__ void __launch_bounds__(MODE == 0 ? 256 : 512, 2) conv2d_hs3x3_kernel(const Conv2dArgs a) {
  constexpr bool VROW = YCELLS || VR;
  constexpr int NT = MODE == 0 ? 256 : 512;            // threads
  constexpr int TH = MODE == 1 ? 16 : 8;               // output rows per workgroup
  constexpr int CT = MODE == 2 ? 2 : 1;                // 64-channel slabs per workgroup
  constexpr int K = 3, PH = TH + 2, PW = 34, PLANE = PH * PW, NITEM = 2 * PLANE;
  constexpr int PIT = (NITEM + NT - 1) / NT;           // patch rounds = slices, one per stage while they last
  constexpr int NW = 9 * 256, WST = 3 * 256 * CT;      // weight cells per chunk and 64-channel slab / per stage
  constexpr int WIT = (WST + NT - 1) / NT;
  static_assert(PIT <= K, "a patch slice travels with each stage of a chunk");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* patch = reinterpret_cast<u32x4*>(smem_raw);   // 2 x [k-half][plane][PLANE]
  u32x4* wl = patch + 2 * 4 * PLANE;                   // 2 x [slab][kw][plane][k-half][64]
  float* ss = reinterpret_cast<float*>(wl + 2 * WST);  // scale[64 CT], shift[64 CT], 8 words for the range reduction
  u32x4* dummy = reinterpret_cast<u32x4*>(ss + 2 * 64 * CT + 8);   // where idle threads of a partial round write
  float* bsl = reinterpret_cast<float*>(dummy + 3);                 // STATS == 2: [mean | rstd | mask scale | mask shift] x 64 CT
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rowpair = MODE == 2 ? (wave & 3) : wave, slab = MODE == 2 ? (wave >> 2) : 0;
#ifdef ADX_HS_TRACE
  hs_trace_id();
#endif
  HS_TRACE(0);
  int bid = blockIdx.x;
  {
    const int per = gridDim.x >> 3;
    if (bid < per * 8) bid = (bid & 7) * per + (bid >> 3);
  }
  int kpart = 0;                       // split reduction (Conv2dArgs::ksplit): this workgroup's share of the chunks
  if (a.ksplit > 1) { kpart = bid % a.ksplit; bid /= a.ksplit; }
  const int ct = bid % a.cout_tiles; bid /= a.cout_tiles;
  const int tx = bid % a.tiles_x; bid /= a.tiles_x;
  const int ty = bid % a.tiles_y; bid /= a.tiles_y;
  const int n = VROW ? 0 : bid;
  const int oy0 = ty * TH, ox0 = tx * kTileW;
  const int iy0 = oy0 - a.pad, ix0 = ox0 - a.pad;
  const int cout0 = (ct * CT + slab) * kHsCout;
  const int l31 = lane & 31, khalf = lane >> 5;
  // YCELLS (the inference executor's launches): column tiles run over a VIRTUAL row -- the images of the batch side by side,
  // vw = W + 1 columns each: the W real ones and ONE all-zero column, which is the right padding of its image and the left
  // padding of the next one (pad = 1) -- so a map 225 / 113 / 57 / 29 wide pays one padded MFMA column per image instead of
  // 31 / 15 / 7 / 3: 452 tiles instead of 512 per tile row at B = 64.  The staged patch row is the 34 consecutive virtual
  // columns around the tile, whatever images they belong to, and the lane of virtual column v reads cell (lane + kw) exactly
  // as in the per-image tiling: consecutive lanes, consecutive 16-byte cells, no bank conflict.  (Round 3 gave every image
  // segment of a tile its own two halo cells, i.e. lanes behind an image boundary read two cells further: a 2-way conflict in
  // nearly every 16-lane group of a tile that holds a boundary -- 25 % of the LDS cycles of the 512-channel layers.)
  const int vx0 = tx * kTileW;
  const int vl = vx0 + l31;
  auto vdiv = [&](int v) { return (int)(((float)v + 0.5f) * a.inv_vw); };     // v / vw for 0 <= v < 2^21 (launch check)
  const int nl = VROW ? vdiv(vl) : 0, xl = VROW ? vl - nl * a.vw : 0;
  const bool lane_valid = !VROW || (nl < a.N && xl < a.W);
  const size_t hw = (size_t)a.H * a.W;
  const int nchunks_all = a.cin_pad / kHsCC;
  const int nchunks = a.ksplit > 1 ? a.cper : nchunks_all;       // chunks THIS workgroup reduces over
  const int chunk0 = kpart * nchunks;
  const float* xin = a.x + ((size_t)n * a.Cin + (size_t)chunk0 * kHsCC) * hw;      // YCELLS: n = 0, chunk0 = 0: the whole tensor
  const int nstages = nchunks * K;
  const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w) + ((size_t)ct * CT * nchunks_all + chunk0) * NW;
  constexpr uint32_t kOutside = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(xin), 0,
      (int)(uint32_t)((VROW ? (size_t)a.N * a.Cin : (size_t)(a.Cin - chunk0 * kHsCC)) * hw * sizeof(float)), 0x00020000);
  const uint32_t plane_bytes = (uint32_t)(hw * sizeof(float));
  uint32_t goff[PIT];
  int pcell[PIT];
#pragma unroll
  for (int k = 0; k < PIT; ++k) {
    const int e = tid + NT * k;
    const int hg = e >= PLANE ? 1 : 0;
    const int p = e - hg * PLANE;
    const int py = p / PW, px = p - py * PW;
    const int iy = iy0 + py;
    int ix = ix0 + px, ni = n;
    if constexpr (VROW) {          // patch column px = virtual column vx0 - 1 + px -> (image, input column); column W of an image is zero
      const int v = vx0 - 1 + px;
      ni = vdiv(v < 0 ? 0 : v);
      ix = v - ni * a.vw;          // -1 for the column left of the first image
    }
    const bool ok = e < NITEM && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W && ni < a.N;
    const size_t ioff = VROW ? (size_t)ni * a.Cin * hw * sizeof(float) : 0;
    goff[k] = !ok ? kOutside
                  : XCELLS ? (uint32_t)(ioff + hg * 8 * hw * sizeof(float) + ((size_t)iy * a.W + ix) * 16)   // cell (2 chunk + hg, hi, iy, ix)
                           : (uint32_t)(ioff + (hg * 8 * hw + (size_t)iy * a.W + ix) * sizeof(float));
    pcell[k] = e < NITEM ? hg * 2 * PLANE + p : -1;
  }
  // weight cells of this thread: global offset inside a stage (slab-major) and LDS cell, or the dummy
  int wsrc_off[WIT], wdst[WIT];
#pragma unroll
  for (int k = 0; k < WIT; ++k) {
    const int e = tid + NT * k;
    const int sl = e / 768, within = e - sl * 768;
    const bool ok = e < WST;
    wsrc_off[k] = ok ? sl * nchunks_all * NW + within : 0;
    wdst[k] = ok ? e : -1;
  }
  // BN scale / shift of this workgroup's channels: requested now, parked in LDS after the first stage's data (a wait
  // here would put one more memory round trip in front of the first patch load)
  float ssv = 0.f;
  if (tid < 2 * 64 * CT) {
    const int half = tid / (64 * CT), cc = tid - half * 64 * CT;
    const int c = ct * CT * kHsCout + cc;
    ssv = a.scale == nullptr ? (half == 0 ? 1.f : 0.f) : (half == 0 ? a.scale[c] : a.shift[c]);
  }
  float bsv = 0.f;
  if (STATS == 2 && tid < 4 * 64 * CT) {
    const int which = tid / (64 * CT), cc = tid - which * 64 * CT;
    const int c = ct * CT * kHsCout + cc;
    // the mask's affine form exactly as the forward pass applied it (resnet_train.hip: bn_affine)
    const float mu = a.bs_mean[c], rs = a.bs_rstd[c];
    const float sc = a.bs_gamma[c] * rs;
    bsv = which == 0 ? mu : (which == 1 ? rs : (which == 2 ? sc : __builtin_fmaf(-mu, sc, a.bs_beta[c])));
  }
  float xs = 1.f, xs_inv = 1.f;
  if (a.x_amax != nullptr && a.x_amax_n < 0) {          // a cell-layout gradient: scaled when it was written
    xs_inv = reinterpret_cast<const float*>(a.x_amax)[1];
  } else if (a.x_amax != nullptr) {
    uint32_t* red = reinterpret_cast<uint32_t*>(ss + 2 * 64 * CT);
    uint32_t b = 0;
    for (int i = tid; i < a.x_amax_n; i += NT) b = a.x_amax[i] > b ? a.x_amax[i] : b;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const uint32_t o = (uint32_t)__shfl_xor((int)b, off, 64);
      b = o > b ? o : b;
    }
    if (lane == 0) red[wave] = b;
    __syncthreads();
    b = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) b = red[w] > b ? red[w] : b;
    const int e = (int)((b >> 23) & 0xFF);
    if (e != 0 && e != 255) {
      int sh = 127 + 14 - e;
      sh = sh > 100 ? 100 : (sh < -100 ? -100 : sh);
      xs = __builtin_bit_cast(float, (uint32_t)(127 + sh) << 23);
      xs_inv = __builtin_bit_cast(float, (uint32_t)(127 - sh) << 23);
    }
  }

  f32x16 accm[2][2], accl[2][2];
#pragma unroll
  for (int r = 0; r < 2; ++r)
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) { accm[r][m][i] = 0.f; accl[r][m][i] = 0.f; }

  // register sets: weights of stage s live in wv[s & 1]; patch slice g (= 3 * chunk + round) in pv[g & 1]
  u32x4 wv[2][WIT];
  float pv[2][8];
  auto load_w = [&](int stage, int set) {
    const u32x4* ws = wsrc + (size_t)stage * 768;
#pragma unroll
    for (int k = 0; k < WIT; ++k) wv[set][k] = ws[wsrc_off[k]];
  };
  auto store_w = [&](int set, int buf) {
#pragma unroll
    for (int k = 0; k < WIT; ++k) {
      u32x4* d = (WST % NT == 0 || wdst[k] >= 0) ? wl + buf * WST + wdst[k] : dummy + 2;
      *d = wv[set][k];
    }
  };
  auto load_p = [&](int chunk, int k, int set) {
    const uint32_t cbase = (uint32_t)chunk * kHsCC * plane_bytes;
    if (XCELLS) {           // the hi cell and, one plane (H W cells) further, the lo cell
      const u32x4 h4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[k], cbase, 0);
      const u32x4 l4 = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, goff[k], cbase + 4 * plane_bytes, 0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        pv[set][j] = u2f(h4[j]);
        pv[set][4 + j] = u2f(l4[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j)
        pv[set][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, goff[k], cbase + j * plane_bytes, 0));
    }
  };
  auto store_p = [&](int set, int k, int buf) {
    u32x4 hi, lo;
    if (XCELLS) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        hi[j] = __builtin_bit_cast(uint32_t, pv[set][j]);
        lo[j] = __builtin_bit_cast(uint32_t, pv[set][4 + j]);
      }
    } else {
      split8(pv[set], xs, hi, lo);
    }
    u32x4* pd = patch + buf * 4 * PLANE + pcell[k];
    u32x4* d0 = pcell[k] >= 0 ? pd : dummy;
    u32x4* d1 = pcell[k] >= 0 ? pd + PLANE : dummy + 1;
    *d0 = hi;
    *d1 = lo;
  };

  const int pb_lane = khalf * 2 * PLANE + (rowpair * 2) * PW + l31;
  const int wa_lane = slab * 768 + khalf * 64 + l31;

  // prologue: stage 0 complete in LDS; weights of stage 1 and the first slice of chunk 1 in flight
  load_w(0, 0);
  store_w(0, 0);
#pragma unroll
  for (int k = 0; k < PIT; ++k) { load_p(0, k, 0); store_p(0, k, 0); }
  load_w(1, 1);                 // nchunks is even (>= 2): stage 1 and chunk 1 exist
  if (tid < 2 * 64 * CT) ss[tid] = ssv;
  if (STATS == 2 && tid < 4 * 64 * CT) bsl[tid] = bsv;
  load_p(1, 0, 1);
  HS_TRACE(1);
  __syncthreads();
  HS_TRACE(2);

  for (int cp = 0; cp < nchunks; cp += 2) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int s = 3 * cp + i;                 // global stage; s & 1 == i & 1 because cp is even
      const int kh = i % 3, cpar = (i / 3) & 1; // kernel row, parity of this stage's chunk
      const u32x4* pb0 = patch + cpar * 4 * PLANE + pb_lane;
      const u32x4* wa0 = wl + (i & 1) * WST + wa_lane;
      // fetch two stages ahead (weights of s+2, patch slice s+4) into the sets that were consumed last stage
      // (past the end the last stage / chunk is fetched again and its copy in the idle buffers is never read)
      load_w(s + 2 < nstages ? s + 2 : nstages - 1, i & 1);
      if ((i + 1) % 3 < PIT) load_p(s + 4 < nstages ? (s + 4) / 3 : nchunks - 1, (i + 1) % 3, i & 1);
      __builtin_amdgcn_sched_barrier(0);      // the fetches stay at the top of the stage: two stages of latency cover
#pragma unroll
      for (int kw = 0; kw < K; ++kw) {
        f16x8 A[2][2], B[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int m = 0; m < 2; ++m) A[pl][m] = __builtin_bit_cast(f16x8, wa0[(kw * 2 + pl) * 128 + m * 32]);
#pragma unroll
          for (int r = 0; r < 2; ++r) B[pl][r] = __builtin_bit_cast(f16x8, pb0[pl * PLANE + (r + kh) * PW + kw]);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            accm[r][m] = hs_mfma(A[0][m], B[0][r], accm[r][m]);
            accl[r][m] = hs_mfma(A[0][m], B[1][r], accl[r][m]);
            accl[r][m] = hs_mfma(A[1][m], B[0][r], accl[r][m]);
          }
        if (kw == 0) {
          // what arrived during the previous stage goes to LDS under this stage's remaining MFMAs:
          // weights of stage s+1 -> the other weight buffer, patch slice s+3 -> the next chunk's patch copy
          store_w((i + 1) & 1, (i + 1) & 1);
          if (i % 3 < PIT) store_p((i + 1) & 1, i % 3, ((i + 3) / 3) & 1);
        }
      }
      __syncthreads();
    }
  }

  HS_TRACE(3);
  // epilogue through buffer descriptors: one instruction per access (wave-uniform channel offset in an SGPR, the
  // lane's pixel in one 32-bit VGPR); lanes outside the map carry the out-of-range offset, so their loads return 0
  // and their stores are dropped; without a residual the descriptor is empty and every load returns 0
  const int ox = VROW ? xl : ox0 + l31;
  const uint32_t plane_ob = (uint32_t)(a.OH * a.OW) * (uint32_t)sizeof(float);
  const size_t img = (size_t)n * a.Cout * a.OH * a.OW;
  // YCELLS: the descriptors cover the whole tensors and the lane's image rides in its offset
  const int img_bytes = (int)(uint32_t)((VROW ? (uint32_t)a.N : 1u) * (uint32_t)a.Cout * plane_ob);
  const uint32_t img_off = VROW ? (uint32_t)nl * (uint32_t)a.Cout * plane_ob : 0u;
  float* const ybase = a.ksplit > 1 ? a.part + (size_t)kpart * a.part_stride : a.y;
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(ybase + img, 0, img_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rrsrc = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.res != nullptr ? a.res + img : a.y), 0, a.res != nullptr ? img_bytes : 0, 0x00020000);
  const float* sst = ss + slab * 64;
  const uint32_t cbase_o = (uint32_t)cout0 * plane_ob;
  const uint32_t cplane = (uint32_t)(a.OH * a.OW) * 16u;                  // one plane of cells of the output map, bytes
  const uint32_t cell0 = (uint32_t)(cout0 >> 3) * 2u * cplane;            // first cell (hi plane) of this wave's 64 channels
  bool inside[2];
  uint32_t pix[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) {
    const int oy = oy0 + rowpair * 2 + rr;
    inside[rr] = lane_valid && oy < a.OH && ox < a.OW;
    pix[rr] = (uint32_t)(oy * a.OW + ox);
  }
  // 4 channels of a cell pair: hi + lo / 2^11, one v_fma_mix_f32 per value (fp16 operands converted inside the instruction; the
  // product with a power of two is exact, so this is the same value as convert, multiply, add -- in a third of the instructions)
  auto half4 = [](uint32_t h0, uint32_t h1, uint32_t l0, uint32_t l1, float* out) {
    const float inv = 1.f / kLoScale;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(out[0]) : "v"(l0), "s"(inv), "v"(h0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(out[1]) : "v"(l0), "s"(inv), "v"(h0));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(out[2]) : "v"(l1), "s"(inv), "v"(h1));
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(out[3]) : "v"(l1), "s"(inv), "v"(h1));
  };
  if constexpr (YCELLS) {
    // cell output (cells_store32); the residual is fetched first: as cells, or as fp32 values at the channels the lane owns AFTER the swap
    const uint32_t khoff = (uint32_t)khalf * 2u * cplane;                  // this lane's cells are the odd ones: one cell further
    uint32_t vcell[2], vres[2];
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
      vcell[rr] = inside[rr] ? img_off + pix[rr] * 16u + khoff : kOutside;
      vres[rr] = !inside[rr] ? kOutside : a.res_cells ? vcell[rr] : img_off + pix[rr] * 4u + (uint32_t)(8 * khalf) * plane_ob;
    }
    uint32_t rraw[2][2][2][8];          // residual of cell (rr, half, i): 8 floats, or the hi and lo cells as they are
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (a.res_cells) {
            const uint32_t so = cell0 + (uint32_t)(half * 4 + 2 * i) * 2u * cplane;
            const u32x4 h4 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vres[rr], so, 0);
            const u32x4 l4 = __builtin_amdgcn_raw_buffer_load_b128(rrsrc, vres[rr], so + cplane, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j) { rraw[rr][half][i][j] = h4[j]; rraw[rr][half][i][4 + j] = l4[j]; }
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
              rraw[rr][half][i][j] = __builtin_amdgcn_raw_buffer_load_b32(rrsrc, vres[rr], cbase_o + (uint32_t)(half * 32 + 16 * i + j) * plane_ob, 0);
          }
        }
#ifdef ADX_HS_TRACE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    HS_TRACE(6);
#endif
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale);      // no dynamic range here: xs = 1
        float res8[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (a.res_cells) {
            half4(rraw[rr][half][i][0], rraw[rr][half][i][1], rraw[rr][half][i][4], rraw[rr][half][i][5], res8[i]);
            half4(rraw[rr][half][i][2], rraw[rr][half][i][3], rraw[rr][half][i][6], rraw[rr][half][i][7], res8[i] + 4);
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) res8[i][j] = u2f(rraw[rr][half][i][j]);
          }
        }
        cells_store32(v, sst, sst + 64 * CT, half * 32 + 8 * khalf, res8, a.relu != 0, yrsrc, vcell[rr],
                      cell0 + (uint32_t)(half * 4) * 2u * cplane, cplane);
      }
  } else {
  uint32_t voff[2];
#pragma unroll
  for (int rr = 0; rr < 2; ++rr) voff[rr] = inside[rr] ? img_off + pix[rr] * 4u + (uint32_t)(4 * khalf) * plane_ob : kOutside;
  float rv[2][2][16];
  if constexpr (STATS != 2) {
#pragma unroll
  for (int rr = 0; rr < 2; ++rr)
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cu = half * 32 + (r & 3) + 8 * (r >> 2);     // + 4 * khalf, which rides in voff
        rv[rr][half][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, voff[rr], cbase_o + cu * plane_ob, 0));
      }
  }
#ifdef ADX_HS_TRACE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HS_TRACE(6);
#endif
  if constexpr (STATS != 0) {
    // Per-channel partial sums of this workgroup's pixels (those outside the map excluded), so that a BatchNorm needs no pass
    // over the tensor this launch writes.  STATS == 1 (training forward): sum and sum of squares of the conv output.  STATS
    // == 2 (data gradient): dx = conv + residual is finished and stored HERE, and the sums are those of the BatchNorm backward
    // of the layer dx flows into: sum dz and sum dz xhat, dz = dx where that layer's ReLU let the value through.
    // Per (half, r) register: the two rows of the lane, then the 16 lanes of a DPP row; lanes 0 / 16 / 32 / 48 park their row's
    // totals in LDS (the patch is dead: every wave passed the main loop's last barrier), 64 CT x 2 threads add the 2 x NW/CT
    // rows up in a fixed order.
    float* red = reinterpret_cast<float*>(smem_raw);          // [wave][lane >> 4][32][2]
    const bool val0 = voff[0] != kOutside, val1 = voff[1] != kOutside;
    auto park = [&](int half, int r, float sm, float sq) {
      sm += hs_dpp<0xB1>(sm);  sq += hs_dpp<0xB1>(sq);      // quad_perm [1,0,3,2]
      sm += hs_dpp<0x4E>(sm);  sq += hs_dpp<0x4E>(sq);      // quad_perm [2,3,0,1]
      sm += hs_dpp<0x141>(sm); sq += hs_dpp<0x141>(sq);     // row_half_mirror
      sm += hs_dpp<0x140>(sm); sq += hs_dpp<0x140>(sq);     // row_mirror: every lane of a row of 16 holds the row's total
      if ((lane & 15) == 0) {
        float* d = red + (((wave * 4 + (lane >> 4)) * 32) + half * 16 + r) * 2;
        d[0] = sm;
        d[1] = sq;
      }
    };
    if constexpr (STATS == 1) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v0 = val0 ? (accm[0][half][r] + accl[0][half][r] * (1.f / kLoScale)) * xs_inv : 0.f;
          const float v1 = val1 ? (accm[1][half][r] + accl[1][half][r] * (1.f / kLoScale)) * xs_inv : 0.f;
          park(half, r, v0 + v1, v0 * v0 + v1 * v1);
        }
    } else {
      // the accumulators fold into accm first (accl is dead afterwards); then one 32-channel half at a time: residual, conv
      // output of the consumer BatchNorm (xhat, and its ReLU mask when the ReLU sits straight behind it) and, otherwise, that
      // layer's output (the mask) are fetched together -- 96 loads in flight -- and dx is stored from the same registers
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 16; ++r) accm[rr][half][r] = (accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
      const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.bs_raw + img), 0, img_bytes, 0x00020000);
      const bool mask_bits = a.bs_mask == 1 && a.bs_bits != nullptr;       // the forward's mask bits instead of its fp32 output
      const bool mask_out = a.bs_mask == 1 && !mask_bits;
      const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(mask_out ? a.bs_out + img : a.y), 0, mask_out ? img_bytes : 0, 0x00020000);
      // mask bits: byte [n][c / 8][pixel]; this lane's channels of a register group r >> 2 are bits 4 khalf .. 4 khalf + 3 of one byte
      const uint32_t bplane = (uint32_t)(a.OH * a.OW);
      const __amdgpu_buffer_rsrc_t brsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint8_t*>(mask_bits ? a.bs_bits + (size_t)n * (a.Cout >> 3) * bplane : reinterpret_cast<const uint8_t*>(a.y)), 0,
          mask_bits ? (int)((VROW ? (uint32_t)a.N : 1u) * (uint32_t)(a.Cout >> 3) * bplane) : 0, 0x00020000);
      uint32_t boff[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr)
        boff[rr] = inside[rr] ? (VROW ? (uint32_t)nl * (uint32_t)(a.Cout >> 3) * bplane : 0u) + pix[rr] : kOutside;
      // the residual's own mask bits (same layout): the identity path's gradient is res where the block's output was positive
      const bool res_masked = a.res_bits != nullptr;
      const __amdgpu_buffer_rsrc_t qrsrc = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<uint8_t*>(res_masked ? a.res_bits + (size_t)n * (a.Cout >> 3) * bplane : reinterpret_cast<const uint8_t*>(a.y)), 0,
          res_masked ? (int)((VROW ? (uint32_t)a.N : 1u) * (uint32_t)(a.Cout >> 3) * bplane) : 0, 0x00020000);
      const float* bsw = bsl + slab * 64;
      const uint32_t use_bit = mask_bits ? 1u : 0u, use_out = mask_out ? 1u : 0u, use_raw = 1u - use_bit - use_out;
      const uint32_t v0u = val0 ? 1u : 0u, v1u = val1 ? 1u : 0u;
      float dmx = 0.f;          // max |dz| of this lane: the BatchNorm-backward apply pass bounds its output's range with it
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float rs_[2][16], rw_[2][16], ro_[2][16];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const uint32_t so = cbase_o + (uint32_t)(half * 32 + (r & 3) + 8 * (r >> 2)) * plane_ob;
            rs_[rr][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rrsrc, voff[rr], so, 0));
            rw_[rr][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(wrsrc, voff[rr], so, 0));
            ro_[rr][r] = 0.f;
          }
        if (mask_out) {        // wave-uniform: the 32 loads of the consumer layer's output only where its ReLU follows the residual add
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const uint32_t so = cbase_o + (uint32_t)(half * 32 + (r & 3) + 8 * (r >> 2)) * plane_ob;
              ro_[rr][r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(orsrc, voff[rr], so, 0));
            }
        }
        if (res_masked) {      // wave-uniform
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const uint32_t m = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(qrsrc, boff[rr], ((uint32_t)(cout0 >> 3) + (uint32_t)(half * 4 + q)) * bplane, 0)
                                 >> (4 * khalf);
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (!((m >> i) & 1u)) rs_[rr][4 * q + i] = 0.f;
            }
        }
        uint32_t mb[2][4] = {{0u, 0u, 0u, 0u}, {0u, 0u, 0u, 0u}};
        if (mask_bits) {       // eight byte loads instead
#pragma unroll
          for (int rr = 0; rr < 2; ++rr)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              mb[rr][q] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b8(brsrc, boff[rr], ((uint32_t)(cout0 >> 3) + (uint32_t)(half * 4 + q)) * bplane, 0)
                          >> (4 * khalf);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = half * 32 + (r & 3) + 8 * (r >> 2);
          const int cl = cu + 4 * khalf;
          const float mu = bsw[cl], rsd = bsw[64 * CT + cl], msc = bsw[2 * 64 * CT + cl], msh = bsw[3 * 64 * CT + cl];
          float p[2], q[2];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const float y = accm[rr][half][r] + rs_[rr][r];            // scale 1, shift 0, no ReLU: what `finish` would store
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, y), yrsrc, voff[rr], cbase_o + cu * plane_ob, 0);
            // all three forms of the consumer's ReLU mask evaluated, one SELECTED (the choice is uniform over the launch), the
            // lane's validity folded in by `&`: as val && (bits ? .. : (out ? .. : ..)) it was four branches per value, 270 per tile
            // (as integer masks: a select between booleans on a uniform condition is turned back into a branch)
            const uint32_t on_bit = (mb[rr][r >> 2] >> (r & 3)) & 1u, on_out = ro_[rr][r] > 0.f ? 1u : 0u;
            const uint32_t on_raw = __builtin_fmaf(rw_[rr][r], msc, msh) > 0.f ? 1u : 0u;
            const bool keep = ((rr == 0 ? v0u : v1u) & ((on_bit & use_bit) | (on_out & use_out) | (on_raw & use_raw))) != 0u;
            p[rr] = keep ? y : 0.f;
            q[rr] = p[rr] * ((rw_[rr][r] - mu) * rsd);
            dmx = __builtin_fmaxf(dmx, __builtin_fabsf(p[rr]));
          }
          park(half, r, p[0] + p[1], q[0] + q[1]);
        }
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) dmx = __builtin_fmaxf(dmx, __shfl_xor(dmx, off, 64));
      if (lane == 0) red[4096 + wave] = dmx;
    }
    __syncthreads();
    if (STATS == 2 && tid >= 2 * 64 * CT && tid < 2 * 64 * CT + CT) {      // one thread per 64-channel slab: max over its waves
      const int sl = tid - 2 * 64 * CT;
      constexpr int WPS = (NT / 64) / CT;
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WPS; ++w) t = __builtin_fmaxf(t, red[4096 + (MODE == 2 ? sl * WPS + w : w)]);
      const int p = (n * a.tiles_y + ty) * a.tiles_x + tx;
      a.stats_part[(size_t)a.Cout * 2 * a.stats_p + (size_t)(ct * CT + sl) * a.stats_p + p] = t;
    }
    if (tid < 2 * 64 * CT) {
      const int which = tid & 1, chs = tid >> 1, sl = chs >> 6, ch = chs & 63;     // channel ch of slab sl
      const int c5 = ch & 31, kh2 = (c5 >> 2) & 1, r = (c5 & 3) + 4 * (c5 >> 3), idx32 = (ch >> 5) * 16 + r;
      constexpr int WPS = (NT / 64) / CT;                   // waves per slab
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WPS; ++w)
#pragma unroll
        for (int j = 0; j < 2; ++j) t += red[((((MODE == 2 ? sl * WPS + w : w) * 4 + kh2 * 2 + j) * 32) + idx32) * 2 + which];
      const int p = (n * a.tiles_y + ty) * a.tiles_x + tx;
      a.stats_part[((size_t)((ct * CT + sl) * kHsCout + ch) * 2 + which) * a.stats_p + p] = t;
    }
  }
  auto finish = [&](auto relu) {       // two copies of the store loop: the ReLU is one v_max, not a compare + select
#pragma unroll
    for (int rr = 0; rr < 2; ++rr)
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cu = half * 32 + (r & 3) + 8 * (r >> 2);
          const int cl = cu + 4 * khalf;
          float v = (accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale)) * xs_inv;
          v = v * sst[cl] + sst[64 * CT + cl];
          v += rv[rr][half][r];
          if (decltype(relu)::value) v = __builtin_fmaxf(v, 0.f);   // a NaN becomes 0, like `v > 0 ? v : 0` did
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrsrc, voff[rr], cbase_o + cu * plane_ob, 0);
        }
  };
  if constexpr (STATS != 2) {
    if (a.relu) finish(std::true_type{}); else finish(std::false_type{});
  }
  }
  HS_TRACE(4);
#ifdef ADX_HS_TRACE
  __builtin_amdgcn_s_waitcnt(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  HS_TRACE(5);
#endif
}

// ---- the stem: Conv2d(3, 64, 7, stride 2, padding 3), modeling/resnet.py:191 --------------------------------
// K = 3 channels x 7 x 7 = 147 has no 16-channel chunks, so the GEMM's k axis is laid out as 21 (channel, kernel
// row) "combos" x 8 kernel columns (7 real + one zero weight): a lane's B fragment -- 8 consecutive k for one
// output pixel -- is then 8 CONSECUTIVE input columns of one patch row, i.e. 16 contiguous bytes of the fp16 patch
// image [plane][channel][row][column] (read as 4 dwords: the start column 2*ox is only 4-byte aligned), and one
// MFMA (K = 16) covers two combos.  11 k-steps x 3 products per 32x32 tile; 84 % of the k slots are real work.
// One workgroup owns a band of 8 output rows of one image and walks its 32-column tiles: the 45 KB split weight
// image is loaded into LDS once per band, the patch of tile i+1 is fetched while tile i is multiplied.
// max with torch's NaN rule (a NaN in the window wins)
__device__ __forceinline__ float pool_max3(float a, float b, float c) {
  float m = a;
  m = (b > m || b != b) ? b : m;
  m = (c > m || c != c) ? c : m;
  return m;
}

constexpr int kStemSteps = 11;
constexpr int kStemPP = 72;                               // patch row pitch in halves (70 columns are staged)

// POOL = true additionally applies MaxPool2d(3, 2, 1) (modeling/resnet.py:197) before anything is written: the stem's
// own output (944 MB at B=64, 3x256x900 -- the largest tensor of the network) is then never stored nor re-read.  A
// workgroup produces a band of 4 pooled rows, one per wave: the wave multiplies the three stem rows of its pooled
// row one after the other (64 accumulators each, folded into a running maximum), so the vertical maximum needs no
// exchange at the price of computing the shared odd rows twice (+50 % MFMAs, still cheaper than the traffic it
// removes); the horizontal maximum takes the neighbouring lanes (shuffles) and, at a tile's left edge, the last
// column of the previous tile (parked in LDS while the workgroup walks its band).  One patch copy (the next tile
// waits in registers), so two workgroups still fit a CU.
template <bool POOL, bool U8>
__global__ void __launch_bounds__(256, 2) conv2d_hs_stem_kernel(const Conv2dArgs a) {
  constexpr int NT = 256;
  constexpr int PH = POOL ? 25 : 21;                       // patch rows (9 / 8 stem rows)
  constexpr int PBUF = POOL ? 1 : 2;
  constexpr int CP = PH * kStemPP;                         // halves per channel plane
  constexpr int PLANEH = 3 * CP;                           // halves per split plane
  constexpr int ITEMS = 3 * PH * 35;                       // (channel, row, column pair) cells per tile
  constexpr int PIT = (ITEMS + NT - 1) / NT;
  constexpr int NROW = POOL ? 3 : 1;                       // passes per tile: stem rows folded into one pooled row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  u32x4* wl = reinterpret_cast<u32x4*>(smem_raw);                         // [step][plane][k-half][64]
  uint32_t* patch = reinterpret_cast<uint32_t*>(wl + kStemSteps * 256);   // PBUF x [plane][channel][row][pitch/2] dwords
  float* ss = reinterpret_cast<float*>(patch + PBUF * PLANEH);            // a buffer = 2 planes x PLANEH/2 dwords
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, khalf = lane >> 5;
  int bid = blockIdx.x, seg = 0;
  if (a.stem_nseg > 1) { seg = bid % a.stem_nseg; bid /= a.stem_nseg; }
  const int n = bid / a.tiles_y, ty = bid - n * a.tiles_y;
  // this workgroup's tiles [tx_own, tx_end); with the pooling fused in, a segment that does not start at the left edge
  // first computes the tile before it without storing anything: the pool's left neighbour column comes from there
  const int tx_own = a.stem_nseg > 1 ? seg * a.stem_seg_tiles : 0;
  const int tx_end = a.stem_nseg > 1 ? min(a.tiles_x, tx_own + a.stem_seg_tiles) : a.tiles_x;
  const int tx_begin = POOL && tx_own > 0 ? tx_own - 1 : tx_own;
  const int oy0 = POOL ? ty * 8 - 1 : ty * 8, iy0 = oy0 * 2 - 3;          // first stem row of the band
  const size_t hw = (size_t)a.H * a.W;
  const float* xin = a.x + (size_t)n * 3 * hw;
  constexpr uint32_t kOutside = 0xC0000000u;
  const __amdgpu_buffer_rsrc_t xrsrc =
      U8 ? __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(a.x_u8 + (size_t)n * 3 * hw), 0, (int)(3 * hw), 0x00020000)
         : __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xin), 0, (int)(3 * hw * sizeof(float)), 0x00020000);

  {
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w);
#pragma unroll
    for (int k = 0; k < kStemSteps; ++k) wl[tid + 256 * k] = wsrc[tid + 256 * k];
  }
  if (tid < 2 * kHsCout) {
    const int c = tid & (kHsCout - 1);
    ss[tid] = a.scale == nullptr ? (tid < kHsCout ? 1.f : 0.f) : (tid < kHsCout ? a.scale[c] : a.shift[c]);
  }
  // POOL: vertical maxima of the previous tile's last column (left neighbour of lane 0): [wave][32 registers][k-half]
  float* cbuf = ss + 2 * kHsCout + wave * 64 + khalf;
  float* cpark = l31 == 31 ? cbuf : ss + 2 * kHsCout + 4 * 64 + (lane & 1);   // 64 dummy words behind the four waves' columns
  if (POOL && l31 == 0) {
#pragma unroll
    for (int i = 0; i < 32; ++i) cbuf[2 * i] = -INFINITY;
  }

  // cell k of this thread: (channel, patch row, column pair); its geometry is recomputed where it is used (a few
  // integer ops per tile) instead of living in registers beside the accumulators
  auto decode = [&](int k, int& c, int& py, int& pp) {
    int e = tid + NT * k;
    asm volatile("" : "+v"(e));    // opaque: otherwise the geometry is hoisted out of the tile loop and spilled
    c = e / (PH * 35);
    const int rem = e - c * (PH * 35);
    py = rem / 35;
    pp = rem - py * 35;
  };
  float pv[PIT][2];
  auto load_p = [&](int tx) {
    const int ix0 = tx * 64 - 3;
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      int c, py, pp;
      decode(k, c, py, pp);
      const int ix = ix0 + 2 * pp, iy = iy0 + py;
      const bool rok = tid + NT * k < ITEMS && iy >= 0 && iy < a.H;
      const bool ok0 = rok && ix >= 0 && ix < a.W, ok1 = rok && ix + 1 >= 0 && ix + 1 < a.W;
      if (U8) {
        // uint8 HWC frame: byte (iy, ix, c); outside the frame the NORMALISED tensor is zero-padded, so the value is 0
        // there, not normalise(0).  Same operation order as image_normalize_kernel (IEEE division, no contraction).
#pragma clang fp contract(off)
        const uint32_t brow = (uint32_t)(iy * a.W) * 3u + (uint32_t)c;
        const uint32_t b0 = __builtin_amdgcn_raw_buffer_load_b8(xrsrc, ok0 ? brow + (uint32_t)ix * 3u : kOutside, 0, 0);
        const uint32_t b1 = __builtin_amdgcn_raw_buffer_load_b8(xrsrc, ok1 ? brow + (uint32_t)(ix + 1) * 3u : kOutside, 0, 0);
        const float mean = c == 0 ? a.u8_mean[0] : (c == 1 ? a.u8_mean[1] : a.u8_mean[2]);
        const float stdv = c == 0 ? a.u8_std[0] : (c == 1 ? a.u8_std[1] : a.u8_std[2]);
        pv[k][0] = ok0 ? ((float)b0 / 255.0f - mean) / stdv : 0.f;
        pv[k][1] = ok1 ? ((float)b1 / 255.0f - mean) / stdv : 0.f;
      } else {
        const uint32_t grow = (uint32_t)(c * (int)hw + iy * a.W) * 4u;
        pv[k][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ok0 ? grow + (uint32_t)ix * 4u : kOutside, 0, 0));
        pv[k][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, ok1 ? grow + (uint32_t)(ix + 1) * 4u : kOutside, 0, 0));
      }
    }
  };
  auto store_p = [&](int buf) {
    uint32_t* pd = patch + buf * PLANEH;
#pragma unroll
    for (int k = 0; k < PIT; ++k) {
      if (PIT * NT == ITEMS || tid + NT * k < ITEMS) {
        f16x2 h, l;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const _Float16 hj = (_Float16)pv[k][j];
          h[j] = hj;
          l[j] = (_Float16)((pv[k][j] - (float)hj) * kLoScale);
        }
        int c, py, pp;
        decode(k, c, py, pp);
        const int dw = (c * CP + py * kStemPP) / 2 + pp;
        pd[dw] = __builtin_bit_cast(uint32_t, h);
        pd[PLANEH / 2 + dw] = __builtin_bit_cast(uint32_t, l);
      }
    }
  };

  const u32x4* wa0 = wl + khalf * 64 + l31;
  // output geometry: the stem map (a.OH x a.OW) or, pooled, its MaxPool2d(3, 2, 1) image
  const int PHo = POOL ? (a.OH - 1) / 2 + 1 : a.OH, PWo = POOL ? (a.OW - 1) / 2 + 1 : a.OW;
  const size_t img = (size_t)n * a.Cout * PHo * PWo;
  // stores through a buffer descriptor: channel offset in an SGPR, the lane's pixel in one VGPR (out of range = dropped)
  const uint32_t plane_ob = (uint32_t)(PHo * PWo) * (uint32_t)sizeof(float);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(a.y + img, 0, (int)(a.Cout * plane_ob), 0x00020000);

#ifdef ADX_HS_TRACE
  // phase totals of this workgroup (tools/stem_trace.py); its records sit behind those of the 3x3 kernel
  long long tr_c = 0, tr_e = 0, tr_s = 0, tr_t = 0;
  unsigned long long* tr = g_hs_trace + (size_t)(8192 + (blockIdx.x & 8191)) * 8;
  if (threadIdx.x == 0) tr[0] = __builtin_readcyclecounter();
#endif
  load_p(tx_begin);
  store_p(PBUF == 2 ? tx_begin & 1 : 0);
  __syncthreads();
#ifdef ADX_HS_TRACE
  if (threadIdx.x == 0) tr[1] = __builtin_readcyclecounter();
#endif
  for (int tx = tx_begin; tx < tx_end; ++tx) {
#ifdef ADX_HS_TRACE
    tr_t = (long long)__builtin_readcyclecounter();
#endif
    if (tx + 1 < tx_end) load_p(tx + 1);
    const int ox = tx * kTileW + l31;
    float vm[2][16];                 // POOL: running vertical maximum of this wave's pooled row
#pragma unroll 1
    for (int pass = 0; pass < NROW; ++pass) {
      // stem rows of this pass: POOL: row 2 (4 ty + w) - 1 + pass (one row); else rows 8 ty + 2 w, + 1
      const int prow = POOL ? (2 * wave + pass) * 2 : wave * 2 * 2;     // first patch row
      const uint32_t* pb0 = patch + (PBUF == 2 ? (tx & 1) * PLANEH : 0) + prow * (kStemPP / 2) + l31;
      constexpr int NR = POOL ? 1 : 2;
      f32x16 accm[NR][2], accl[NR][2];
#pragma unroll
      for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int i = 0; i < 16; ++i) { accm[r][m][i] = 0.f; accl[r][m][i] = 0.f; }
      struct Frags { f16x8 A[2][2], B[2][NR]; };
      auto fetch = [&](Frags& f, int step) {
        // this lane's (channel, kernel row): combo 2*step + khalf; the 22nd combo has zero weights, re-reads the 21st
        const int c0 = 2 * step, c1 = 2 * step + 1 > 20 ? 20 : 2 * step + 1;
        const int off0 = ((c0 / 7) * CP + (c0 % 7) * kStemPP) / 2, off1 = ((c1 / 7) * CP + (c1 % 7) * kStemPP) / 2;
        const uint32_t* pb = pb0 + (khalf ? off1 : off0);
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
          for (int m = 0; m < 2; ++m) f.A[pl][m] = __builtin_bit_cast(f16x8, wa0[(step * 2 + pl) * 128 + m * 32]);
#pragma unroll
          for (int r = 0; r < NR; ++r) {
            const uint32_t* q = pb + pl * (PLANEH / 2) + r * 2 * (kStemPP / 2);
            u32x4 v;
            v[0] = q[0]; v[1] = q[1]; v[2] = q[2]; v[3] = q[3];
            f.B[pl][r] = __builtin_bit_cast(f16x8, v);
          }
        }
      };
      auto mfmas = [&](const Frags& f) {
#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            accm[r][m] = hs_mfma(f.A[0][m], f.B[0][r], accm[r][m]);
            accl[r][m] = hs_mfma(f.A[0][m], f.B[1][r], accl[r][m]);
            accl[r][m] = hs_mfma(f.A[1][m], f.B[0][r], accl[r][m]);
          }
      };
      if (POOL) {
        // one stem row per pass = 6 MFMAs per k-step: the fragments of step s+1 are read under the MFMAs of step s
        // (with the 128 accumulators of the two-row variant the second fragment set does not fit: it spills)
        Frags fr[2];
        fetch(fr[0], 0);
#pragma unroll
        for (int step = 0; step < kStemSteps; ++step) {
          if (step + 1 < kStemSteps) fetch(fr[(step + 1) & 1], step + 1);
          __builtin_amdgcn_sched_barrier(0);
          mfmas(fr[step & 1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll 1
        for (int step = 0; step < kStemSteps; ++step) {
          Frags f;
          fetch(f, step);
          mfmas(f);
        }
      }
      if (!POOL) {
#pragma unroll
        for (int rr = 0; rr < NR; ++rr) {
          const int oy = oy0 + wave * 2 + rr;
          const uint32_t voff = (oy < a.OH && ox < a.OW) ? (uint32_t)(oy * a.OW + ox) * 4u + (uint32_t)(4 * khalf) * plane_ob : kOutside;
#pragma unroll
          for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int cu = half * 32 + (r & 3) + 8 * (r >> 2);
              const int cl = cu + 4 * khalf;
              float v = accm[rr][half][r] + accl[rr][half][r] * (1.f / kLoScale);
              v = v * ss[cl] + ss[kHsCout + cl];
              if (a.relu) v = v > 0.f ? v : 0.f;
              __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, v), yrsrc, voff, cu * plane_ob, 0);
            }
        }
      } else {
        // BN + ReLU; rows / columns outside the stem map are the pool's padding
        const int oy = oy0 + 2 * wave + pass;
        const bool in = oy >= 0 && oy < a.OH && ox < a.OW;
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cl = half * 32 + (r & 3) + 8 * (r >> 2) + 4 * khalf;
            float t = (accm[0][half][r] + accl[0][half][r] * (1.f / kLoScale)) * ss[cl] + ss[kHsCout + cl];
            if (a.relu) t = t > 0.f ? t : 0.f;
            t = in ? t : -INFINITY;
            vm[half][r] = pass == 0 ? t : pool_max3(vm[half][r], t, -INFINITY);
          }
      }
    }
#ifdef ADX_HS_TRACE
    { const long long t = (long long)__builtin_readcyclecounter(); tr_c += t - tr_t; tr_t = t; }
#endif
    if (POOL) {
      // horizontal: pooled column 16 tx + i sits on lane 2 i; left neighbour of lane 0 = previous tile's lane 31
      const int pr = ty * 4 + wave, pq = tx * 16 + (l31 >> 1);
      const bool st = pr < PHo && (l31 & 1) == 0 && pq < PWo && tx >= tx_own;
      const uint32_t voff = st ? (uint32_t)(pr * PWo + pq) * 4u + (uint32_t)(4 * khalf) * plane_ob : kOutside;
      float mm[2][16];          // a.y_cells: the pooled values, stored as cells below
      // the previous tile's last column (written by lane 31 one tile ago): all 32 values in flight at once, read by EVERY lane
      // (a broadcast; only lane 0 of a 32-lane group uses them).  As `l31 == 0 ? cbuf[..] : up` each read sat in a branch of its
      // own with a full LDS round trip behind it, 32 times per tile (round 6: 550 -> 516 us per launch at B = 64; stepping the
      // staging geometry from cell to cell instead of dividing per cell, tried with it, costs +24 %: profiles/README.md)
      float lf[2][16];
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) lf[half][r] = cbuf[2 * (half * 16 + r)];
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          // neighbours by DPP wavefront shifts (one VALU op each; the 32-lane groups' ends are overridden below)
          const int vi = __builtin_bit_cast(int, vm[half][r]);
          const float up = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x138 /* wave_shr:1 */, 0xf, 0xf, false));
          const float dn = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, vi, 0x130 /* wave_shl:1 */, 0xf, 0xf, false));
          const float left = l31 == 0 ? lf[half][r] : up;
          const float right = l31 == 31 ? -INFINITY : dn;
          const float m = pool_max3(left, vm[half][r], right);
          cpark[2 * (half * 16 + r)] = vm[half][r];      // lane 31 parks its column for the next tile, the others hit a dummy row
          mm[half][r] = m;
        }
      if (!a.y_cells) {         // (one uniform branch around all 32 stores, not one per value)
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int cu = half * 32 + (r & 3) + 8 * (r >> 2);       // + 4 * khalf, which rides in voff
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, mm[half][r]), yrsrc, voff, cu * plane_ob, 0);
          }
      } else {          // the pooled map as a cell tensor (conv2d_hs3x3_kernel: XCELLS): layer1's first conv copies cells
        const uint32_t cplane = (uint32_t)(PHo * PWo) * 16u;
        const uint32_t vcell = st ? (uint32_t)(pr * PWo + pq) * 16u + (uint32_t)khalf * 2u * cplane : kOutside;
#pragma unroll
        for (int half = 0; half < 2; ++half)
          cells_store32<false>(mm[half], nullptr, nullptr, 0, nullptr, false, yrsrc, vcell, (uint32_t)(half * 4) * 2u * cplane, cplane);
      }
    }
#ifdef ADX_HS_TRACE
    { const long long t = (long long)__builtin_readcyclecounter(); tr_e += t - tr_t; tr_t = t; }
#endif
    if (tx + 1 < tx_end) {
      if (PBUF == 1) __syncthreads();        // every wave is done with the only patch copy
      store_p(PBUF == 2 ? (tx + 1) & 1 : 0);
    }
    __syncthreads();
#ifdef ADX_HS_TRACE
    { const long long t = (long long)__builtin_readcyclecounter(); tr_s += t - tr_t; tr_t = t; }
#endif
  }
#ifdef ADX_HS_TRACE
  if (threadIdx.x == 0) {
    tr[2] = (unsigned long long)tr_c;
    tr[3] = (unsigned long long)tr_e;
    tr[4] = (unsigned long long)tr_s;
    tr[5] = __builtin_readcyclecounter();
  }
#endif
}

// [64][3][7][7] fp32 -> [step][plane][k-half][64][8] fp16 with k = (combo = 2*step + k-half -> channel combo/7,
// kernel row combo%7; element j = kernel column, the 8th and the 22nd combo are zero)
__global__ void conv2d_hs_stem_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ p) {
  const int idx = blockIdx.x * 256 + threadIdx.x;      // over [step][k-half][64][8]
  if (idx >= kStemSteps * 2 * 64 * 8) return;
  const int j = idx & 7, ml = (idx >> 3) & 63, h = (idx >> 9) & 1, step = idx >> 10;
  const int combo = 2 * step + h;
  float v = 0.f;
  if (combo < 21 && j < 7) v = w[((size_t)ml * 3 + combo / 7) * 49 + (combo % 7) * 7 + j];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * kLoScale);
  const size_t cell = ((size_t)(step * 2 + 0) * 2 + h) * 64 + ml;
  p[cell * 8 + j] = hi;
  p[(cell + 128) * 8 + j] = lo;
}

static bool hs_is_stem(const ConvSpec& L) { return L.k == 7 && L.stride == 2 && L.pad == 3 && L.cin == 3 && L.cout == 64; }

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

// The same for up to kHsPackJobs weight tensors in ONE launch (the training step re-lays every conv weight every step: 36
// forward images and 29 data-gradient images, each a ~5 us launch of its own otherwise).  The job table travels in the kernel
// arguments; a block finds its job by a binary search of the (wave-uniform) first-block column.
constexpr int kHsPackJobs = 48;
struct HsPackTable {
  const float* w[kHsPackJobs];
  _Float16* p[kHsPackJobs];
  int M[kHsPackJobs], Kc[kHsPackJobs], Kreal[kHsPackJobs], taps[kHsPackJobs], dgrad[kHsPackJobs];
  unsigned first[kHsPackJobs + 1];     // first block of job j; first[n] = the grid
  int n;
};
__global__ void __launch_bounds__(256) conv2d_hs_pack_many_kernel(const HsPackTable t) {
  int j = 0, jend = t.n;               // binary search of the job: six kernel-argument reads, not one per job
  while (jend - j > 1) {
    const int mid = (j + jend) >> 1;
    if (blockIdx.x >= t.first[mid]) j = mid; else jend = mid;
  }
  const int M = t.M[j], Kc = t.Kc[j], Kreal = t.Kreal[j], taps = t.taps[j], dgrad = t.dgrad[j];
  const float* __restrict__ w = t.w[j];
  _Float16* __restrict__ p = t.p[j];
  const size_t total = (size_t)M * Kc * taps;
  const size_t idx = (size_t)(blockIdx.x - t.first[j]) * 256 + threadIdx.x;   // over [M/64][Kc/16][tap][k-half][64][8]
  if (idx >= total) return;
  const int e = idx & 7;
  const int ml = (idx >> 3) & 63;
  const int h = (idx >> 9) & 1;
  size_t rest = idx >> 10;
  const int tap = rest % taps; rest /= taps;
  const int nchunks = Kc / kHsCC;
  const int chunk = rest % nchunks;
  const int ct = rest / nchunks;
  const int m = ct * 64 + ml, kc = chunk * kHsCC + h * 8 + e;
  float v = 0.f;
  if (kc < Kreal) v = dgrad ? w[((size_t)kc * M + m) * taps + (taps - 1 - tap)] : w[((size_t)m * Kreal + kc) * taps + tap];
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * kLoScale);
  const size_t cell = ((((size_t)(ct * nchunks + chunk) * taps + tap) * 2 + 0) * 2 + h) * 64 + ml;
  p[cell * 8 + e] = hi;
  p[(cell + 128) * 8 + e] = lo;
}

int conv2d_hs_pack_many(const HsPackJob* jobs, int n, hipStream_t s) {
  for (int j0 = 0; j0 < n; j0 += kHsPackJobs) {
    HsPackTable t;
    t.n = std::min(kHsPackJobs, n - j0);
    unsigned blocks = 0;
    for (int j = 0; j < t.n; ++j) {
      const HsPackJob& q = jobs[j0 + j];
      ADX_REQUIRE(q.w != nullptr && q.packed != nullptr && q.cin_pad % kHsCC == 0 && q.cout % kHsCout == 0,
                  "conv2d_hs_pack_many: job %d is not a split-fp16 weight image", j0 + j);
      t.w[j] = q.w; t.p[j] = (_Float16*)q.packed;
      t.M[j] = q.cout; t.Kc[j] = q.cin_pad; t.Kreal[j] = q.cin; t.taps[j] = q.taps; t.dgrad[j] = q.dgrad;
      t.first[j] = blocks;
      blocks += (unsigned)(((size_t)q.cout * q.cin_pad * q.taps + 255) / 256);
    }
    t.first[t.n] = blocks;
    conv2d_hs_pack_many_kernel<<<dim3(blocks), dim3(256), 0, s>>>(t);
    ADX_LAUNCH_CHECK();
  }
  return ADX_OK;
}

bool conv2d_hs_eligible(const ConvSpec& L) {
  if (debug_switches().conv_exact) return false;     // ADX_CONV_EXACT=1: every conv on the exact-fp32 MFMA kernels
  if (hs_is_stem(L) && !L.dgrad) return true;
  if (L.cin % kHsCC != 0 || L.cin_pad != L.cin || L.cout % kHsCout != 0) return false;
  return L.k == 3 && (L.stride == 1 || L.stride == 2);
}

size_t conv2d_packed_floats(const ConvSpec& L) {
  const size_t direct = (size_t)L.k * L.k * L.cin_pad * L.cout;
  return direct;
}

bool conv2d_hs_pack_batchable(const ConvSpec& c, int dgrad) { return conv2d_hs_eligible(c) && !(hs_is_stem(c) && !dgrad); }

int conv2d_hs_pack(const ConvSpec& c, const float* w, void* packed, int dgrad, hipStream_t s) {
  if (hs_is_stem(c) && !dgrad) {
    conv2d_hs_stem_pack_kernel<<<dim3(ceil_div(kStemSteps * 2 * 64 * 8, 256)), dim3(256), 0, s>>>(w, (_Float16*)packed);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  const size_t total = (size_t)c.cout * c.cin_pad * c.k * c.k;
  conv2d_hs_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
      w, (_Float16*)packed, c.cout, c.cin_pad, c.cin, c.k * c.k, dgrad, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

template <int STRIDE, int K, int ROWS, int PBUF, bool DS, bool XCELLS = false, bool YCELLS = false>
static int hs_launch_t(Conv2dArgs a, hipStream_t s) {
  constexpr int TH = 4 * ROWS;
  constexpr int PH = (TH - 1) * STRIDE + K, PW = (kTileW - 1) * STRIDE + K;
  constexpr size_t lds = (size_t)PBUF * 64 * PH * PW + (size_t)2 * K * 256 * 16 + (DS ? 256 * 16 : 0) +
                         (DS ? 4 : 2) * kHsCout * sizeof(float) + 16;
  static_assert(lds <= 80 * 1024, "two workgroups per CU need <= 80 KB each");
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs_kernel<STRIDE, K, ROWS, PBUF, DS, XCELLS, YCELLS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once.commit();
  }
  a.tiles_x = ceil_div(a.OW, kTileW); a.tiles_y = ceil_div(a.OH, TH); a.cout_tiles = a.Cout / kHsCout;
  const size_t grid = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y * a.N;
  ADX_REQUIRE(grid < (1u << 31), "conv2d_hs: grid too large");
  ADX_REQUIRE((size_t)a.Cout * a.OH * a.OW * sizeof(float) < 0x7FFFFFFFu, "conv2d_hs: one image of the output exceeds the 32-bit byte offsets");
  conv2d_hs_kernel<STRIDE, K, ROWS, PBUF, DS, XCELLS, YCELLS><<<dim3((unsigned)grid), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

template <bool POOL, bool U8 = false>
static int hs_stem_launch(Conv2dArgs a, hipStream_t s) {
  constexpr int PH = POOL ? 25 : 21, PBUF = POOL ? 1 : 2;
  constexpr size_t lds = (size_t)kStemSteps * 4096 + (size_t)PBUF * 4 * 3 * PH * kStemPP + 2 * kHsCout * sizeof(float) +
                         (POOL ? (4 * 64 + 2) * sizeof(float) : 0);     // parked columns + the dummy words of the other lanes
  static_assert(lds <= 80 * 1024, "two workgroups per CU need <= 80 KB each");
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs_stem_kernel<POOL, U8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once.commit();
  }
  a.tiles_x = ceil_div(a.OW, kTileW); a.cout_tiles = 1;
  a.tiles_y = POOL ? ceil_div((a.OH - 1) / 2 + 1, 4) : ceil_div(a.OH, 8);
  // few images: split the bands' tile walks so that the grid covers the chip (one 256x900 frame: 16 bands of 15 tiles)
  const int bands = a.tiles_y * a.N;
  a.stem_seg_tiles = 0; a.stem_nseg = 1;
  if (bands < 128 && a.tiles_x > 1) {
    const int want = std::min(a.tiles_x, std::max(1, 256 / bands));
    a.stem_seg_tiles = ceil_div(a.tiles_x, want);
    a.stem_nseg = ceil_div(a.tiles_x, a.stem_seg_tiles);
    if (a.stem_nseg <= 1) { a.stem_seg_tiles = 0; a.stem_nseg = 1; }
  }
  conv2d_hs_stem_kernel<POOL, U8><<<dim3((unsigned)(bands * a.stem_nseg)), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int conv2d_hs_stem_pool(const ConvSpec& L, const float* x, const float* w, const float* scale, const float* shift,
                        float* pooled, int N, int H, int W, hipStream_t s, const uint8_t* frames_u8, const float* mean,
                        const float* stdv, int y_cells) {
  ADX_REQUIRE((x || frames_u8) && w && scale && shift && pooled, "conv2d_hs_stem_pool: null pointer");
  ADX_REQUIRE((size_t)3 * H * W * sizeof(float) < 0xC0000000u, "conv2d_hs_stem_pool: image too large for 32-bit offsets");
  Conv2dArgs a{};
  a.x = x; a.w = w; a.scale = scale; a.shift = shift; a.y = pooled;
  if (frames_u8 != nullptr) {
    ADX_REQUIRE(mean && stdv, "conv2d_hs_stem_pool: uint8 frames need mean / std");
    a.x_u8 = frames_u8;
    for (int c = 0; c < 3; ++c) { a.u8_mean[c] = mean[c]; a.u8_std[c] = stdv[c]; }
  }
  a.N = N; a.Cin = 3; a.H = H; a.W = W; a.Cout = 64;
  a.OH = conv_out_dim(H, 7, 2, 3); a.OW = conv_out_dim(W, 7, 2, 3);
  a.KH = 7; a.KW = 7; a.stride = 2; a.pad = 3; a.relu = 1;
  a.cin_pad = L.cin_pad; a.cc = L.cc;
  a.y_cells = y_cells;
  return frames_u8 != nullptr ? hs_stem_launch<true, true>(a, s) : hs_stem_launch<true>(a, s);
}

int conv2d_hs_launch_block_s2(const ConvSpec& c1, const ConvSpec& ds, const float* x, const float* w1, const float* scale1,
                              const float* shift1, float* y1, const float* wd, const float* scaled, const float* shiftd,
                              float* yd, int N, int H, int W, hipStream_t s, int x_cells, int y_cells, int relu) {
  ADX_REQUIRE(x && w1 && y1 && wd && yd && (scale1 != nullptr) == (shift1 != nullptr) && (scaled != nullptr) == (shiftd != nullptr),
              "conv2d_hs block launch: null pointer");
  Conv2dArgs a{};
  a.x = x; a.w = w1; a.scale = scale1; a.shift = shift1; a.res = nullptr; a.y = y1; a.x_amax = nullptr; a.x_amax_n = 0;
  a.w_ds = wd; a.scale_ds = scaled; a.shift_ds = shiftd; a.y_ds = yd;
  a.N = N; a.Cin = c1.cin; a.H = H; a.W = W; a.Cout = c1.cout;
  a.OH = conv_out_dim(H, 3, 2, 1); a.OW = conv_out_dim(W, 3, 2, 1);
  a.KH = 3; a.KW = 3; a.stride = 2; a.pad = 1; a.relu = relu;
  a.cin_pad = c1.cin_pad; a.cc = c1.cc;
  a.x_cells = x_cells;
  a.y_cells = y_cells;
  (void)ds;
  return conv2d_hs_launch(c1, a, s);
}

// second half of a split 3x3 conv: y = [relu](sum_p part[p] * scale[c] + shift[c] [+ res]), four elements per thread
// (parts are added in index order: deterministic)
__global__ void __launch_bounds__(256) conv2d_split_reduce_kernel(const float* __restrict__ part, size_t part_stride, int nparts,
                                                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                                                  const float* __restrict__ res, float* __restrict__ y,
                                                                  int cout, int plane4, size_t total4, int relu) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const int c = (int)((i / plane4) % cout);
  f32x4 v = reinterpret_cast<const f32x4*>(part)[i];
  for (int p0 = 1; p0 < nparts; p0 += 8) {        // eight loads in flight, added in index order
    f32x4 t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = reinterpret_cast<const f32x4*>(part + (size_t)(p0 + j < nparts ? p0 + j : 0) * part_stride)[i];
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (p0 + j < nparts) v += t[j];
  }
  const float sc = scale != nullptr ? scale[c] : 1.f, sh = shift != nullptr ? shift[c] : 0.f;
  v = v * sc + sh;
  if (res != nullptr) v += reinterpret_cast<const f32x4*>(res)[i];
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaxf(v[k], 0.f);
  }
  reinterpret_cast<f32x4*>(y)[i] = v;
}

// fp32-layout launches of a whole batch tile the virtual row too (conv2d_hs3x3_kernel: VR) -- training forward / data gradient
static bool hs_vrow_ok(const Conv2dArgs& a) {
  return debug_switches().conv_vrow && a.N > 1 && (long)a.N * (a.OW + 1) < (1L << 21) &&
         (size_t)a.N * a.Cout * a.OH * a.OW * sizeof(float) < 0xC0000000u && (size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u;
}
static int hs_vrow_tiles_x(const Conv2dArgs& a) { return ceil_div(a.N * (a.OW + 1) - 1, kTileW); }

template <int MODE>
static int hs3x3_launch(Conv2dArgs a, hipStream_t s) {
  constexpr int NT = MODE == 0 ? 256 : 512, TH = MODE == 1 ? 16 : 8, CT = MODE == 2 ? 2 : 1;
  constexpr size_t lds = (size_t)2 * 64 * (TH + 2) * 34 + (size_t)2 * 3 * 256 * CT * 16 + (2 * 64 * CT + 8) * sizeof(float) + 48;
  constexpr size_t lds_bs = lds + 4 * 64 * CT * sizeof(float);       // STATS == 2: + the consumer BatchNorm's constants
  static_assert(lds_bs <= (MODE == 0 ? 80 : 160) * 1024, "LDS budget");
  static std::atomic<uint64_t> attr{0};
  if (DeviceOnce once{attr}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 1>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 2>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bs));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 1, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 2, false, false, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bs));
    once.commit();
  }
  a.tiles_x = ceil_div(a.OW, kTileW); a.tiles_y = ceil_div(a.OH, TH); a.cout_tiles = a.Cout / (kHsCout * CT);
  const size_t grid = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y * a.N;
  ADX_REQUIRE(grid < (1u << 31), "conv2d_hs: grid too large");
  ADX_REQUIRE((size_t)a.Cout * a.OH * a.OW * sizeof(float) < 0x7FFFFFFFu, "conv2d_hs: one image of the output exceeds the 32-bit byte offsets");
  // Small batches (one camera frame per tick): a 512->512 layer on one 8x29 map is 8 workgroups, each walking 96 stages
  // (83 us); with a scratch buffer the chunks are split over up to 16 workgroups per tile and a reduce launch finishes.
  constexpr bool split_on = true;
  const size_t out_floats = (size_t)a.N * a.Cout * a.OH * a.OW;
  const size_t cap = a.part != nullptr ? a.part_stride : 0;      // conv2d_launch_raw parks the scratch capacity here
  const int nchunks = a.cin_pad / kHsCC;
  a.ksplit = 1; a.cper = nchunks; a.part_stride = 0;
  const bool fp32_layout = !(a.x_cells || a.y_cells || a.res_cells);
  // (a training forward may read cells, and so may a data gradient whose cells carry their scale)
  const bool xscaled = a.x_amax != nullptr && a.x_amax_n < 0;
  const bool vrow = (fp32_layout || ((a.stats_part != nullptr || xscaled) && !a.y_cells && !a.res_cells)) && hs_vrow_ok(a);
  auto set_vrow = [&]() -> size_t {           // column tiles over the images side by side (one shared zero column between neighbours)
    a.vw = a.OW + 1;
    a.inv_vw = 1.f / (float)a.vw;
    a.tiles_x = hs_vrow_tiles_x(a);
    return (size_t)a.cout_tiles * a.tiles_x * a.tiles_y;
  };
  if (a.stats_part != nullptr && a.x_cells && (conv2d_hs3x3q_train_eligible(a) || conv2d_hs3x3q_dgrad_eligible(a))) {      // the 16x16x32 kernel where its tile rules hold
    a.part = nullptr;
    return conv2d_hs3x3q_launch(a, s);
  }
  if (a.stats_part != nullptr) {          // training forward: statistics in the epilogue (one workgroup per tile: no split)
    const size_t sgrid = vrow ? set_vrow() : grid;
    const int slots = vrow ? a.tiles_y * a.tiles_x : a.N * a.tiles_y * a.tiles_x;
    ADX_REQUIRE(a.stats_p == slots, "conv2d_hs: statistics buffer laid out for %d tiles, launch has %d", a.stats_p, slots);
    a.part = nullptr;
    if (a.x_cells) {
      // training forward on a cell-layout input (resnet_train.hip: the activation between a block's two convs): the staging copies
      // cells instead of converting fp32 values, everything else -- fp32 conv output, statistics -- as below
      // ... or a data gradient with the consumer BatchNorm's sums in its epilogue, reading a gradient that was written as cells
      ADX_REQUIRE((a.x_amax == nullptr || xscaled) && (a.bs_raw == nullptr) == (a.x_amax == nullptr) && !a.y_cells && !a.res_cells,
                  "conv2d_hs: a cell-layout input with statistics belongs to a training-forward launch or to a data gradient with its scale");
      ADX_REQUIRE((size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u, "conv2d_hs: a cell-layout tensor exceeds the 32-bit byte offsets");
      static std::atomic<uint64_t> xattr{0};
      if (DeviceOnce once{xattr}; once) {
        ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 1, true, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 1, true, false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 2, true, false, true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bs));
        ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 2, true, false, false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bs));
        once.commit();
      }
      if (a.bs_raw != nullptr) {
        if (vrow) conv2d_hs3x3_kernel<MODE, 2, true, false, true><<<dim3((unsigned)sgrid), dim3(NT), lds_bs, s>>>(a);
        else conv2d_hs3x3_kernel<MODE, 2, true, false, false><<<dim3((unsigned)sgrid), dim3(NT), lds_bs, s>>>(a);
      } else if (vrow) conv2d_hs3x3_kernel<MODE, 1, true, false, true><<<dim3((unsigned)sgrid), dim3(NT), lds, s>>>(a);
      else conv2d_hs3x3_kernel<MODE, 1, true, false, false><<<dim3((unsigned)sgrid), dim3(NT), lds, s>>>(a);
    } else if (vrow) {
      if (a.bs_raw != nullptr) conv2d_hs3x3_kernel<MODE, 2, false, false, true><<<dim3((unsigned)sgrid), dim3(NT), lds_bs, s>>>(a);
      else conv2d_hs3x3_kernel<MODE, 1, false, false, true><<<dim3((unsigned)sgrid), dim3(NT), lds, s>>>(a);
    } else {
      if (a.bs_raw != nullptr) conv2d_hs3x3_kernel<MODE, 2><<<dim3((unsigned)sgrid), dim3(NT), lds_bs, s>>>(a);
      else conv2d_hs3x3_kernel<MODE, 1><<<dim3((unsigned)sgrid), dim3(NT), lds, s>>>(a);
    }
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  if (MODE == 0 && split_on && cap > 0 && grid <= 64 && nchunks >= 8 && a.x_amax == nullptr && (a.OH * a.OW) % 4 == 0 &&
      (reinterpret_cast<uintptr_t>(a.y) & 15) == 0 && (a.res == nullptr || (reinterpret_cast<uintptr_t>(a.res) & 15) == 0)) {
    int S = 16;
    while (S > 1 && (nchunks % S != 0 || (nchunks / S) % 2 != 0 || grid * S > 256 || out_floats * S > cap)) S >>= 1;
    if (S > 1) {
      Conv2dArgs c = a;
      c.ksplit = S; c.cper = nchunks / S; c.part_stride = out_floats;
      c.scale = nullptr; c.shift = nullptr; c.res = nullptr; c.relu = 0;
      conv2d_hs3x3_kernel<MODE><<<dim3((unsigned)(grid * S)), dim3(NT), lds, s>>>(c);
      ADX_LAUNCH_CHECK();
      const size_t total4 = out_floats / 4;
      conv2d_split_reduce_kernel<<<dim3((unsigned)ceil_div((long)total4, 256L)), dim3(256), 0, s>>>(
          a.part, out_floats, S, a.scale, a.shift, a.res, a.y, a.Cout, a.OH * a.OW / 4, total4, a.relu);
      ADX_LAUNCH_CHECK();
      return ADX_OK;
    }
  }
  a.part = nullptr;
  if (a.x_cells && xscaled && !a.y_cells && !a.res_cells) {
    // a data gradient (no statistics wanted from it) on a gradient that was written as cells: fp32 output [+ fp32 residual]
    static std::atomic<uint64_t> dattr{0};
    if (DeviceOnce once{dattr}; once) {
      ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0, true, false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0, true, false, false>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      once.commit();
    }
    ADX_REQUIRE((size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u, "conv2d_hs: a cell-layout tensor exceeds the 32-bit byte offsets");
    if (vrow) {
      const size_t vgrid = set_vrow();
      conv2d_hs3x3_kernel<MODE, 0, true, false, true><<<dim3((unsigned)vgrid), dim3(NT), lds, s>>>(a);
    } else {
      conv2d_hs3x3_kernel<MODE, 0, true, false, false><<<dim3((unsigned)grid), dim3(NT), lds, s>>>(a);
    }
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  if (a.x_cells || a.y_cells || a.res_cells) {
    // the executor keeps a layer's 3x3 convs in the cell layout from the first one's output to the last one's (the stride-2
    // kernel and the average pool read cells too), so a cell operand always comes with a cell output
    ADX_REQUIRE(a.y_cells && a.x_amax == nullptr, "conv2d_hs: cell-layout operands come with a cell-layout output (and no dynamic range)");
    if (conv2d_hs3x3q_eligible(a)) return conv2d_hs3x3q_launch(a, s);      // the 16x16x32 kernel (conv2d_hs16.hip) where its tile rules hold
    constexpr size_t clds = lds;
    static std::atomic<uint64_t> cattr{0};
    if (DeviceOnce once{cattr}; once) {
      const void* fns[2] = {reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0, true, true>),
                            reinterpret_cast<const void*>(&conv2d_hs3x3_kernel<MODE, 0, false, true>)};
      for (const void* f : fns) ADX_CHECK_HIP(hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clds));
      once.commit();
    }
    // column tiles over the images side by side with one shared zero column between neighbours (conv2d_hs3x3_kernel: YCELLS):
    // one padded MFMA column per image instead of the round-up to 32
    a.vw = a.N > 1 ? a.OW + 1 : a.OW;
    a.inv_vw = 1.f / (float)a.vw;
    ADX_REQUIRE((long)a.N * a.vw < (1L << 21), "conv2d_hs: batch x width exceeds the virtual-row arithmetic");
    a.tiles_x = ceil_div(a.N * a.vw - (a.N > 1 ? 1 : 0), kTileW);       // the last image's zero column needs no tile
    const size_t cgrid = (size_t)a.cout_tiles * a.tiles_x * a.tiles_y;
    ADX_REQUIRE((size_t)a.N * a.Cout * a.OH * a.OW * sizeof(float) < 0xC0000000u && (size_t)a.N * a.Cin * a.H * a.W * sizeof(float) < 0xC0000000u,
                "conv2d_hs: a cell-layout tensor exceeds the 32-bit byte offsets");
    if (a.x_cells) conv2d_hs3x3_kernel<MODE, 0, true, true><<<dim3((unsigned)cgrid), dim3(NT), clds, s>>>(a);
    else conv2d_hs3x3_kernel<MODE, 0, false, true><<<dim3((unsigned)cgrid), dim3(NT), clds, s>>>(a);
    ADX_LAUNCH_CHECK();
    return ADX_OK;
  }
  if (vrow) {
    const size_t vgrid = set_vrow();
    conv2d_hs3x3_kernel<MODE, 0, false, false, true><<<dim3((unsigned)vgrid), dim3(NT), lds, s>>>(a);
  } else {
    conv2d_hs3x3_kernel<MODE><<<dim3((unsigned)grid), dim3(NT), lds, s>>>(a);
  }
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// tile mode of the pipelined 3x3 stride-1 kernel for this launch, -1 when another kernel serves it
static int hs3x3_mode(const ConvSpec& L, const Conv2dArgs& a) {
  if (!(L.k == 3 && L.stride == 1 && a.w_ds == nullptr)) return -1;
  if (!((L.cin_pad / kHsCC) % 2 == 0 && L.pad == 1)) return -1;
  const int mode_env = debug_switches().hs_mode;       // ADX_HS_MODE=0|1|2 pins the tile mode (default: by shape)
  // The three tiles time within 3 % of each other on every ResNet-34 shape (the MFMA rate the chip sustains on
  // random data paces all of them): short K loops take the small tile (better tail balance), long ones the
  // 8-wave tiles that move fewer operand bytes per MFMA.
  int mode = a.Cin < 256 ? 0 : (a.OH > 8 ? 1 : (a.Cout % 128 == 0 ? 2 : 0));
  // small batches (the deployed case: one camera frame per tick): the 8-wave tiles leave most of the chip idle (512->512
  // @8x29 at B = 1: four workgroups); the 4-wave tile doubles the workgroup count
  if ((long)a.N * ceil_div(a.OH, 8) * ceil_div(a.OW, kTileW) * (a.Cout / kHsCout) <= 256) mode = 0;
  if (mode_env >= 0 && !(mode_env == 2 && a.Cout % 128 != 0)) mode = mode_env;
  return mode;
}

bool conv2d_hs3x3_plain(const ConvSpec& L, int N, int H, int W) {
  if (!debug_switches().conv_cells || !conv2d_hs_eligible(L) || L.dgrad) return false;   // ADX_CONV_CELLS=0: fp32 NCHW between all layers
  Conv2dArgs a{};
  a.N = N; a.Cin = L.cin; a.Cout = L.cout; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, L.k, L.stride, L.pad); a.OW = conv_out_dim(W, L.k, L.stride, L.pad);
  const int mode = hs3x3_mode(L, a);
  if (mode < 0 || L.cin % 16 != 0 || L.cout % 64 != 0) return false;
  // the cell kernels address whole tensors through one descriptor with 32-bit offsets (batches of ~870 frames at 256x900 and up
  // stay on the fp32 layout with its per-image descriptors)
  if ((size_t)N * L.cin * H * W * sizeof(float) >= 0xC0000000u || (size_t)N * L.cout * a.OH * a.OW * sizeof(float) >= 0xC0000000u) return false;
  const long grid0 = (long)a.N * ceil_div(a.OH, 8) * ceil_div(a.OW, kTileW) * (a.Cout / kHsCout);
  return mode != 0 || grid0 > 64 || L.cin_pad / kHsCC < 8;       // hs3x3_launch<0> splits the reduction of smaller launches
}

bool conv2d_hs3x3_dgrad_cells(const ConvSpec& L, int N, int H, int W) {
  if (debug_switches().train_cells < 4 || !conv2d_hs_eligible(L) || !L.dgrad) return false;
  Conv2dArgs a{};
  a.N = N; a.Cin = L.cin; a.Cout = L.cout; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, L.k, L.stride, L.pad); a.OW = conv_out_dim(W, L.k, L.stride, L.pad);
  return hs3x3_mode(L, a) >= 0 && L.cin % 16 == 0 && L.cin == L.cin_pad && L.cout % 64 == 0 &&
         (size_t)N * L.cin * H * W * sizeof(float) < 0xC0000000u;
}

bool conv2d_hs3x3_dgrad_stats(const ConvSpec& L, int N, int H, int W, bool x_cells, size_t stats_floats) {
  if (!conv2d_hs_eligible(L) || !L.dgrad) return false;
  Conv2dArgs a{};
  a.N = N; a.Cin = L.cin; a.Cout = L.cout; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, L.k, L.stride, L.pad); a.OW = conv_out_dim(W, L.k, L.stride, L.pad);
  a.KH = L.k; a.KW = L.k; a.stride = L.stride; a.pad = L.pad; a.cin_pad = L.cin_pad;
  a.x_cells = x_cells ? 1 : 0;
  const int tiles = conv2d_hs_stats_tiles(L, a);
  return L.k == 3 && L.stride == 1 && tiles > 0 && (size_t)tiles * (L.cout * 2 + L.cout / 64) <= stats_floats;
}

bool conv2d_hs3x3_train_cells(const ConvSpec& L, int N, int H, int W, size_t stats_floats) {
  if (debug_switches().train_cells == 0 || !conv2d_hs_eligible(L) || L.dgrad) return false;     // ADX_TRAIN_CELLS=0: fp32 NCHW everywhere
  Conv2dArgs a{};
  a.N = N; a.Cin = L.cin; a.Cout = L.cout; a.H = H; a.W = W;
  a.OH = conv_out_dim(H, L.k, L.stride, L.pad); a.OW = conv_out_dim(W, L.k, L.stride, L.pad);
  a.x_cells = 1;
  if (hs3x3_mode(L, a) < 0 || L.cin % 16 != 0 || L.cin != L.cin_pad || L.cout % 64 != 0) return false;
  if ((size_t)N * L.cin * H * W * sizeof(float) >= 0xC0000000u) return false;
  const int tiles = conv2d_hs_stats_tiles(L, a);
  return tiles > 0 && (size_t)tiles * L.cout * 2 <= stats_floats;
}

int conv2d_hs_stats_tiles(const ConvSpec& L, const Conv2dArgs& a) {
  const int mode = hs3x3_mode(L, a);
  if (mode < 0) return 0;
  if (a.x_cells && (conv2d_hs3x3q_train_eligible(a) || conv2d_hs3x3q_dgrad_eligible(a))) return conv2d_hs3x3q_train_tiles(a);
  if (!(a.y_cells || a.res_cells) && hs_vrow_ok(a)) return ceil_div(a.OH, mode == 1 ? 16 : 8) * hs_vrow_tiles_x(a);
  return a.N * ceil_div(a.OH, mode == 1 ? 16 : 8) * ceil_div(a.OW, kTileW);
}

// forward weight [cout][cin][3][3] -> weight [4 cin][cout][2][2] of the 2x2 conv that is the stride-2 data gradient:
// class (py, px), window cell (r, c) of dy -> the forward tap (kh, kw) that links them, or none.  Row rule (columns alike):
// an even input row 2j is reached from output row j through kh = 1 only (window row 0); an odd row 2j + 1 from output row j
// through kh = 2 (window row 0) and from output row j + 1 through kh = 0 (window row 1).
__global__ void dgrad_s2_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;     // over [4][cin][cout][2][2]
  if (idx >= (size_t)16 * cin * cout) return;
  const int c = idx & 1, r = (idx >> 1) & 1;
  size_t rest = idx >> 2;
  const int co = rest % cout; rest /= cout;
  const int ci = rest % cin;
  const int cls = (int)(rest / cin), py = cls >> 1, px = cls & 1;
  const int kh = py == 0 ? (r == 0 ? 1 : -1) : (r == 0 ? 2 : 0);
  const int kw = px == 0 ? (c == 0 ? 1 : -1) : (c == 0 ? 2 : 0);
  out[idx] = (kh >= 0 && kw >= 0) ? w[(((size_t)co * cin + ci) * 3 + kh) * 3 + kw] : 0.f;
}

bool conv2d_hs_dgrad_s2_eligible(int cin, int cout) {
  ConvSpec probe{};
  probe.cin = cout; probe.cin_pad = cout; probe.cout = 4 * cin; probe.k = 3; probe.stride = 1; probe.dgrad = 1;
  return cin % kHsCout == 0 && cout % kHsCC == 0 && conv2d_hs_eligible(probe);
}

int conv2d_hs_dgrad_s2(const float* w, const float* dy, float* dx, int accumulate, int N, int cin, int cout, int H, int W,
                       float* wbuild, float* wimg, const uint32_t* dy_amax, int dy_amax_n, hipStream_t s) {
  ADX_REQUIRE(w && dy && dx && wbuild && wimg, "conv2d_hs_dgrad_s2: null pointer");
  ADX_REQUIRE(conv2d_hs_dgrad_s2_eligible(cin, cout), "conv2d_hs_dgrad_s2: cin %d / cout %d outside the kernel's rules", cin, cout);
  const size_t nw = (size_t)16 * cin * cout;
  dgrad_s2_weights_kernel<<<dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, s>>>(w, wbuild, cin, cout);
  ADX_LAUNCH_CHECK();
  ConvSpec g{};
  g.cin = cout; g.cin_pad = cout; g.cout = 4 * cin; g.k = 2; g.stride = 1; g.pad = 0; g.cc = kHsCC;
  int rc = conv2d_hs_pack(g, wbuild, wimg, 0, s);
  if (rc != ADX_OK) return rc;
  Conv2dArgs a{};
  a.x = dy; a.w = wimg; a.res = accumulate ? dx : nullptr; a.y = dx; a.x_amax = dy_amax; a.x_amax_n = dy_amax_n;
  a.N = N; a.Cin = cout; a.H = conv_out_dim(H, 3, 2, 1); a.W = conv_out_dim(W, 3, 2, 1); a.Cout = 4 * cin;
  a.OH = a.H; a.OW = a.W;                  // one 2x2 window per low-resolution pixel; the window's far row / column may lie
  a.KH = 2; a.KW = 2; a.stride = 1; a.pad = 0; a.relu = 0;     // outside dy: zeros, like any padding
  a.cin_pad = cout; a.cc = kHsCC;
  a.d2s_cin = cin; a.d2s_h = H; a.d2s_w = W;
  ADX_REQUIRE((size_t)cin * H * W * sizeof(float) < 0x7FFFFFFFu, "conv2d_hs_dgrad_s2: one image of dx exceeds the 32-bit byte offsets");
  return hs_launch_t<1, 2, 2, 2, false>(a, s);
}

int conv2d_hs_launch(const ConvSpec& L, Conv2dArgs a, hipStream_t s) {
  ADX_REQUIRE((size_t)L.cin * a.H * a.W * sizeof(float) < 0xC0000000u, "conv2d_hs: one image of the input exceeds the 32-bit byte offsets");
  const bool ds = a.w_ds != nullptr;
  if (hs_is_stem(L) && !ds) {
    ADX_REQUIRE(a.x_amax == nullptr && a.res == nullptr, "conv2d_hs stem: no residual / dynamic range");
    return hs_stem_launch<false>(a, s);
  }
  if (L.k == 3 && L.stride == 1 && !ds) {
    const int mode = hs3x3_mode(L, a);
    if (mode >= 0) return mode == 1 ? hs3x3_launch<1>(a, s) : (mode == 2 ? hs3x3_launch<2>(a, s) : hs3x3_launch<0>(a, s));
    ADX_REQUIRE(a.stats_part == nullptr, "conv2d_hs: statistics requested from a launch the pipelined kernel does not serve");
    return hs_launch_t<1, 3, 2, 2, false>(a, s);
  }
  if (L.k == 3 && L.stride == 2 && L.pad == 1) {
    if (a.x_cells || a.y_cells) {
      ADX_REQUIRE(ds && a.x_amax == nullptr && a.res == nullptr && a.d2s_cin == 0,
                  "conv2d_hs: cell-layout operands of the stride-2 conv need the fused downsample launch (no residual, no dynamic range)");
      if (a.x_cells && a.y_cells) return hs_launch_t<2, 3, 1, 1, true, true, true>(a, s);
      return a.x_cells ? hs_launch_t<2, 3, 1, 1, true, true, false>(a, s) : hs_launch_t<2, 3, 1, 1, true, false, true>(a, s);
    }
    return ds ? hs_launch_t<2, 3, 1, 1, true>(a, s) : hs_launch_t<2, 3, 1, 1, false>(a, s);
  }
  set_error("conv2d_hs: no kernel for k=%d stride=%d pad=%d%s", L.k, L.stride, L.pad, ds ? " with a fused downsample" : "");
  return ADX_ERR_INVALID;
}

}  // namespace adx

#ifdef ADX_HS_TRACE
extern "C" int adx_hs_trace_read(unsigned long long* host, int nwords) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(adx::g_hs_trace), (size_t)nwords * 8);
}
#endif
