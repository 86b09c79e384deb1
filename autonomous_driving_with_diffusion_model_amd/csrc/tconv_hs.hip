// Temporal (1-D) convolution on the fp16 matrix cores of gfx950 with fp32-grade results ("split fp16").
//
// Same operator family and the same fused epilogue as tconv.hip (Conv1dBlock = Conv1d -> GroupNorm -> Mish
// [+ time bias] [+ residual], Downsample1d, Upsample1d, 1x1 convs, the block Linears: modeling/helpers.py:77-112,
// modeling/temporal.py:23-55), but the multiply runs as v_mfma_f32_32x32x16_f16 on operands split like the
// perception convolutions (conv2d_hs.hip): x = hi + 2^-11 lo with hi = fp16(x), lo = fp16((x - hi) * 2^11); a product is
// three MFMAs -- hi*hi into one fp32 accumulator, hi*lo + lo*hi into a second one that is scaled by 2^-11 at the end
// (the dropped lo*lo term is 2^-22 relative).  fp16 x fp16 products are exact in the MFMA's fp32 datapath, so the
// result carries fp32 accumulation error only.  One 32x32x16 MFMA retires 32768 MACs in 32 cycles against 1024 MACs
// in 32 cycles for v_mfma_f32_16x16x4_f32: with 3 products per multiply the matrix time drops 10x, which turns these
// kernels from matrix-bound into what they should be at 32..4096 rows: bound by streaming the weights once.
//
// GEMM view: rows m = (sample, position), 32 per workgroup (= bt whole samples, so every GroupNorm group the tile
// touches is complete on chip); cols n = output channels, one or two 32-channel tiles (two when the GroupNorm group
// is 64 wide); k = (tap, input channel), 16 per MFMA.
//   A (activations): staged ONCE per workgroup into LDS as 16-byte cells of 8 consecutive channels of one position,
//     hi cell next to lo cell, row pitch an odd number of 16-byte units => a lane's fragment is one ds_read_b128 per
//     plane and the 32 rows of a read fall into distinct banks.  Only real positions are stored: a tap that falls
//     outside the sample (zero padding, odd phase of the transposed conv) reads one shared all-zero row.
//   B (weights): pre-split and pre-packed per (32-channel tile, K-step, plane) in fragment order, so a wave's load of
//     one K-step is a contiguous 2 KB; NW waves split the K-steps and keep PF of them in flight in a register ring.
//   The NW partial tiles meet in LDS in the output tensor's [sample][channel][pos] order and go through
//   tconv_epilogue (shared with the exact kernel): fixed summation order, bit-reproducible.
#include <algorithm>
#include <array>
#include <map>
#include <mutex>

#define ADX_TCONV_TRACE_TU
#include "tconv_internal.h"
#include "tconv_pack.h"

namespace adx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kLoScale = 2048.0f;          // 2^11
constexpr float kLoInv = 1.0f / 2048.0f;

struct HsArgs {
  TConvArgs t;          // geometry + io (ct, log2_ct, bt, ntiles, ck, cin_pad, ncb, nkb as in TConvArgs)
  int pitch16;          // LDS row pitch in 16-byte units (odd)
  int nrows;            // staged input rows per workgroup = bt * lin; row index nrows is the all-zero row
  int vec_stage;        // 16-byte staging loads (dense [B][C][L] inputs and enough items to occupy the workgroup)
  int fast_epi;         // hs_epilogue4 applies (lout >= 4, GroupNorm group of 64..256 elements or none)
  int pc;               // floats per (sample, channel) line of a partial tile: lout, or lout + 4 when that makes the
                        // 16-byte accesses of 32 lanes (one channel each) bank-conflict free (fast epilogue only)
  int ptile;            // floats per partial tile = bt * ct * pc
  int log2_lin, log2_nrows;
  // split of the reduction over workgroups (tiny batches: the grid of a 512-channel layer at B = 1 is 8 workgroups, each
  // streaming a 655 KB slab through one CU).  ksplit workgroups per (row tile, channel slab) take cper input channels each
  // and write their partial tile to `part`; tconv_hs_reduce_kernel sums them in a fixed order and runs the epilogue.  The
  // ordering between the two is the kernel boundary: no fences, no counters, nothing to get wrong.
  int ksplit, cper;
  float* part;
  // tickets != nullptr: no reduce launch.  Every workgroup publishes its partial tile with write-through (sc1) stores, drains
  // them, and one lane draws a ticket from the (row tile, slab)'s counter with a relaxed agent-scope atomic; the workgroup
  // that draws the last one reads all partial tiles back with sc1 loads (they bypass its L1; nothing needs a fence: the
  // hand-off recipe of cdna_hip_programming.md, Guideline 16) and runs the epilogue, then zeroes the counter for the next launch.
  // (Round 2 tried this with __threadfence() on both sides: a device-scope release / acquire pair writes back and invalidates
  // an XCD's L2, 18 us per layer.)
  unsigned* tickets;
  size_t part_bytes;
  // K-split kernel: taps [tap0, tap0 + ntap) are the ones that reach a real input position from some output position; the others
  // multiply the zero row only and are not walked (a k5 conv on 2 positions -- the deepest level at horizon 16: three of five
  // taps, 40 % of the weight stream and of the MFMAs gone; every tap is live from 3 positions on)
  int tap0, ntap;
};

__device__ __forceinline__ void split8(const float (&v)[8], h8& hi, h8& lo) {
#if defined(ADX_TCONV_STAGE_PROBE)
  // timing-only builds (csrc/build.sh -DADX_TCONV_STAGE_PROBE=1|2, garbage results): what the staging's fp32 -> hi / lo split costs
  // -- the upper bound of what pre-split activation tensors between the temporal layers could save (profiles/README.md, round 6)
  hi = __builtin_bit_cast(h8, f32x4{v[0], v[1], v[2], v[3]});
  lo = __builtin_bit_cast(h8, f32x4{v[4], v[5], v[6], v[7]});
#else
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * kLoScale);
  }
#endif
}

// Stage rows [0, nrows] x channels [c0, c0 + ckc) of this workgroup's samples into LDS as split cells.  Every global load
// is UNCONDITIONAL (clamped address, value zeroed afterwards): a guarded load makes the compiler wait for it before the
// next one is issued, i.e. eight dependent round trips per item instead of one.
template <int NT>
__device__ __forceinline__ void hs_stage(const TConvArgs& a, const HsArgs& ha, u32x4* cells, int c0, int ckc, int b0, int tid) {
  const int batch = a.io.batch;
  const int pitch = ha.pitch16;
  const int ncell = ckc >> 3;
  const int cmax = a.cin - 1, bmax = batch - 1;
  if (ha.vec_stage) {
    // item = (sample, quad of 4 positions, 8-channel octet): eight 16-byte loads -> four (hi, lo) cell pairs
    const int items = (ha.nrows >> 2) * ncell;
    for (int it = tid; it < items; it += NT) {
      const int rq = it & ((ha.nrows >> 2) - 1), oc = it >> (ha.log2_nrows - 2);  // row quads fastest
      const int q = rq & ((a.lin >> 2) - 1), sb = rq >> (ha.log2_lin - 2);
      const int b = b0 + sb, bc = min(b, bmax);
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = c0 + 8 * oc + j, cc = min(ci, cmax);
        const bool first = cc < a.c0;
        const float* base = first ? a.io.x0 : a.io.x1;
        const int64_t off = first ? (int64_t)cc * a.io.x0_sc + (int64_t)bc * a.io.x0_sb
                                  : (int64_t)(cc - a.c0) * a.io.x1_sc + (int64_t)bc * a.io.x1_sb;
#if defined(ADX_TCONV_STAGE_PROBE) && ADX_TCONV_STAGE_PROBE == 2
        v[j] = f32x4{(float)off, 1.f, 2.f, 3.f};          // (2: no global loads at all -- the whole staging round trip)
#else
        v[j] = *reinterpret_cast<const f32x4*>(base + off + 4 * q);
#endif
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!(c0 + 8 * oc + j < a.cin && b < batch)) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = v[j][p];
        h8 hi, lo;
        split8(t8, hi, lo);
        u32x4* dst = cells + (sb * a.lin + 4 * q + p) * pitch + 2 * oc;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[1] = __builtin_bit_cast(u32x4, lo);
      }
    }
  } else {
    // item = (row, 8-channel octet): eight 4-byte loads -> one (hi, lo) cell pair; rows fastest across lanes so that
    // a wave's load instruction covers consecutive positions of one channel
    const int nrows = ha.nrows;
    const int items = nrows * ncell;
    for (int it = tid; it < items; it += NT) {
      const int row = it & (nrows - 1), oc = it >> ha.log2_nrows;      // nrows and lin are powers of two
      const int sb = row >> ha.log2_lin, ip = row & (a.lin - 1);
      const int b = b0 + sb, bc = min(b, bmax);
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = c0 + 8 * oc + j, cc = min(ci, cmax);
        const bool first = cc < a.c0;
        const float* base = first ? a.io.x0 : a.io.x1;
        const int64_t off = first ? (int64_t)bc * a.io.x0_sb + (int64_t)cc * a.io.x0_sc + (int64_t)ip * a.io.x0_sl
                                  : (int64_t)bc * a.io.x1_sb + (int64_t)(cc - a.c0) * a.io.x1_sc + (int64_t)ip * a.io.x1_sl;
#if defined(ADX_TCONV_STAGE_PROBE) && ADX_TCONV_STAGE_PROBE == 2
        t8[j] = (float)off;
#else
        t8[j] = base[off];
#endif
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!(c0 + 8 * oc + j < a.cin && b < batch)) t8[j] = 0.f;
      h8 hi, lo;
      split8(t8, hi, lo);
      ADX_TSTAMP(10);
      u32x4* dst = cells + row * pitch + 2 * oc;
      dst[0] = __builtin_bit_cast(u32x4, hi);
      dst[1] = __builtin_bit_cast(u32x4, lo);
    }
  }
  for (int it = tid; it < 2 * ncell; it += NT) cells[ha.nrows * pitch + it] = u32x4{0u, 0u, 0u, 0u};
}

// ---- sums over segments of 16 / 32 / 64 consecutive lanes: DPP inside a row of 16, LDS permute across rows ------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float seg_sum(float v, int lanes) {   // every lane of a segment receives the segment's sum
  v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_f<0x141>(v);   // row_half_mirror: the other quad of each 8
  v += dpp_f<0x140>(v);   // row_mirror: the other half of the row
  if (lanes >= 32) v += __shfl_xor(v, 16, 64);
  if (lanes >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Mish with the hardware exp and reciprocal: e = 2^(x log2 e) (v_exp_f32, 1 ulp), n = e (e + 2), x n / (n + 2)
// (same closed form as mish_f; ~3e-7 relative, far inside the 1e-4 trajectory budget)
__device__ __forceinline__ float mish_fast(float x) {
  if (x > 20.f) return x;
  const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  const float n = e * (e + 2.f);
  return x * n * __builtin_amdgcn_rcpf(n + 2.f);
}

// Fused epilogue on the summed tile, four consecutive positions of one (sample, channel) per thread: no barrier, the
// GroupNorm statistics are segment sums inside a wave (a group = cg * lout / 4 consecutive lanes, 16 / 32 / 64).
// Valid when lout >= 4 and (no GroupNorm or 64 <= cg * lout <= 256); other geometries use tconv_epilogue.
template <int NW, int TILE, bool SC1 = false>   // NW = number of partial tiles at P (compile time), or 0: `nparts` of them (run time);
                                               // SC1: they were published by other workgroups of THIS launch: loads that bypass the L1
__device__ __forceinline__ void hs_epilogue4(const TConvArgs& a, const float* P, int pc, int ptile, int tid, int nt,
                                             int b0, int nparts = 0, size_t part_bytes = 0, const float* part_base = nullptr) {
  if (tid >= TILE / 4) return;
  // four consecutive elements of the [sample][channel][pos] tile: with lout >= 4 they are 4 positions of ONE channel;
  // with lout == 2 (the 512-channel level at horizon 16) they are 2 positions of channel cA and 2 of cA + 1, which lie in
  // the same GroupNorm group and, in a plain [B][C][L] tensor, still form one 16-byte run
  const int e0 = 4 * tid;
  const bool two = a.lout == 2;
  const int l0 = e0 & (a.lout - 1);
  const int cA = nt * a.ct + ((e0 >> a.log2_lout) & (a.ct - 1));
  const int cB = two ? cA + 1 : cA;
  const int b = b0 + (e0 >> (a.log2_lout + a.log2_ct));
  const bool live = b < a.io.batch && cA < a.cout;
  auto ch_of = [&](int k) { return (two && k >= 2) ? cB : cA; };
  auto l_of = [&](int k) { return two ? (k & 1) : l0 + k; };
  // the loads of the epilogue first: they land while the partial tiles are summed
  float biasA = 0.f, biasB = 0.f, gmA = 1.f, gmB = 1.f, beA = 0.f, beB = 0.f, tbA = 0.f, tbB = 0.f;
  f32x4 rs = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    if (a.io.bias != nullptr) { biasA = a.io.bias[cA]; biasB = a.io.bias[cB]; }
    if (a.groups > 0) { gmA = a.io.gamma[cA]; beA = a.io.beta[cA]; gmB = a.io.gamma[cB]; beB = a.io.beta[cB]; }
    if (a.io.tbias != nullptr) {
      tbA = a.io.tbias[(int64_t)b * a.io.tbias_stride + cA];
      tbB = a.io.tbias[(int64_t)b * a.io.tbias_stride + cB];
    }
    if (a.io.res != nullptr) {
      const float* rb = a.io.res + (int64_t)b * a.io.res_sb;
      const bool run = a.io.res_sl == 1 && a.io.res_sc == a.lout && (a.io.res_sb & 3) == 0 &&
                       (reinterpret_cast<uintptr_t>(a.io.res) & 15) == 0;        // 4 elements = one aligned 16-byte run
      if (run) {
        rs = *reinterpret_cast<const f32x4*>(rb + (int64_t)cA * a.lout + l0);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q] = rb[(int64_t)ch_of(q) * a.io.res_sc + (int64_t)l_of(q) * a.io.res_sl];
      }
    }
  }
  const float* pp = P + (e0 >> a.log2_lout) * pc + l0;     // line (sample, channel) of the partial tile (pc == lout when lout < 8)
  f32x4 v;
  if (SC1) {
    const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(part_base), 0, (int)part_bytes, 0x00020000);
    const int o0 = (int)((pp - part_base) * sizeof(float));
    v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, o0, 0, 16));          // aux 16 = sc1
    for (int w0 = 1; w0 < nparts; w0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int w = w0 + j < nparts ? w0 + j : 0;
        t[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(prs, o0 + w * ptile * (int)sizeof(float), 0, 16));
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (w0 + j < nparts) v += t[j];
    }
  } else {
  v = *reinterpret_cast<const f32x4*>(pp);
  if (NW > 0) {
#pragma unroll
    for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(pp + w * ptile);   // fixed order: deterministic
  } else {
    // the partial tiles of a split reduction (global memory, written by workgroups on other XCDs a moment ago): all loads of a
    // batch of 8 are issued before the first add, so the thread pays one memory round trip per batch, not one per tile; the
    // adds keep the index order
    for (int w0 = 1; w0 < nparts; w0 += 8) {
      f32x4 t[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int w = w0 + j < nparts ? w0 + j : 0;        // past the end: re-read tile 0 (valid), value unused
        t[j] = *reinterpret_cast<const f32x4*>(pp + (size_t)w * ptile);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (w0 + j < nparts) v += t[j];
    }
  }
  }
  v += f32x4{biasA, biasA, two ? biasB : biasA, two ? biasB : biasA};
  if (a.io.pre != nullptr && live)          // dense [B][cout][lout]: the four elements are contiguous in both cases
    *reinterpret_cast<f32x4*>(a.io.pre + ((int64_t)b * a.cout + cA) * a.lout + l0) = v;
  ADX_TSTAMP(5);
  f32x4 o = v;
  if (a.groups > 0) {
    const int n = a.cg << a.log2_lout;          // elements per (sample, group); n / 4 consecutive lanes hold them
    const float inv_n = 1.0f / (float)n;
    const float mean = seg_sum((v[0] + v[1]) + (v[2] + v[3]), n >> 2) * inv_n;
    const f32x4 d = v - mean;
    const float q = seg_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]), n >> 2);
    const float rstd = 1.0f / sqrtf(q * inv_n + a.eps);
    if (a.io.stats != nullptr && live && (e0 & (n - 1)) == 0) {
      float* st = a.io.stats + ((int64_t)b * a.groups + cA / a.cg) * 2;
      st[0] = mean;
      st[1] = rstd;
    }
    const float scA = rstd * gmA, scB = rstd * gmB;
    o[0] = mish_fast(d[0] * scA + beA);
    o[1] = mish_fast(d[1] * scA + beA);
    o[2] = mish_fast(d[2] * (two ? scB : scA) + (two ? beB : beA));
    o[3] = mish_fast(d[3] * (two ? scB : scA) + (two ? beB : beA));
  }
  ADX_TSTAMP(7);
  if (!live) return;
  o += f32x4{tbA, tbA, two ? tbB : tbA, two ? tbB : tbA};
  o += rs;
  float* yb = a.io.y + (int64_t)b * a.io.y_sb;
  const bool run = a.io.y_sl == 1 && a.io.y_sc == a.lout && (a.io.y_sb & 3) == 0 && (reinterpret_cast<uintptr_t>(a.io.y) & 15) == 0;
  if (run) {
    *reinterpret_cast<f32x4*>(yb + (int64_t)cA * a.lout + l0) = o;
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) yb[(int64_t)ch_of(k) * a.io.y_sc + (int64_t)l_of(k) * a.io.y_sl] = o[k];
  }
}

template <int NF, int NW, int PF>
__device__ __forceinline__ void tconv_hs_body(const HsArgs& ha, const int bid) {
  constexpr int NT = 64 * NW;
  const TConvArgs& a = ha.t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4* cells = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = bid % a.ntiles;   // blocks b, b + 8 share an XCD: with ntiles | 8 or 8 | ntiles one XCD's L2 serves one weight slab
  const int rest = bid / a.ntiles;
  const int kpart = ha.ksplit > 1 ? rest % ha.ksplit : 0;
  const int rowtile = ha.ksplit > 1 ? rest / ha.ksplit : rest;
  const int b0 = rowtile * a.bt;
  const int cbeg = ha.ksplit > 1 ? kpart * ha.cper : 0;
  const int cend = ha.ksplit > 1 ? min(a.cin_pad, cbeg + ha.cper) : a.cin_pad;
  const int r = lane & 31, kg = lane >> 5;
  const int bl = r >> a.log2_lout, l = r & (a.lout - 1);
  const int pitch = ha.pitch16;
  const int zrow = ha.nrows;

  ADX_TSTAMP(0);
  f32x16 accm[NF], accx[NF];
#pragma unroll
  for (int j = 0; j < NF; ++j)
#pragma unroll
    for (int i = 0; i < 16; ++i) { accm[j][i] = 0.f; accx[j][i] = 0.f; }

  // weight image: [32-channel tile][K-step][plane][lane] x 16 bytes.  Read through a buffer descriptor over this workgroup's
  // NF tiles: the lane supplies 16 * lane, the (wave-uniform) K-step is an SGPR offset -- one instruction per 16-byte load
  // and no per-lane 64-bit address arithmetic in the K loop (round 3: at this occupancy a wave's time is its instruction
  // count; the loop went from ~60 to ~30 instructions per step)
  const int tile_bytes = a.nkb * 2048;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<float*>(a.io.packed_w) + (size_t)nt * NF * a.nkb * 512, 0, NF * tile_bytes, 0x00020000);
  const int lane16 = lane * 16;
  u32x4 wq[PF][NF][2];

  for (int c0 = cbeg; c0 < cend; c0 += a.ck) {
    const int ckc = min(a.ck, cend - c0);
    const int ncbc = ckc >> 4;
    const int nblk = ha.ntap * ncbc;
    const int nbw = nblk > wave ? (nblk - wave + NW - 1) / NW : 0;   // K-steps of this wave in this chunk
    const int cb0 = c0 >> 4;
    // this wave's K-steps: i = wave, wave + NW, ...  (tap, 16-channel block) = (tap0 + i / ncbc, i % ncbc)
    int ltap = ha.tap0, lcb = wave;
    while (lcb >= ncbc) { lcb -= ncbc; ++ltap; }
    auto issue = [&](u32x4 (&dst)[NF][2]) {
      const int tp = min(ltap, a.taps - 1);  // past-the-end slots re-read a valid block and are never used
      const int so = (tp * a.ncb + cb0 + lcb) * 2048;          // wave-uniform
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        dst[j][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so + j * tile_bytes, 0);
        dst[j][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so + j * tile_bytes + 1024, 0);
      }
      lcb += NW;
      while (lcb >= ncbc) { lcb -= ncbc; ++ltap; }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) issue(wq[s]);     // in flight while the activations are staged
    ADX_TSTAMP(9);
    if (c0 > cbeg) __syncthreads();
    hs_stage<NT>(a, ha, cells, c0, ckc, b0, tid);
    ADX_TSTAMP(11);
    __syncthreads();
    ADX_TSTAMP(1);
    // ---- K loop over this wave's steps, PF-deep weight ring, activation fragments fetched one step ahead -------
    int ctap = ha.tap0, ccb = wave;
    while (ccb >= ncbc) { ccb -= ncbc; ++ctap; }
    u32x4 ah, al;
    const bool kind0 = a.kind == 0;
    // the tap of a step is wave-uniform, so the lane's LDS row is recomputed only when it changes (conv: ip = l stride + tap -
    // pad; ConvTranspose1d, stride 2: o = 2 i - pad + tap  <=>  i = (o + pad - tap) / 2 when that is even)
    auto row_of = [&](int tap) {
      const int vt = l + a.pad - tap;
      const int ip = kind0 ? l * a.stride + tap - a.pad : vt >> 1;
      const bool ok = ((unsigned)ip < (unsigned)a.lin) & (kind0 | ((vt & 1) == 0)) & (tap < a.taps);
      return cells + (ok ? bl * a.lin + ip : zrow) * pitch + 2 * kg;
    };
    const u32x4* rowp = row_of(ctap);
    auto fetch_a = [&]() {     // fragment of step (ctap, ccb); then advance to this wave's next step
      const u32x4* xp = rowp + 4 * ccb;
      ah = xp[0];
      al = xp[1];
      ccb += NW;
      if (ccb >= ncbc) {       // uniform
        do { ccb -= ncbc; ++ctap; } while (ccb >= ncbc);
        rowp = row_of(ctap);
      }
    };
    if (nbw > 0) fetch_a();
    auto compute = [&](const u32x4 (&w)[NF][2]) {
      const h8 ch = __builtin_bit_cast(h8, ah);
      const h8 cl = __builtin_bit_cast(h8, al);
      fetch_a();                        // next step's fragment is in flight under this step's MFMAs
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < NF; ++j) {
        const h8 wh = __builtin_bit_cast(h8, w[j][0]);
        const h8 wl = __builtin_bit_cast(h8, w[j][1]);
        accm[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, wh, accm[j], 0, 0, 0);
        accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ch, wl, accx[j], 0, 0, 0);
        accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cl, wh, accx[j], 0, 0, 0);
      }
    };
    int j0 = 0;
    for (; j0 + PF <= nbw; j0 += PF) {  // full groups: branch-free, ring slots are compile-time indices
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        compute(wq[s]);
        issue(wq[s]);
      }
    }
#pragma unroll
    for (int s = 0; s < PF; ++s)        // tail group: its weights are already in the ring
      if (j0 + s < nbw) compute(wq[s]);
  }

  // ---- the NW partial tiles -> LDS, [wave][sample][channel][pos] ------------------------------------------------
  ADX_TSTAMP(2);
  __syncthreads();
  ADX_TSTAMP(3);
  constexpr int TILE = 32 * 32 * NF;
  float* P = smem + wave * ha.ptile;
#pragma unroll
  for (int j = 0; j < NF; ++j) {
    const int c = j * 32 + r;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int m0 = 8 * g + 4 * kg;                      // 4 consecutive rows m0 .. m0 + 3 in registers 4g .. 4g + 3
      f32x4 o;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = accm[j][4 * g + q] + accx[j][4 * g + q] * kLoInv;
      if (a.log2_lout >= 2) {                             // the 4 rows are 4 consecutive positions of one sample
        const int sb = m0 >> a.log2_lout, l0 = m0 & (a.lout - 1);
        *reinterpret_cast<f32x4*>(P + ((sb << a.log2_ct) + c) * ha.pc + l0) = o;
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + q;
          const int sb = m >> a.log2_lout, lq = m & (a.lout - 1);
          P[(((sb << a.log2_ct) + c) << a.log2_lout) + lq] = o[q];
        }
      }
    }
  }
  __syncthreads();
  ADX_TSTAMP(4);
  if (ha.ksplit > 1) {
    // this workgroup's share of the reduction: the 8 waves' partial tiles summed, written in the (padded) tile layout
    if (tid < TILE / 4) {
      const int e0 = 4 * tid;
      const int off = (e0 >> a.log2_lout) * ha.pc + (e0 & (a.lout - 1));
      f32x4 v = *reinterpret_cast<const f32x4*>(smem + off);
#pragma unroll
      for (int w = 1; w < NW; ++w) v += *reinterpret_cast<const f32x4*>(smem + w * ha.ptile + off);
      const size_t poff = ((size_t)(rowtile * a.ntiles + nt) * ha.ksplit + kpart) * ha.ptile + off;
      if (ha.tickets == nullptr) {
        *reinterpret_cast<f32x4*>(ha.part + poff) = v;
      } else {
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(ha.part, 0, (int)ha.part_bytes, 0x00020000);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), prs, (int)(poff * sizeof(float)), 0, 16);   // sc1
      }
    }
    if (ha.tickets == nullptr) return;               // tconv_hs_reduce_kernel adds the tiles up
    // ---- ticket: the last workgroup of this (row tile, slab) to publish runs the epilogue -----------------------------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its write-through stores ...
    __syncthreads();                                  // ... before ONE lane signals
    int* last = reinterpret_cast<int*>(smem);         // (the partial tiles in LDS are dead: every wave passed the barrier)
    if (tid == 0) {
      unsigned* ctr = ha.tickets + (rowtile * a.ntiles + nt);
      const unsigned t = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int is_last = t == (unsigned)(ha.ksplit - 1);
      if (is_last) __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      *last = is_last;
    }
    __syncthreads();
    if (*last == 0) return;
    const float* P = ha.part + (size_t)(rowtile * a.ntiles + nt) * ha.ksplit * ha.ptile;
    hs_epilogue4<0, TILE, true>(a, P, ha.pc, ha.ptile, tid, nt, b0, ha.ksplit, ha.part_bytes, ha.part);
    return;
  }
  if (ha.fast_epi) {
    hs_epilogue4<NW, TILE>(a, smem, ha.pc, ha.ptile, tid, nt, b0);
    ADX_TSTAMP(8);
  } else {
    tconv_epilogue<NT, NW, TILE>(a, smem, tid, lane, wave, nt, b0);
  }
}

template <int NF, int NW, int PF>
__global__ void __launch_bounds__(64 * NW) tconv_hs_kernel(const HsArgs ha) {
  tconv_hs_body<NF, NW, PF>(ha, (int)blockIdx.x);
}

// second half of a split reduction (HsArgs::ksplit > 1): one workgroup per (row tile, channel slab)
template <int NF>
__global__ void __launch_bounds__(256 * NF) tconv_hs_reduce_kernel(const HsArgs ha) {
  constexpr int TILE = 32 * 32 * NF;
  const TConvArgs& a = ha.t;
  const int nt = blockIdx.x % a.ntiles, rowtile = blockIdx.x / a.ntiles;
  const float* P = ha.part + (size_t)blockIdx.x * ha.ksplit * ha.ptile;
  hs_epilogue4<0, TILE>(a, P, ha.pc, ha.ptile, threadIdx.x, nt, rowtile * a.bt, ha.ksplit);
}

// ---- short-K variant: no K split --------------------------------------------------------------------------------
// For the layers whose reduction is short (taps * cin <= 1536: every conv of the 64/128-channel levels, the 1x1 and
// down/up-sampling convs, the block Linears -- 29 of the 47 launches of a forward) splitting K over the waves buys
// nothing and costs the partial-tile round trip through LDS (write, barrier, 8-way sum: ~1800 cycles of a ~10000-cycle
// workgroup).  Here every wave owns a 16 x 16 corner of the 32-row x 32-channel tile with the WHOLE reduction
// (v_mfma_f32_16x16x32_f16: 4 accumulator registers = 4 consecutive rows = 4 consecutive positions of one channel), and
// the epilogue runs straight from those registers.  K is the flattened sequence of 8-channel cells (tap-major), four
// cells per MFMA step, so a step may straddle taps when cin < 32.  GroupNorm statistics: lane shuffles inside the wave,
// one LDS exchange with the partner waves when a (sample, group) spans the tile's two row halves (lout = 32) or its two
// channel halves (group width 32); two passes (mean, centred second moment), fixed summation order.
#ifndef ADX_HSD_PF
#define ADX_HSD_PF 6
#endif
constexpr int kHsdPF = ADX_HSD_PF;   // depth of the short-K kernel's weight-fragment ring

struct HsdArgs {
  HsArgs h;
  int nsteps;           // MFMA steps = ceil(taps * cin_pad / 8 / 4)
  int kcells;           // taps * cin_pad / 8
  int log2_ncell;       // cells per tap = cin_pad / 8 (a power of two)
  int red_off;          // float offset of the GroupNorm exchange area [2][4][64] behind the staged cells
};

// Two independent short-K convolutions may share ONE launch (workgroups [0, n_a) run `a`, the rest run `b`): the 1x1
// residual conv of a residual block and the block's first Conv1dBlock read the same input and nothing of each other
// (modeling/temporal.py:51-55), so the pair costs one launch latency instead of two.
struct HsdPair {
  HsdArgs a, b;
  int n_a;              // workgroups of `a`; a single conv has n_a = its whole grid
};

template <int PF, bool UT>
__device__ __forceinline__ void tconv_hsd_body(const HsdArgs& da, const int bid) {
  constexpr int NT = 256;
  const HsArgs& ha = da.h;
  const TConvArgs& a = ha.t;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  u32x4* cells = reinterpret_cast<u32x4*>(smem);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave & 1, wc = wave >> 1;          // row half, channel half of the 32 x 32 tile
  const int nt = bid % a.ntiles;
  const int b0 = (bid / a.ntiles) * a.bt;
  const int batch = a.io.batch;
  const int r = lane & 15, kg = lane >> 4;
  const int ma = 16 * wr + r;                       // this lane's row as an A operand
  const int bl = ma >> a.log2_lout, l = ma & (a.lout - 1);
  const int pitch = ha.pitch16;
  const int zrow = ha.nrows;
  ADX_TSTAMP(0);
  // weight image: [16-channel tile][step][plane][lane] x 16 bytes
  const u32x4* __restrict__ wp = reinterpret_cast<const u32x4*>(a.io.packed_w) + (size_t)(nt * 2 + wc) * da.nsteps * 128 + lane;
  u32x4 wq[PF][2];
#pragma unroll
  for (int s = 0; s < PF; ++s) {
    const int st = min(s, da.nsteps - 1);
    wq[s][0] = wp[(size_t)st * 128];
    wq[s][1] = wp[(size_t)st * 128 + 64];
  }
  ADX_TSTAMP(9);
  hs_stage<NT>(a, ha, cells, 0, a.cin_pad, b0, tid);     // the short-K layers fit one chunk
  ADX_TSTAMP(11);
  __syncthreads();
  ADX_TSTAMP(1);
  // ---- K loop: all steps on this wave's 16 x 16 corner, weights PF steps ahead, activations one step ahead ---------
  f32x4 accm = f32x4{0.f, 0.f, 0.f, 0.f}, accx = f32x4{0.f, 0.f, 0.f, 0.f};
  u32x4 ah, al;
  const bool kind0 = a.kind == 0;
  const int ncm1 = (1 << da.log2_ncell) - 1;
  // UT (cin_pad >= 32): the four cells of a step lie in ONE tap, so the tap and the first cell are wave-uniform and move
  // incrementally -- the lane's LDS row is recomputed only when the tap changes; otherwise (cin_pad = 16) each lane has
  // its own (tap, cell) per step
  int f_tap = 0, f_cell = 0;                         // UT: position of the NEXT fetch
  int f_row = zrow;
  auto row_of = [&](int tap) {
    const int vt = l + a.pad - tap;
    const int ip = kind0 ? l * a.stride + tap - a.pad : vt >> 1;
    const bool ok = ((unsigned)ip < (unsigned)a.lin) & (kind0 | ((vt & 1) == 0)) & (tap < a.taps);
    return ok ? bl * a.lin + ip : zrow;
  };
  if (UT) f_row = row_of(0);
  auto fetch_a = [&](int step) {
    const u32x4* xp;
    if (UT) {
      xp = cells + f_row * pitch + 2 * (f_cell + kg);
      f_cell += 4;
      if (f_cell > ncm1) {                           // uniform: next tap
        f_cell = 0;
        ++f_tap;
        f_row = row_of(f_tap);
      }
    } else {
      const int kc = 4 * step + kg;                  // flattened (tap, cell)
      const int tap = kc >> da.log2_ncell, cell = kc & ncm1;
      xp = cells + row_of(tap) * pitch + 2 * cell;
    }
    ah = xp[0];
    al = xp[1];
  };
  fetch_a(0);
  auto compute = [&](const u32x4 (&w)[2], int next_step) {
    const h8 ch = __builtin_bit_cast(h8, ah);
    const h8 cl = __builtin_bit_cast(h8, al);
    fetch_a(next_step);
    __builtin_amdgcn_sched_barrier(0);    // the next step's LDS reads stay in front of this step's MFMAs
    const h8 wh = __builtin_bit_cast(h8, w[0]);
    const h8 wl = __builtin_bit_cast(h8, w[1]);
    accm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wh, accm, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wl, accx, 0, 0, 0);
    accx = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, wh, accx, 0, 0, 0);
  };
  int j0 = 0;
  for (; j0 + PF <= da.nsteps; j0 += PF) {
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int nxt = min(j0 + s + 1, da.nsteps - 1);
      compute(wq[s], nxt);
      const int st = min(j0 + s + PF, da.nsteps - 1);
      wq[s][0] = wp[(size_t)st * 128];
      wq[s][1] = wp[(size_t)st * 128 + 64];
    }
  }
#pragma unroll
  for (int s = 0; s < PF; ++s)
    if (j0 + s < da.nsteps) compute(wq[s], min(j0 + s + 1, da.nsteps - 1));
  ADX_TSTAMP(2);

  // ---- epilogue from the accumulators: lane = (channel lane & 15, four consecutive rows 4 kg .. 4 kg + 3) ------------
  const int c = nt * a.ct + 16 * wc + r;
  const int m0 = 16 * wr + 4 * kg;
  if (a.lout < 4) {
    // rows are (sample, position) pairs with fewer than 4 positions per sample (the block Linears: lout = 1): the four
    // accumulator rows belong to different samples; no GroupNorm on this path (hsd_geometry), one element at a time
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = m0 + q;
      const int b = b0 + (m >> a.log2_lout), lq = m & (a.lout - 1);
      if (b < batch && c < a.cout) {
        float o = accm[q] + accx[q] * kLoInv;
        if (a.io.bias != nullptr) o += a.io.bias[c];
        if (a.io.pre != nullptr) a.io.pre[((int64_t)b * a.cout + c) * a.lout + lq] = o;
        if (a.io.tbias != nullptr) o += a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
        if (a.io.res != nullptr) o += a.io.res[(int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)lq * a.io.res_sl];
        a.io.y[(int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)lq * a.io.y_sl] = o;
      }
    }
    return;
  }
  const int sb = m0 >> a.log2_lout, l0 = m0 & (a.lout - 1);
  const int b = b0 + sb;
  const bool live = b < batch && c < a.cout;
  float bias = 0.f, gm = 1.f, be = 0.f, tb = 0.f;
  f32x4 rs = f32x4{0.f, 0.f, 0.f, 0.f};
  if (live) {
    if (a.io.bias != nullptr) bias = a.io.bias[c];
    if (a.groups > 0) { gm = a.io.gamma[c]; be = a.io.beta[c]; }
    if (a.io.tbias != nullptr) tb = a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
    if (a.io.res != nullptr) {
      const float* rp = a.io.res + (int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)l0 * a.io.res_sl;
      if (a.io.res_sl == 1 && ((a.io.res_sb | a.io.res_sc) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.io.res) & 15) == 0) {
        rs = *reinterpret_cast<const f32x4*>(rp);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) rs[q] = rp[(int64_t)q * a.io.res_sl];
      }
    }
  }
  f32x4 v = accm + accx * kLoInv;
  v += bias;
  if (a.io.pre != nullptr && live) *reinterpret_cast<f32x4*>(a.io.pre + ((int64_t)b * a.cout + c) * a.lout + l0) = v;
  ADX_TSTAMP(5);
  f32x4 o = v;
  if (a.groups > 0) {
    // (sample, group) of this lane: channel lanes that share the group = bits 0..2 (and 3 when cg >= 16); row quads that
    // share the sample = bit 4 when lout >= 8, bit 5 when lout >= 16; partner waves: wr ^ 1 when lout == 32, wc ^ 1 when cg == 32
    float* red = smem + da.red_off;                  // behind the staged cells: [2 passes][4 waves][64 lanes]
    const bool x_rows = a.lout == 32, x_ch = a.cg == 32;
    auto group_sum = [&](float s, int pass) -> float {
      s += dpp_f<0xB1>(s);
      s += dpp_f<0x4E>(s);
      s += dpp_f<0x141>(s);                          // bits 0..2: the 8 channel lanes
      if (a.cg >= 16) s += dpp_f<0x140>(s);          // bit 3
      if (a.lout >= 8) s += __shfl_xor(s, 16, 64);
      if (a.lout >= 16) s += __shfl_xor(s, 32, 64);
      if (x_rows || x_ch) {                          // uniform branch: same for every lane of the workgroup
        float* rp = red + pass * 256;
        rp[wave * 64 + lane] = s;
        __syncthreads();
        const int w0 = (x_rows ? 0 : wr) + 2 * (x_ch ? 0 : wc);     // lowest partner wave: canonical summation order
        float t = rp[w0 * 64 + lane];
        if (x_rows) t += rp[(w0 + 1) * 64 + lane];
        if (x_ch) {
          t += rp[(w0 + 2) * 64 + lane];
          if (x_rows) t += rp[(w0 + 3) * 64 + lane];
        }
        s = t;
      }
      return s;
    };
    const int n = a.cg << a.log2_lout;
    const float inv_n = 1.0f / (float)n;
    const float mean = group_sum((v[0] + v[1]) + (v[2] + v[3]), 0) * inv_n;
    const f32x4 d = v - mean;
    const float q = group_sum((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]), 1);
    const float rstd = 1.0f / sqrtf(q * inv_n + a.eps);
    if (a.io.stats != nullptr && live && l0 == 0 && (c & (a.cg - 1)) == 0) {
      float* st = a.io.stats + ((int64_t)b * a.groups + c / a.cg) * 2;
      st[0] = mean;
      st[1] = rstd;
    }
    const float sc = rstd * gm;
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] = mish_fast(d[k] * sc + be);
  }
  ADX_TSTAMP(7);
  if (live) {
    o += tb;
    o += rs;
    float* yp = a.io.y + (int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)l0 * a.io.y_sl;
    if (a.io.y_sl == 1 && ((a.io.y_sb | a.io.y_sc) & 3) == 0 && (reinterpret_cast<uintptr_t>(a.io.y) & 15) == 0) {
      *reinterpret_cast<f32x4*>(yp) = o;
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) yp[(int64_t)k * a.io.y_sl] = o[k];
    }
  }
  ADX_TSTAMP(8);
}

template <int PF, bool UT>
__global__ void __launch_bounds__(256) tconv_hsd_kernel(const HsdArgs da) {
  tconv_hsd_body<PF, UT>(da, (int)blockIdx.x);
}

// the paired form: two bodies behind one wave-uniform branch, so that a workgroup loads only ITS argument block (selecting
// between the two blocks field by field doubled the kernel-argument loads of every launch: +1 us each, measured)
template <int PF, bool UT>
__global__ void __launch_bounds__(256) tconv_hsd_pair_kernel(const HsdPair pr) {
  if ((int)blockIdx.x < pr.n_a) tconv_hsd_body<PF, UT>(pr.a, (int)blockIdx.x);
  else tconv_hsd_body<PF, UT>(pr.b, (int)blockIdx.x - pr.n_a);
}

// A K-split conv and a short-K conv that read the same input and nothing of each other in ONE launch (a residual block's
// first conv beside its 1x1 residual conv where the first is too long for the short-K kernel: the 256/512-channel levels).
// The short-K workgroups use the first four of the launch's eight waves.
struct HsMixed {
  HsArgs a;
  HsdArgs b;
  int n_a;
};
template <int NF, int NW, int PF, bool UT>
__global__ void __launch_bounds__(64 * NW) tconv_hs_mixed_kernel(const HsMixed pr) {
  if ((int)blockIdx.x < pr.n_a) {
    tconv_hs_body<NF, NW, PF>(pr.a, (int)blockIdx.x);
  } else {
    if (threadIdx.x >= 256) return;          // (a finished wave no longer takes part in the workgroup's barriers)
    tconv_hsd_body<kHsdPF, UT>(pr.b, (int)blockIdx.x - pr.n_a);
  }
}

// weight image of the short-K variant: [cout_pad32 / 16][nsteps][2 planes][64 lanes][8 halfs]; element j of lane ln at
// `step` is W[n = 16 tile + (ln & 15)][flattened cell kc = 4 step + (ln >> 4): tap = kc / ncell, ci = 8 (kc % ncell) + j]
// (tconv_pack.hip, kPackCell); weight image of the K-split kernel: [cout_pad32 / 32][nkb][2 planes][64 lanes][8 halfs]; element j
// of lane ln in K-step (tap, cb) is W[n = 32 tile + (ln & 31)][ci = 16 cb + 8 (ln >> 5) + j][tap]  (B operand of 32x32x16;
// kPackHs)

static int ilog2_exact_hs(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

constexpr size_t kMaxHsLds = 144 * 1024;

bool tconv_hs_supported(const adx_tconv_desc* d) {
  if (debug_switches().tconv_exact || d->exact != 0) return false;
  if ((d->lin_valid > 0 && d->lin_valid != d->lin) || (d->lout_valid > 0 && d->lout_valid != d->lout)) return false;
  if (d->lout > 32 || ilog2_exact_hs(d->lout) < 0 || ilog2_exact_hs(d->lin) < 0) return false;   // 32 rows per tile = whole samples
  if (32 % d->lout != 0) return false;
  if (d->groups > 0) {
    const int cg = d->cout / d->groups;
    if (cg > 64 || ilog2_exact_hs(cg) < 0) return false;
    if ((cg * d->lout) % 64 != 0 || d->cout % 16 != 0) return false;     // both epilogues reduce a (sample, group) in whole waves
  }
  return true;
}

// short-K variant (tconv_hsd_kernel): geometry it covers, decided from the descriptor alone so that pack and forward agree
static bool hsd_geometry(const adx_tconv_desc* d, int* nsteps, int* kcells, int* log2_ncell) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  const int ncell = cin_pad / 8;
  const int lg = ilog2_exact_hs(ncell);
  if (lg < 0 || d->lout > 32) return false;
  if (d->groups > 0) {
    const int cg = d->cout / d->groups;
    if (d->lout < 4 || (cg != 8 && cg != 16 && cg != 32)) return false;   // lout < 4 (the block Linears: lout = 1) without GroupNorm only
  }
  const int kc = d->taps * ncell, ns = ceil_div(kc, 4);
  if (ns > 48) return false;                                                   // taps * cin_pad <= 1536
  const int nrows = (32 / d->lout) * d->lin;
  if ((size_t)(nrows + 1) * (cin_pad / 4 + 1) * 16 > 60 * 1024) return false;  // one staging chunk
  *nsteps = ns; *kcells = kc; *log2_ncell = lg;
  return true;
}

static bool hsd_enabled() { return true; }

// true when tconv_pack leaves this layer's weights in the K-split kernel's image ([cout / 32][tap x cin / 16][plane][lane][8 halfs])
bool tconv_hs_kernel_image(const adx_tconv_desc* d) {
  int ns, kc, lg;
  return tconv_hs_supported(d) && !(hsd_enabled() && hsd_geometry(d, &ns, &kc, &lg));
}

size_t tconv_hs_packed_floats(const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  size_t f = (size_t)round_up(d->cout, 32) * d->taps * cin_pad;     // 2 halfs = 4 bytes per (padded) weight
  int ns, kc, lg;
  if (hsd_geometry(d, &ns, &kc, &lg)) f = std::max(f, (size_t)round_up(d->cout, 32) * ns * 32);
  return f;
}

int tconv_hs_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s) {
  {
    int ns, kc, lg;
    if (hsd_enabled() && hsd_geometry(d, &ns, &kc, &lg)) {
      const size_t total = (size_t)(round_up(d->cout, 32) / 16) * ns * 512;
      PackJob j{};
      j.w = w; j.out = packed; j.total = (uint32_t)total; j.kind = kPackCell;
      j.layout = d->kind == 1 ? 1 - d->w_layout : d->w_layout; j.flip = d->w_flip; j.taps = d->taps; j.cin = d->c0 + d->c1;
      j.cout = d->cout; j.a = 1 << lg; j.b = ns; j.tile_steps = ns; j.step0 = 0;
      return pack_submit(j, s);
    }
  }
  const int cin = d->c0 + d->c1;
  const int ncb = round_up(cin, 16) / 16;
  const int nkb = d->taps * ncb;
  const size_t total = (size_t)(round_up(d->cout, 32) / 32) * nkb * 512;
  PackJob j{};
  j.w = w; j.out = packed; j.total = (uint32_t)total; j.kind = kPackHs;
  j.layout = d->kind == 1 ? 1 - d->w_layout : d->w_layout; j.flip = d->w_flip; j.taps = d->taps; j.cin = cin; j.cout = d->cout;
  j.a = ncb; j.b = nkb;
  return pack_submit(j, s);
}

struct HsTile {
  int nf, nw, bt, ct, ntiles, ck, pitch16, nrows;
  size_t lds_bytes;
};

static int hs_tile(const adx_tconv_desc* d, int batch, HsTile* t) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  const int cout_pad = round_up(d->cout, 32);
  t->bt = 32 / d->lout;
  t->nf = 1;
  if (d->groups > 0 && d->cout / d->groups == 64) t->nf = 2;
  t->ct = 32 * t->nf;
  ADX_REQUIRE(cout_pad % t->ct == 0, "tconv_hs: cout %d not divisible by the channel tile %d", d->cout, t->ct);
  t->ntiles = cout_pad / t->ct;
  t->nrows = t->bt * d->lin;
  t->nw = 8;     // also where K is short: the staging and the epilogue are spread over 512 threads
  // staged chunk: (nrows + 1) rows x (ck / 4 + 1) 16-byte units; <= ~68 KB so that two workgroups share a CU, unless the
  // grid puts at most one workgroup on a CU anyway: then one chunk of up to 140 KB saves the second staging phase
  const bool one_per_cu = (size_t)ceil_div(batch, t->bt) * t->ntiles <= 256;
  const size_t budget = (one_per_cu ? 140 : 70) * 1024;
  int ck = cin_pad;
  while ((size_t)(t->nrows + 1) * (ck / 4 + 1) * 16 > budget && ck > 16) ck = round_up(ck / 2, 16);
  const int nchunks = ceil_div(cin_pad, ck);
  t->ck = round_up(ceil_div(cin_pad, nchunks), 16);
  t->pitch16 = t->ck / 4 + 1;
  const size_t stage = (size_t)(t->nrows + 1) * t->pitch16 * 16;
  const size_t epi = ((size_t)t->nw * 1024 * t->nf + 2 * 16 * t->nf) * sizeof(float);
  t->lds_bytes = stage > epi ? stage : epi;
  ADX_REQUIRE(t->lds_bytes <= kMaxHsLds, "tconv_hs: LDS tile of %zu bytes exceeds %zu", t->lds_bytes, kMaxHsLds);
  (void)batch;
  return ADX_OK;
}

template <int NF, int NW, int PF>
static int hs_launch(const HsArgs& a, int grid, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};  // dynamic LDS above 64 KB must be opted into once per kernel
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hs_kernel<NF, NW, PF>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxHsLds));
    once.commit();
  }
  tconv_hs_kernel<NF, NW, PF><<<dim3(grid), dim3(64 * NW), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

// short reduction: 32-row x 32-channel tiles, four waves with the whole K each (no partial tiles)
static bool hsd_prepare(const adx_tconv_desc* d, const HsArgs& ha, const HsTile& t, HsdArgs* da, size_t* lds, int* grid) {
  if (!hsd_enabled() || !hsd_geometry(d, &da->nsteps, &da->kcells, &da->log2_ncell)) return false;
  const TConvArgs& a = ha.t;
  HsArgs& h = da->h;
  h = ha;
  h.t.ct = 32; h.t.log2_ct = 5; h.t.ntiles = round_up(d->cout, 32) / 32;
  h.t.ck = a.cin_pad;
  h.pitch16 = a.cin_pad / 4 + 1;
  h.vec_stage = a.dense && (t.nrows / 4) * (a.cin_pad / 8) >= 128;
  const size_t stage = (size_t)(t.nrows + 1) * h.pitch16 * 16;
  da->red_off = (int)(round_up((int)stage, 16) / 4);
  *lds = (size_t)da->red_off * 4 + 2 * 4 * 64 * sizeof(float);
  *grid = ceil_div(a.io.batch, t.bt) * h.t.ntiles;
  return true;
}

static int hsd_launch(const HsdArgs& da, int grid, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hsd_kernel<kHsdPF, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hsd_kernel<kHsdPF, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    once.commit();
  }
  if (da.log2_ncell >= 2) tconv_hsd_kernel<kHsdPF, true><<<dim3(grid), dim3(256), lds, s>>>(da);
  else tconv_hsd_kernel<kHsdPF, false><<<dim3(grid), dim3(256), lds, s>>>(da);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

static int hsd_launch_pair(const HsdPair& pr, int grid, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hsd_pair_kernel<kHsdPF, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hsd_pair_kernel<kHsdPF, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    once.commit();
  }
  if (pr.a.log2_ncell >= 2) tconv_hsd_pair_kernel<kHsdPF, true><<<dim3(grid), dim3(256), lds, s>>>(pr);
  else tconv_hsd_pair_kernel<kHsdPF, false><<<dim3(grid), dim3(256), lds, s>>>(pr);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

static int hs_prepare(const adx_tconv_desc* d, const adx_tconv_io* io, HsTile* tp, HsArgs* hap);

static int hs_prepare(const adx_tconv_desc* d, const adx_tconv_io* io, HsTile* tp, HsArgs* hap) {
  HsTile& t = *tp;
  int rc = hs_tile(d, io->batch, &t);
  if (rc != ADX_OK) return rc;
  HsArgs& ha = *hap;
  TConvArgs& a = ha.t;
  a.io = *io;
  a.kind = d->kind; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
  a.c0 = d->c0; a.cin = d->c0 + d->c1; a.cout = d->cout; a.lin = d->lin; a.lout = d->lout;
  a.lin_valid = d->lin; a.lout_valid = d->lout;
  a.log2_lout = ilog2_exact_hs(d->lout);
  a.groups = d->groups; a.cg = d->groups > 0 ? d->cout / d->groups : 1; a.eps = d->eps;
  a.cin_pad = round_up(a.cin, 16);
  a.ncb = a.cin_pad / 16; a.nkb = d->taps * a.ncb;
  a.bt = t.bt; a.ct = t.ct; a.log2_ct = ilog2_exact_hs(t.ct); a.pl = 0; a.lp = 0; a.rs = 0; a.ck = t.ck;
  a.ntiles = t.ntiles;
  auto dense_src = [&](const float* p, int64_t sb, int64_t sc, int64_t sl) {
    return sl == 1 && sc % 4 == 0 && sb % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  a.dense = d->lin % 4 == 0 && dense_src(io->x0, io->x0_sb, io->x0_sc, io->x0_sl) &&
            (d->c1 == 0 || dense_src(io->x1, io->x1_sb, io->x1_sc, io->x1_sl));
  ha.pitch16 = t.pitch16;
  ha.nrows = t.nrows;
  ha.tap0 = 0; ha.ntap = d->taps;
  if (d->kind == 0) {      // ip = l stride + tap - pad lies in [0, lin) for some l in [0, lout)
    const int lo = std::max(0, d->pad - (d->lout - 1) * d->stride), hi = std::min(d->taps - 1, d->pad + d->lin - 1);
    if (hi >= lo) { ha.tap0 = lo; ha.ntap = hi - lo + 1; }
  }
  ha.vec_stage = a.dense && t.bt * (d->lin / 4) * (t.ck / 8) >= 256;
  const int n_gn = a.cg * d->lout;
  ha.fast_epi = d->lout >= 2 && (d->groups == 0 || (n_gn >= 64 && n_gn <= 256)) && (d->lout >= 4 || d->cout % 2 == 0);
  ha.pc = (ha.fast_epi && d->lout >= 8) ? d->lout + 4 : d->lout;     // (lout + 4) / 4 is odd for lout = 8, 16, 32
  ha.ptile = t.bt * t.ct * ha.pc;
  ha.log2_lin = ilog2_exact_hs(d->lin);
  ha.log2_nrows = ilog2_exact_hs(t.nrows);
  ADX_REQUIRE(ha.log2_lin >= 0 && ha.log2_nrows >= 0, "tconv_hs: lin %d must be a power of two", d->lin);
  const size_t epi_bytes = ((size_t)t.nw * ha.ptile + 2 * 16 * t.nf) * sizeof(float);
  if (epi_bytes > t.lds_bytes) t.lds_bytes = epi_bytes;
  ADX_REQUIRE(t.lds_bytes <= kMaxHsLds, "tconv_hs: LDS tile of %zu bytes exceeds %zu", t.lds_bytes, kMaxHsLds);
  ha.ksplit = 1; ha.cper = a.cin_pad; ha.part = nullptr; ha.tickets = nullptr; ha.part_bytes = 0;
  return ADX_OK;
}

// How the K-split kernel serves one conv: its arguments with the split decision made, the grid, and whether a reduce launch
// has to follow (a split without ticket words).
struct HsPlan {
  HsTile t;
  HsArgs ha;
  int grid;            // workgroups of the main launch (the unsplit grid x ksplit)
  int base_grid;       // (row tile, slab) pairs
  bool reduce;         // tconv_hs_reduce_kernel must run after it
};

static int hs_plan(const adx_tconv_desc* d, const adx_tconv_io* io, HsPlan* p) {
  HsTile& t = p->t;
  HsArgs& ha = p->ha;
  int rc = hs_prepare(d, io, &t, &ha);
  if (rc != ADX_OK) return rc;
  const int grid = ceil_div(io->batch, t.bt) * t.ntiles;
  p->grid = grid; p->base_grid = grid; p->reduce = false;
  // a grid that fits the chip one workgroup per CU must not be packed two per CU (the dispatcher does that with
  // 128 workgroups on 256 CUs): the two would share one CU's L2->L1 fill rate, which is what bounds the K loop
  // (forcing that through the LDS size was measured: no gain)
  // tiny batches: split the input channels over up to 16 workgroups per (row tile, slab) + one reduce launch
  constexpr bool split_on = true;
  const TConvArgs& a = ha.t;
  // Measured at 2 rows, H = 16 (tools/bench_tconv.py, SCRATCH=1 vs 0): 512->512 x5 15.1 -> 10.9 us, 1024->256 x5 20.2 -> 9.6,
  // 1024->256 x1 13.4 -> 8.8; but 512->128 x5 9.4 -> 10.5, 512->128 x1 5.7 -> 9.9, 256->256 x3 7.1 -> 8.5: the second launch
  // costs ~4 us, so only reductions of >= 256 K-steps per workgroup (x2 fragments at 64 channels/group) or >= 1024 staged
  // channels are split.
  const bool worth = a.taps * a.ncb * t.nf >= 256 || a.cin_pad >= 1024;
  // Round 3: with the ticket reduction the second launch is gone, and a grid of 33..128 workgroups (the 512-channel layers at
  // UNet batch 128: 128 workgroups on 256 CUs, each streaming a 655 KB weight slab through one CU's L1) takes a split in TWO:
  // 341 -> 332 us per forward at 128 rows (tools/chain_time.py); in four: 395 us (the partial tiles' round trip).
  constexpr int split_grid = 128;
  if (split_on && worth && io->scratch != nullptr && ha.fast_epi && grid <= split_grid && a.ncb >= 8 &&
      (grid <= 32 || io->tickets != nullptr)) {
    int S = grid <= 32 ? 16 : 2;
    while (S > 1 && (a.ncb % S != 0 || a.ncb / S < 2 || (size_t)grid * S * ha.ptile > (size_t)io->scratch_floats || grid * S > 512)) S >>= 1;
    if (S > 1) {
      ha.ksplit = S;
      ha.cper = (a.ncb / S) * 16;          // the staged chunk now holds cper channels at most
      ha.part = io->scratch;
      ha.part_bytes = (size_t)grid * S * ha.ptile * sizeof(float);
      p->grid = grid * S;
      if (io->tickets != nullptr && grid <= 256 && ha.part_bytes < 0x7FFFFFFFu) {
        // one launch: the last workgroup of each (row tile, slab) to publish its partial tile adds them up (HsArgs::tickets)
        ha.tickets = io->tickets;
      } else {
        p->reduce = true;
      }
    }
  }
  return ADX_OK;
}

int tconv_hs_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s) {
  {
    HsTile t;
    HsArgs ha;
    int rc = hs_prepare(d, io, &t, &ha);
    if (rc != ADX_OK) return rc;
    HsdArgs da;
    size_t lds;
    int grid_d;
    if (hsd_prepare(d, ha, t, &da, &lds, &grid_d)) return hsd_launch(da, grid_d, lds, s);
  }
  HsPlan p;
  int rc = hs_plan(d, io, &p);
  if (rc != ADX_OK) return rc;
  rc = p.t.nf == 2 ? hs_launch<2, 8, 4>(p.ha, p.grid, p.t.lds_bytes, s) : hs_launch<1, 8, 6>(p.ha, p.grid, p.t.lds_bytes, s);
  if (rc != ADX_OK || !p.reduce) return rc;
  if (p.t.nf == 2) tconv_hs_reduce_kernel<2><<<dim3(p.base_grid), dim3(512), 0, s>>>(p.ha);
  else tconv_hs_reduce_kernel<1><<<dim3(p.base_grid), dim3(256), 0, s>>>(p.ha);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

template <int NF, int PF>
static int hs_launch_mixed(const HsMixed& pr, int grid, size_t lds, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hs_mixed_kernel<NF, 8, PF, true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxHsLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_hs_mixed_kernel<NF, 8, PF, false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxHsLds));
    once.commit();
  }
  if (pr.b.log2_ncell >= 2) tconv_hs_mixed_kernel<NF, 8, PF, true><<<dim3(grid), dim3(512), lds, s>>>(pr);
  else tconv_hs_mixed_kernel<NF, 8, PF, false><<<dim3(grid), dim3(512), lds, s>>>(pr);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

#ifdef ADX_TCONV_TRACE
extern "C" int adx_debug_tconv_trace(unsigned long long* host_dst, int n) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_tconv_trace), sizeof(unsigned long long) * n);
}
#endif

// Two independent convolutions in one launch when both run on the short-K kernel with the same addressing variant;
// otherwise two launches.  Same results either way.
int tconv_hs_forward_pair(const adx_tconv_desc* da, const adx_tconv_io* ioa, const adx_tconv_desc* db,
                          const adx_tconv_io* iob, hipStream_t s) {
  constexpr bool pair_on = true;
  HsTile ta, tb;
  HsArgs ha, hb;
  HsdPair pr;
  size_t la = 0, lb = 0;
  int ga = 0, gb = 0;
  if (pair_on && tconv_hs_supported(da) && tconv_hs_supported(db) && hs_prepare(da, ioa, &ta, &ha) == ADX_OK &&
      hs_prepare(db, iob, &tb, &hb) == ADX_OK) {
    const bool a_short = hsd_prepare(da, ha, ta, &pr.a, &la, &ga), b_short = hsd_prepare(db, hb, tb, &pr.b, &lb, &gb);
    if (a_short && b_short && (pr.a.log2_ncell >= 2) == (pr.b.log2_ncell >= 2)) {
      pr.n_a = ga;
      return hsd_launch_pair(pr, ga + gb, la > lb ? la : lb, s);
    }
    // `a` on the K-split kernel (one launch: unsplit, or split with ticket words) beside a short-K `b`
    constexpr bool mixed_on = true;
    HsPlan p;
    if (mixed_on && !a_short && b_short && hs_plan(da, ioa, &p) == ADX_OK && !p.reduce) {
      HsMixed mx;
      mx.a = p.ha;
      mx.b = pr.b;
      mx.n_a = p.grid;
      const size_t lds = p.t.lds_bytes > lb ? p.t.lds_bytes : lb;
      return p.t.nf == 2 ? hs_launch_mixed<2, 4>(mx, p.grid + gb, lds, s) : hs_launch_mixed<1, 6>(mx, p.grid + gb, lds, s);
    }
  }
  int rc = tconv_forward(da, ioa, s);
  if (rc == ADX_OK) rc = tconv_forward(db, iob, s);
  return rc;
}

}  // namespace adx
