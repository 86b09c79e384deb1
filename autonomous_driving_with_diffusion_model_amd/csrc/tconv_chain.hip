// A whole run of temporal layers in ONE launch ("chain"): every layer of a UNet level whose channel count lets one
// workgroup hold ALL channels of its samples -- the two ResidualTemporalMapBlockConcat of the level, the Downsample1d /
// Upsample1d behind them and, on the last level, final_conv (modeling/temporal.py:46-55,118-194,219-244).
//
// Why: at the sizes of this model a temporal layer is a few hundred nanoseconds of matrix work behind ~4-6 us of launch
// boundary, kernel prologue, one global round trip to fetch activations another workgroup wrote a moment ago, and an
// epilogue (43 such launches per denoising step).  A convolution needs every input channel of its
// samples but nothing of other samples, so a workgroup that owns `bt` whole samples and all channels can run layer after
// layer with the activations never leaving its LDS: one launch, one prologue, no global hand-off -- at the price of
// every workgroup streaming every weight of the chain from L2 (164 KB per residual block at 64 channels, 655 KB at 128),
// which is why only the 64- and 128-channel levels are chained and the 256/512-channel levels keep one launch per layer
// with the weights split over workgroups (tconv_hs.hip).
//
// Arithmetic = tconv_hs.hip's: split-fp16 operands (x = hi + 2^-11 lo), three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulation; GroupNorm statistics by pairwise (Chan) merges of (mean, M2) in a fixed order; Mish with the hardware
// exp / rcp.  A stage is  conv (+ 1x1 residual conv as a second reduction) -> bias -> [GroupNorm -> Mish] -> + time
// bias -> + residual  and leaves its result (a) as fp32 in an LDS tile (the next block's identity residual, updated in
// place), (b) re-split into 16-byte cells of 8 channels (hi cell next to lo cell) = the next stage's A operand, (c) in
// global memory where a later launch needs it (the level's skip output, the chain's result).
#include <algorithm>
#include <vector>

#include "tconv_chain.h"

namespace adx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr float kChLoScale = 2048.0f;
constexpr float kChLoInv = 1.0f / 2048.0f;
constexpr int kChPF = 4;          // weight-fragment ring depth (K-steps in flight per wave)
constexpr int kChNT = 64 * kChainWaves;

template <int CTRL>
__device__ __forceinline__ float ch_dpp(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

__device__ __forceinline__ float ch_mish(float x) {
  if (x > 20.f) return x;
  const float e = __builtin_amdgcn_exp2f(x * 1.4426950408889634f);
  const float n = e * (e + 2.f);
  return x * n * __builtin_amdgcn_rcpf(n + 2.f);
}

__device__ __forceinline__ void ch_split8(const float (&v)[8], h8& hi, h8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const _Float16 h = (_Float16)v[j];
    hi[j] = h;
    lo[j] = (_Float16)((v[j] - (float)h) * kChLoScale);
  }
}

// (mean, M2) of two equally sized sets of n elements each -> of their union (Chan et al.); symmetric in its arguments,
// so both partners of an exchange compute the same bits
__device__ __forceinline__ void ch_merge(float& m, float& s, float mo, float so, float n_each) {
  const float d = mo - m;
  m = 0.5f * (m + mo);
  s = (s + so) + (d * d) * (0.5f * n_each);
}

// Chain input: global [B][C][L] (two sources = skip concat, arbitrary strides) -> split cells, rows = (sample, position)
__device__ __forceinline__ void ch_stage_input(const ChainArgs& ca, u32x4* cells, int b0, int tid) {
  const int lin = ca.in_len, cin = ca.in_c0 + ca.in_c1;
  const int ncell = ca.in_cpad >> 3, pitch = 2 * ncell + 1;
  const int rows = ca.bt * lin;
  const int cmax = cin - 1, bmax = ca.batch - 1;
  if (ca.in_vec) {
    const int nq = lin >> 2;
    const int items = ca.bt * nq * ncell;
    for (int it = tid; it < items; it += kChNT) {
      const int rq = it % (ca.bt * nq), oc = it / (ca.bt * nq);
      const int q = rq % nq, sb = rq / nq;
      const int b = b0 + sb, bc = min(b, bmax);
      f32x4 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = 8 * oc + j, cc = min(ci, cmax);
        const bool first = cc < ca.in_c0;
        const float* base = first ? ca.in0 : ca.in1;
        const int64_t off = first ? (int64_t)cc * ca.in0_sc + (int64_t)bc * ca.in0_sb
                                  : (int64_t)(cc - ca.in_c0) * ca.in1_sc + (int64_t)bc * ca.in1_sb;
        v[j] = *reinterpret_cast<const f32x4*>(base + off + 4 * q);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!(8 * oc + j < cin && b < ca.batch)) v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        float t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = v[j][p];
        h8 hi, lo;
        ch_split8(t8, hi, lo);
        u32x4* dst = cells + (sb * lin + 4 * q + p) * pitch + 2 * oc;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[1] = __builtin_bit_cast(u32x4, lo);
      }
    }
  } else {
    const int items = rows * ncell;
    for (int it = tid; it < items; it += kChNT) {
      const int row = it % rows, oc = it / rows;
      const int sb = row / lin, ip = row - sb * lin;
      const int b = b0 + sb, bc = min(b, bmax);
      float t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int ci = 8 * oc + j, cc = min(ci, cmax);
        const bool first = cc < ca.in_c0;
        const float* base = first ? ca.in0 : ca.in1;
        const int64_t off = first ? (int64_t)bc * ca.in0_sb + (int64_t)cc * ca.in0_sc + (int64_t)ip * ca.in0_sl
                                  : (int64_t)bc * ca.in1_sb + (int64_t)(cc - ca.in_c0) * ca.in1_sc + (int64_t)ip * ca.in1_sl;
        t8[j] = base[off];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (!(8 * oc + j < cin && b < ca.batch)) t8[j] = 0.f;
      h8 hi, lo;
      ch_split8(t8, hi, lo);
      u32x4* dst = cells + row * pitch + 2 * oc;
      dst[0] = __builtin_bit_cast(u32x4, hi);
      dst[1] = __builtin_bit_cast(u32x4, lo);
    }
  }
  for (int it = tid; it < pitch; it += kChNT) cells[rows * pitch + it] = u32x4{0u, 0u, 0u, 0u};   // the all-zero row
}

// One reduction (conv taps x input channels) of NR row tiles x one 16-channel tile.  `wp` = this lane's pointer into the
// tile's weight image [step][plane][64 lanes] x 16 bytes; `wq` = the fragment ring, holding steps 0 .. PF-1 on entry when
// `prefilled`.  Row r of tile i reads LDS row  rbase[i] + input position  (or the zero row).
template <int NR>
__device__ __forceinline__ void ch_gemm(const u32x4* __restrict__ cells, int pitch, int zrow, int kind, int taps, int stride,
                                        int pad, int lin, int log2_ncell, int nsteps, const u32x4* __restrict__ wp,
                                        u32x4 (&wq)[kChPF][2], bool prefilled, const int (&rbase)[4], const int (&rl)[4],
                                        const bool (&rok)[4], int kg, f32x4 (&accm)[4], f32x4 (&accx)[4]) {
  if (!prefilled) {
#pragma unroll
    for (int s = 0; s < kChPF; ++s) {
      const int st = min(s, nsteps - 1);
      wq[s][0] = wp[(size_t)st * 128];
      wq[s][1] = wp[(size_t)st * 128 + 64];
    }
  }
  const bool kind0 = kind == 0;
  const int ncm1 = (1 << log2_ncell) - 1;
  u32x4 ah[NR], al[NR];
  auto fetch = [&](int step) {
    const int kc = 4 * step + kg;                       // flattened (tap, 8-channel cell)
    const int tap = kc >> log2_ncell, cell = kc & ncm1;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const int vt = rl[i] + pad - tap;
      const int ip = kind0 ? rl[i] * stride + tap - pad : vt >> 1;
      const bool ok = rok[i] & ((unsigned)ip < (unsigned)lin) & (kind0 | ((vt & 1) == 0)) & (tap < taps);
      const u32x4* xp = cells + (ok ? rbase[i] + ip : zrow) * pitch + 2 * cell;
      ah[i] = xp[0];
      al[i] = xp[1];
    }
  };
  fetch(0);
  auto compute = [&](const u32x4 (&w)[2], int next_step) {
    h8 ch[NR], cl[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      ch[i] = __builtin_bit_cast(h8, ah[i]);
      cl[i] = __builtin_bit_cast(h8, al[i]);
    }
    fetch(next_step);
    __builtin_amdgcn_sched_barrier(0);      // the next step's LDS reads stay in front of this step's MFMAs
    const h8 wh = __builtin_bit_cast(h8, w[0]);
    const h8 wl = __builtin_bit_cast(h8, w[1]);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      accm[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[i], wh, accm[i], 0, 0, 0);
      accx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch[i], wl, accx[i], 0, 0, 0);
      accx[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl[i], wh, accx[i], 0, 0, 0);
    }
  };
  int j0 = 0;
  for (; j0 + kChPF <= nsteps; j0 += kChPF) {
#pragma unroll
    for (int s = 0; s < kChPF; ++s) {
      compute(wq[s], min(j0 + s + 1, nsteps - 1));
      const int st = min(j0 + s + kChPF, nsteps - 1);
      wq[s][0] = wp[(size_t)st * 128];
      wq[s][1] = wp[(size_t)st * 128 + 64];
    }
  }
#pragma unroll
  for (int s = 0; s < kChPF; ++s)
    if (j0 + s < nsteps) compute(wq[s], min(j0 + s + 1, nsteps - 1));
}

__global__ void __launch_bounds__(kChNT) tconv_chain_kernel(const ChainArgs ca) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const int b0 = blockIdx.x * ca.bt;
  const float* __restrict__ pk = ca.packed;

  ch_stage_input(ca, reinterpret_cast<u32x4*>(smem + ca.cell_off[ca.st[0].src]), b0, tid);
  __syncthreads();

  u32x4 wq[kChPF][2];
  for (int si = 0; si < ca.n_stages; ++si) {
    const ChainStage& st = ca.st[si];
    const int rows_out = ca.bt * st.lout;
    const int n_rt = (rows_out + 15) >> 4;
    const int n_ct = st.n_ct;
    // this wave's tiles: with fewer than 8 channel tiles the waves also split the row tiles
    int ct0, ct_step, rs, nsplit;
    if (n_ct >= kChainWaves) { ct0 = wave; ct_step = kChainWaves; rs = 0; nsplit = 1; }
    else { ct0 = wave & (n_ct - 1); ct_step = n_ct; rs = wave / n_ct; nsplit = kChainWaves / n_ct; }
    const int my_nr = rs < n_rt ? (n_rt - rs + nsplit - 1) / nsplit : 0;
    // per-lane row geometry of my row tiles, as A-operand rows (row = 16 rt + r16) ...
    int rbase[4], rl[4], rrb[4];
    bool rok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = 16 * (rs + i * nsplit) + r16;
      const int sb = m >> st.log2_lout;
      rok[i] = i < my_nr && m < rows_out;
      rl[i] = m & (st.lout - 1);
      rbase[i] = sb * st.lin;
      rrb[i] = m;                      // the 1x1 residual conv reads the block input at the output row itself
    }
    const u32x4* cells = reinterpret_cast<const u32x4*>(smem + ca.cell_off[st.src]);
    const int zrow = ca.bt * st.lin;
    float* F = smem + ca.f_off[st.f_dst];
    const int fp = st.cout_pad + 4;    // fp32 tile pitch

    for (int ct = ct0; ct < n_ct && my_nr > 0; ct += ct_step) {
      f32x4 accm[4], accx[4], racm[4], racx[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        accm[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i] = accm[i]; racm[i] = accm[i]; racx[i] = accm[i];
      }
      const u32x4* wp = reinterpret_cast<const u32x4*>(pk + st.w_off) + (size_t)ct * st.nsteps * 128 + lane;
      if (my_nr == 1) ch_gemm<1>(cells, st.src_pitch, zrow, st.kind, st.taps, st.stride, st.pad, st.lin, st.log2_ncell, st.nsteps, wp, wq, false, rbase, rl, rok, kg, accm, accx);
      else if (my_nr == 2) ch_gemm<2>(cells, st.src_pitch, zrow, st.kind, st.taps, st.stride, st.pad, st.lin, st.log2_ncell, st.nsteps, wp, wq, false, rbase, rl, rok, kg, accm, accx);
      else ch_gemm<4>(cells, st.src_pitch, zrow, st.kind, st.taps, st.stride, st.pad, st.lin, st.log2_ncell, st.nsteps, wp, wq, false, rbase, rl, rok, kg, accm, accx);
      if (st.r_src >= 0) {             // R(x): a 1x1 conv of the block input, same rows, its own accumulators
        const u32x4* rcells = reinterpret_cast<const u32x4*>(smem + ca.cell_off[st.r_src]);
        const u32x4* rwp = reinterpret_cast<const u32x4*>(pk + st.r_w_off) + (size_t)ct * st.r_nsteps * 128 + lane;
        const int rz = ca.bt * st.lout;
        int zb[4], zl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { zb[i] = rrb[i]; zl[i] = 0; }
        // taps = 1, stride 1, pad 0, lin = 1: input position 0 relative to rbase = the row itself
        if (my_nr == 1) ch_gemm<1>(rcells, st.r_pitch, rz, 0, 1, 1, 0, 1, st.r_log2_ncell, st.r_nsteps, rwp, wq, false, zb, zl, rok, kg, racm, racx);
        else if (my_nr == 2) ch_gemm<2>(rcells, st.r_pitch, rz, 0, 1, 1, 0, 1, st.r_log2_ncell, st.r_nsteps, rwp, wq, false, zb, zl, rok, kg, racm, racx);
        else ch_gemm<4>(rcells, st.r_pitch, rz, 0, 1, 1, 0, 1, st.r_log2_ncell, st.r_nsteps, rwp, wq, false, zb, zl, rok, kg, racm, racx);
      }
      // ---- epilogue of this channel tile, from the accumulators: lane = (channel r16, rows 4 kg .. 4 kg + 3 of a tile) ----
      const int c = 16 * ct + r16;
      const bool cok = c < st.cout;
      const float bias = (st.b_off >= 0 && cok) ? pk[st.b_off + c] : 0.f;
      float gm = 1.f, be = 0.f, rbias = 0.f;
      if (st.g_off >= 0 && cok) { gm = pk[st.g_off + c]; be = pk[st.be_off + c]; }
      if (st.r_src >= 0 && st.r_b_off >= 0 && cok) rbias = pk[st.r_b_off + c];
      f32x4 v[4];
      float gmean[4], gm2[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = accm[i] + accx[i] * kChLoInv;
        v[i] += bias;
      }
      if (st.g_off >= 0) {
        // GroupNorm statistics of (sample, group): 4 positions per lane -> channel lanes of the group -> row quads of
        // the sample inside the tile -> (lout = 32) the sample's second row tile through LDS
        const int cgl = st.cg_log2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float m = 0.25f * ((v[i][0] + v[i][1]) + (v[i][2] + v[i][3]));
          const f32x4 d = v[i] - m;
          float s = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
          float n = 4.f;
          if (cgl >= 1) { ch_merge(m, s, ch_dpp<0xB1>(m), ch_dpp<0xB1>(s), n); n *= 2.f; }
          if (cgl >= 2) { ch_merge(m, s, ch_dpp<0x4E>(m), ch_dpp<0x4E>(s), n); n *= 2.f; }
          if (cgl >= 3) { ch_merge(m, s, ch_dpp<0x141>(m), ch_dpp<0x141>(s), n); n *= 2.f; }
          if (cgl >= 4) { ch_merge(m, s, ch_dpp<0x140>(m), ch_dpp<0x140>(s), n); n *= 2.f; }
          if (st.lout >= 8) { ch_merge(m, s, __shfl_xor(m, 16, 64), __shfl_xor(s, 16, 64), n); n *= 2.f; }
          if (st.lout >= 16) { ch_merge(m, s, __shfl_xor(m, 32, 64), __shfl_xor(s, 32, 64), n); n *= 2.f; }
          gmean[i] = m;
          gm2[i] = s;
        }
        if (st.lout >= 32) {           // uniform: a sample spans two row tiles (which one wave may or may not both hold)
          float* xch = smem + ca.xch_off;            // [row tile][channel tile][group of the tile] x (mean, M2)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rt = rs + i * nsplit;
            if (i < my_nr && kg == 0 && (r16 & ((1 << cgl) - 1)) == 0) {
              float* p = xch + (((rt * n_ct + ct) << 2) + (r16 >> cgl)) * 2;
              p[0] = gmean[i];
              p[1] = gm2[i];
            }
          }
        }
      }
      // park what the second half of the epilogue needs across the (possible) barrier in registers: nothing else to do
      if (st.g_off >= 0 && st.lout >= 32) {
        // NOTE: every wave reaches this barrier the same number of times: the ct loop trip count is uniform over the
        // waves only when n_ct >= 8 divides evenly or n_ct < 8 (one trip); the host guarantees n_ct % 8 == 0 or n_ct < 8,
        // and waves without tiles (my_nr == 0) are sent through a matching barrier below
        __syncthreads();
        const float* xch = smem + ca.xch_off;
        const int cgl = st.cg_log2;
        const float n_half = (float)((16 << cgl) >> 0);          // elements of (16 positions x cg channels)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int rt = rs + i * nsplit;
          if (i < my_nr) {
            const int rt0 = rt & ~1;                              // the sample's first row tile
            const float* p0 = xch + (((rt0 * n_ct + ct) << 2) + (r16 >> cgl)) * 2;
            const float* p1 = xch + ((((rt0 + 1) * n_ct + ct) << 2) + (r16 >> cgl)) * 2;
            float m = p0[0], s = p0[1];
            ch_merge(m, s, p1[0], p1[1], n_half);
            gmean[i] = m;
            gm2[i] = s;
          }
        }
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i >= my_nr) continue;
        const int rt = rs + i * nsplit;
        const int m0 = 16 * rt + 4 * kg;
        const int sb = m0 >> st.log2_lout, l0 = m0 & (st.lout - 1);
        const int b = b0 + sb;
        const bool live = m0 < rows_out && b < ca.batch && cok;
        f32x4 o = v[i];
        if (st.g_off >= 0) {
          const float inv_n = 1.0f / (float)(st.lout << st.cg_log2);
          const float rstd = 1.0f / sqrtf(gm2[i] * inv_n + st.eps);
          const float sc = rstd * gm;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = ch_mish((v[i][k] - gmean[i]) * sc + be);
        }
        if (st.tb_col >= 0 && live) o += ca.tb[(int64_t)b * ca.tb_stride + st.tb_col + c];
        if (st.r_src >= 0) o += (racm[i] + racx[i] * kChLoInv) + rbias;
        float* fr = F + (size_t)m0 * fp + c;
        if (st.res_identity) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] += fr[(size_t)k * fp];
        }
        if (m0 < rows_out) {
#pragma unroll
          for (int k = 0; k < 4; ++k) fr[(size_t)k * fp] = o[k];
        }
        if (st.out >= 0 && live) {
          const ChainOut& go = ca.out[st.out];
          float* yp = go.p + (int64_t)b * go.sb + (int64_t)c * go.sc + (int64_t)l0 * go.sl;
          if (go.sl == 1 && go.vec) {
            *reinterpret_cast<f32x4*>(yp) = o;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) yp[(int64_t)k * go.sl] = o[k];
          }
        }
      }
    }
    if (my_nr == 0 && st.g_off >= 0 && st.lout >= 32) {
      const int trips = n_ct >= kChainWaves ? n_ct / kChainWaves : 1;
      for (int t = 0; t < trips; ++t) __syncthreads();            // keep the barrier count of the waves that have tiles
    }
    __syncthreads();
    // ---- fp32 tile -> split cells of the next stage's input (one (row, 8-channel cell) per thread and trip) ----------
    if (st.dst >= 0) {
      u32x4* dcells = reinterpret_cast<u32x4*>(smem + ca.cell_off[st.dst]);
      const int ncell = st.cout_pad >> 3;
      const int dp = st.dst_pitch;
      const int items = rows_out * ncell;
      for (int it = tid; it < items; it += kChNT) {
        const int cell = it % ncell, row = it / ncell;
        const float* fr = F + (size_t)row * fp + 8 * cell;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(fr);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(fr + 4);
        const float t8[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        h8 hi, lo;
        ch_split8(t8, hi, lo);
        u32x4* dst = dcells + row * dp + 2 * cell;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[1] = __builtin_bit_cast(u32x4, lo);
      }
      for (int it = tid; it < dp; it += kChNT) dcells[rows_out * dp + it] = u32x4{0u, 0u, 0u, 0u};
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side

// weight image of one chain layer: tconv_hs.hip's short-K layout [cout_pad16 / 16][nsteps][2 planes][64 lanes][8 halfs],
// without that kernel's cap on the number of steps
extern __global__ void tconv_hsd_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ packed, int layout, int flip,
                                             int taps, int cin, int cout, int ncell, int nsteps, size_t total);

static int ilog2_exact_ch(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

bool chain_layer_ok(const adx_tconv_desc* d) {
  if (!tconv_hs_supported(d)) return false;
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  if (ilog2_exact_ch(cin_pad / 8) < 0) return false;
  if (d->lout < 4 || d->lout > 32 || d->lin > 32) return false;
  const int n_ct = round_up(d->cout, 16) / 16;
  if (n_ct < kChainWaves ? ilog2_exact_ch(n_ct) < 0 : n_ct % kChainWaves != 0) return false;
  if (d->groups > 0) {
    const int cg = d->cout / d->groups;
    if (cg != 2 && cg != 4 && cg != 8 && cg != 16) return false;       // a group lies inside one 16-channel tile
  }
  return true;
}

size_t chain_packed_floats(const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  const int ns = ceil_div(d->taps * (cin_pad / 8), 4);
  return (size_t)(round_up(d->cout, 16) / 16) * ns * 256;               // 1024 bytes per (tile, step)
}

int chain_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  const int ncell = cin_pad / 8;
  const int ns = ceil_div(d->taps * ncell, 4);
  const size_t total = (size_t)(round_up(d->cout, 16) / 16) * ns * 512;
  tconv_hsd_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
      w, reinterpret_cast<_Float16*>(packed), d->kind == 1 ? 1 - d->w_layout : d->w_layout, d->w_flip, d->taps,
      d->c0 + d->c1, d->cout, ncell, ns, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

void chain_fill_stage(ChainStage* st, const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  st->kind = d->kind; st->taps = d->taps; st->stride = d->stride; st->pad = d->pad;
  st->log2_ncell = ilog2_exact_ch(cin_pad / 8);
  st->nsteps = ceil_div(d->taps * (cin_pad / 8), 4);
  st->lin = d->lin; st->lout = d->lout; st->log2_lout = ilog2_exact_ch(d->lout);
  st->cout = d->cout; st->cout_pad = round_up(d->cout, 16); st->n_ct = st->cout_pad / 16;
  st->cg_log2 = d->groups > 0 ? ilog2_exact_ch(d->cout / d->groups) : 0;
  st->eps = d->eps;
  st->src_pitch = 2 * (cin_pad / 8) + 1;
  st->dst_pitch = 2 * (st->cout_pad / 8) + 1;
}

int chain_launch(const ChainArgs& ca, int grid, size_t lds_bytes, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_chain_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainMaxLds));
    attr_set = true;
  }
  ADX_REQUIRE(lds_bytes <= kChainMaxLds, "tconv_chain: %zu bytes of LDS exceed %zu", lds_bytes, kChainMaxLds);
  tconv_chain_kernel<<<dim3(grid), dim3(kChNT), lds_bytes, s>>>(ca);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
