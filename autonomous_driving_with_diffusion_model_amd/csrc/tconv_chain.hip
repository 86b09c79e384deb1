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
// STATUS (round 3): correct (3e-6 against the oracle through whole forwards) but NOT faster than the launches it replaces, so
// the executor leaves it off unless ADX_UNET_CHAIN=1.  Measured (rocprofv3, SQ counters, tools/chain_time.py): a chain takes
// 35-60 us against 29-42 us for its 5-7 launches.  Neither the weight stream nor the matrix work matters (zero weights or no
// MFMAs: same time); a wave retires one instruction per ~12 clocks at one or two waves per SIMD -- the per-layer kernels
// show the same rate -- and this kernel's stage costs ~1700 instructions per wave (descriptor, per-step addressing of a
// general reduction, pairwise statistics, re-split) where a whole per-layer launch costs ~1100 spread over 4x more waves.
// The budget a chain has to meet to win is ~450 instructions per stage and wave.
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

// -DADX_CHAIN_TRACE: thread 0 of every workgroup stamps the shader clock at its phase boundaries (tools/chain_trace.py)
#ifdef ADX_CHAIN_TRACE
__device__ unsigned long long g_chain_trace[64 * 256];
#define CH_STAMP(i)                                                                                        \
  do {                                                                                                     \
    if (threadIdx.x == 0 && blockIdx.x < 256 && (i) < 64) g_chain_trace[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define CH_STAMP(i) do { } while (0)
#endif

constexpr float kChLoScale = 2048.0f;
constexpr float kChLoInv = 1.0f / 2048.0f;
#ifndef ADX_CHAIN_PF
#define ADX_CHAIN_PF 10
#endif
constexpr int kChPF = ADX_CHAIN_PF;         // weight-fragment ring depth: K-steps (2 KB each) in flight per wave; the k5 layers have 10 n steps
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

// One reduction (conv taps x input channels) of NR row tiles x one 16-channel tile over steps step0 .. step0 + nsteps - 1
// of the tile's weight image [step][plane][64 lanes] x 16 bytes; the fragment ring `wq` holds steps
// 0 .. PF-1 ON ENTRY (the caller issued them earlier: the previous stage's epilogue ran under their latency) and keeps
// being refilled PF steps ahead through the tile's buffer descriptor `wrs` (loads past the image return zeros without
// traffic), so that a second reduction stored behind this one (the 1x1 residual conv, step0 = this one's step count)
// finds ITS first steps in the ring when nsteps is a multiple of PF.  Row r of tile i reads LDS row
// rbase[i] + input position (or the zero row).
template <int NR>
__device__ __forceinline__ void ch_gemm(const u32x4* __restrict__ cells, int pitch, int zrow, int kind, int taps, int stride,
                                        int pad, int lin, int log2_ncell, int nsteps, const __amdgpu_buffer_rsrc_t wrs, int step0,
                                        int rot, int lane16, u32x4 (&wq)[kChPF][2], const int (&rbase)[2], const int (&rl)[2],
                                        const bool (&rok)[2], int kg, f32x4 (&accm)[2], f32x4 (&accx)[2]) {
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
  // the reduction's steps are visited in the order rot, rot + 1, .., nsteps - 1, 0, .., rot - 1 (rot differs between
  // workgroups: they all stream the same weight image, and in lockstep they would all pull on the same L2 lines at once)
  auto phys = [&](int j) { const int p = j + rot; return p >= nsteps ? p - nsteps : p; };
  fetch(phys(0));
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
      compute(wq[s], phys(min(j0 + s + 1, nsteps - 1)));
      const int jn = j0 + s + kChPF;                        // logical index of the refill; past this reduction: what lies behind it
      const int so = (step0 + (jn < nsteps ? phys(jn) : jn)) * 2048;   // (past the tile's image the range check returns zeros, no traffic)
      wq[s][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so, 0);
      wq[s][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so + 1024, 0);
    }
  }
#pragma unroll
  for (int s = 0; s < kChPF; ++s)
    if (j0 + s < nsteps) compute(wq[s], phys(min(j0 + s + 1, nsteps - 1)));
}

// logical steps 0 .. PF-1 of a reduction of `nsteps` steps visited from `rot` (what lies behind the reduction is not rotated)
__device__ __forceinline__ void ch_ring_fill(u32x4 (&wq)[kChPF][2], const __amdgpu_buffer_rsrc_t wrs, int lane16, int nsteps, int rot) {
#pragma unroll
  for (int s = 0; s < kChPF; ++s) {
    int p = s;
    if (s < nsteps) { p = s + rot; if (p >= nsteps) p -= nsteps; }
    wq[s][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, p * 2048, 0);
    wq[s][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, p * 2048 + 1024, 0);
  }
}
__device__ __forceinline__ int ch_rot(int seed, int nsteps) { return seed % nsteps; }

// Which tiles of a stage a wave owns.  A wave multiplies up to TWO row tiles (tiles 2g, 2g + 1: with 32 positions per
// sample these are the two halves of one sample, so its GroupNorm statistics stay inside the wave) against one 16-channel
// tile per pass, so the weight fragments of a channel tile are fetched by as few waves as possible (the per-CU fill rate,
// ~64 B/clk, is what a chain's time is made of); waves beyond n_ct * ceil(n_rt / 2) idle in the K loops.
struct ChTiles {
  int ct0, ct_step, rt0, my_nr;
};
__device__ __forceinline__ ChTiles ch_tiles(const ChainStage& st, int bt, int wave) {
  ChTiles t;
  const int n_rt = (bt * st.lout + 15) >> 4;
  const int n_rg = (n_rt + 1) >> 1;                 // row groups of two tiles
  if (st.n_ct >= kChainWaves) {                     // n_rg == 1 (rows <= 32): every wave walks its channel tiles
    t.ct0 = wave; t.ct_step = kChainWaves; t.rt0 = 0;
    t.my_nr = min(n_rt, 2);
  } else {
    t.ct0 = wave & (st.n_ct - 1); t.ct_step = st.n_ct;
    const int rg = wave / st.n_ct;
    t.rt0 = 2 * rg;
    t.my_nr = rg < n_rg ? min(n_rt - 2 * rg, 2) : 0;
  }
  return t;
}
// buffer descriptor of one 16-channel tile's weight image ([step][plane][64 lanes] x 16 bytes; the residual conv's steps
// behind the main conv's): uniform base, the lane supplies 16 * lane, the step is an SGPR offset
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_wrsrc(const float* pk, const ChainStage& st, int ct) {
  const int tile_bytes = (st.nsteps + st.r_nsteps) * 2048;
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk + st.w_off + (size_t)ct * (tile_bytes / 4)), 0, tile_bytes, 0x00020000);
}

// The argument block lives in the kernarg segment, i.e. in HOST memory on this platform: every 64-byte line of it a wave
// touches for the first time is a ~2 us round trip, and the compiler reads fields where they are used -- a chain that walks
// its stage table that way pays that latency several times per stage (measured: 6 us of a 7 us stage).  So the block is
// copied into LDS once (all lines in flight together) and a stage's descriptor is then one LDS read per lane + readlanes.
constexpr int kChStageWords = (int)(sizeof(ChainStage) / 4);
static_assert(sizeof(ChainStage) % 4 == 0 && kChStageWords <= 64, "ChainStage must fit one dword per lane");
constexpr int kChOutWords = (int)(sizeof(ChainOut) / 4);
static_assert(sizeof(ChainOut) % 4 == 0 && kChOutWords <= 64, "ChainOut must fit one dword per lane");

template <typename T, int W>
__device__ __forceinline__ T ch_from_lds(const int* words, int lane) {
  const int v = words[min(lane, W - 1)];
  int raw[W];
#pragma unroll
  for (int k = 0; k < W; ++k) raw[k] = __builtin_amdgcn_readlane(v, k);
  T t;
  __builtin_memcpy(&t, raw, sizeof(T));
  return t;
}

__global__ void __launch_bounds__(kChNT) tconv_chain_kernel(const ChainArgs ca) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const int lane16 = lane * 16;
  const int b0 = blockIdx.x * ca.bt;
  const int rseed = ca.rotate ? (int)blockIdx.x * 3 + wave : 0;     // where this wave enters every reduction (see ch_gemm)
  const float* __restrict__ pk = ca.packed;

  CH_STAMP(0);
  int* largs = reinterpret_cast<int*>(smem + ca.args_off);
  {
#if defined(__HIP_DEVICE_COMPILE__)
    typedef const __attribute__((address_space(4))) int* kernarg_words;
    kernarg_words kraw = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr();     // ChainArgs is the only argument
    for (int i = tid; i < (int)(sizeof(ChainArgs) / 4); i += kChNT) largs[i] = kraw[i];
#endif
  }
  const int* lstages = largs + (int)(offsetof(ChainArgs, st) / 4);
  const int* louts = largs + (int)(offsetof(ChainArgs, out) / 4);
  const int n_stages = ca.n_stages, bt = ca.bt, batch = ca.batch;
  u32x4 wq[kChPF][2];
  {   // the first stage's first weight fragments travel while the input is staged
    const ChTiles t0 = ch_tiles(ca.st[0], ca.bt, wave);
    if (t0.my_nr > 0) ch_ring_fill(wq, ch_wrsrc(pk, ca.st[0], t0.ct0), lane16, ca.st[0].nsteps, ch_rot(rseed, ca.st[0].nsteps));
  }
  // every stage's per-channel parameters and this workgroup's time-bias rows -> LDS, once: the epilogues then issue no
  // global load at all (a load issued behind the weight ring returns behind it: in order).  One item per thread and
  // trip, all stages in one flat index space, so that the loads of a trip are in flight together.
  __syncthreads();            // the argument block is in LDS
  {
    const int per_stage = 4 * 128 + bt * 128;              // upper bound of a stage's items (cout_pad <= 128: host)
    const int o_cout = (int)(offsetof(ChainStage, cout) / 4), o_cp = (int)(offsetof(ChainStage, cout_pad) / 4);
    const int o_par = (int)(offsetof(ChainStage, par) / 4), o_b = (int)(offsetof(ChainStage, b_off) / 4);
    const int o_g = (int)(offsetof(ChainStage, g_off) / 4), o_be = (int)(offsetof(ChainStage, be_off) / 4);
    const int o_rs = (int)(offsetof(ChainStage, r_src) / 4), o_rb = (int)(offsetof(ChainStage, r_b_off) / 4);
    const int o_tb = (int)(offsetof(ChainStage, tb_col) / 4);
    for (int it = tid; it < n_stages * per_stage; it += kChNT) {
      const int k = it / per_stage, e = it - k * per_stage;
      const int* sw = lstages + k * kChStageWords;         // per-lane reads of the LDS copy (k differs between lanes)
      struct { int cout, cout_pad, par, b_off, g_off, be_off, r_src, r_b_off, tb_col; } sk =
          {sw[o_cout], sw[o_cp], sw[o_par], sw[o_b], sw[o_g], sw[o_be], sw[o_rs], sw[o_rb], sw[o_tb]};
      const int cp = sk.cout_pad;
      float* par = smem + sk.par;
      if (e < 4 * 128) {
        const int which = e >> 7, c = e & 127;
        if (c < cp) {
          const bool ok = c < sk.cout;
          float v = which == 1 ? 1.f : 0.f;
          if (which == 0 && ok && sk.b_off >= 0) v = pk[sk.b_off + c];
          if (which == 1 && ok && sk.g_off >= 0) v = pk[sk.g_off + c];
          if (which == 2 && ok && sk.g_off >= 0) v = pk[sk.be_off + c];
          if (which == 3 && ok && sk.r_src >= 0 && sk.r_b_off >= 0) v = pk[sk.r_b_off + c];
          par[which * cp + c] = v;
        }
      } else if (sk.tb_col >= 0) {
        const int sb = (e - 4 * 128) >> 7, c = e & 127;
        if (c < cp) {
          const int b = min(b0 + sb, batch - 1);
          par[(4 + sb) * cp + c] = c < sk.cout ? ca.tb[(int64_t)b * ca.tb_stride + sk.tb_col + c] : 0.f;
        }
      }
    }
  }
  ch_stage_input(ca, reinterpret_cast<u32x4*>(smem + ca.cell_off[ca.st[0].src]), b0, tid);
  __syncthreads();
  CH_STAMP(1);

  for (int si = 0; si < n_stages; ++si) {
    const ChainStage st = ch_from_lds<ChainStage, kChStageWords>(lstages + si * kChStageWords, lane);   // in scalar registers
    // of the next stage only what the weight prefetch needs
    ChainStage nx;
    {
      const int* nw = lstages + min(si + 1, n_stages - 1) * kChStageWords;
      nx.n_ct = __builtin_amdgcn_readfirstlane(nw[offsetof(ChainStage, n_ct) / 4]);
      nx.lout = __builtin_amdgcn_readfirstlane(nw[offsetof(ChainStage, lout) / 4]);
      nx.w_off = __builtin_amdgcn_readfirstlane(nw[offsetof(ChainStage, w_off) / 4]);
      nx.nsteps = __builtin_amdgcn_readfirstlane(nw[offsetof(ChainStage, nsteps) / 4]);
      nx.r_nsteps = __builtin_amdgcn_readfirstlane(nw[offsetof(ChainStage, r_nsteps) / 4]);
    }
    const int rows_out = bt * st.lout;
    const int n_ct = st.n_ct;
    const ChTiles T = ch_tiles(st, bt, wave);
    const int my_nr = T.my_nr;
    // per-lane row geometry of my row tiles, as A-operand rows (row = 16 rt + r16)
    int rbase[2], rl[2], rrow[2], rzero[2];
    bool rok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = 16 * (T.rt0 + i) + r16;
      rok[i] = i < my_nr && m < rows_out;
      rl[i] = m & (st.lout - 1);
      rbase[i] = (m >> st.log2_lout) * st.lin;
      rrow[i] = m;                     // the 1x1 residual conv reads the block input at the output row itself
      rzero[i] = 0;
    }
    const int* lcell = largs + (int)(offsetof(ChainArgs, cell_off) / 4);
    const u32x4* cells = reinterpret_cast<const u32x4*>(smem + lcell[st.src]);
    const int zrow = bt * st.lin;
    float* F = smem + largs[(int)(offsetof(ChainArgs, f_off) / 4) + st.f_dst];
    const int fp = st.cout_pad + 4;    // fp32 tile pitch
    const bool gn = st.g_off >= 0;
    const int rot = ch_rot(rseed, st.nsteps);
    const float* par = smem + st.par;            // this stage's parameters in LDS: bias | gamma | beta | residual bias | time bias

    for (int ct = T.ct0; ct < n_ct && my_nr > 0; ct += T.ct_step) {
      f32x4 v[2], rv[2];
      const __amdgpu_buffer_rsrc_t wrs = ch_wrsrc(pk, st, ct);
      {
        f32x4 accm[2], accx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { accm[i] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[i] = accm[i]; }
        if (my_nr == 1) ch_gemm<1>(cells, st.src_pitch, zrow, st.kind, st.taps, st.stride, st.pad, st.lin, st.log2_ncell, st.nsteps, wrs, 0, rot, lane16, wq, rbase, rl, rok, kg, accm, accx);
        else ch_gemm<2>(cells, st.src_pitch, zrow, st.kind, st.taps, st.stride, st.pad, st.lin, st.log2_ncell, st.nsteps, wrs, 0, rot, lane16, wq, rbase, rl, rok, kg, accm, accx);
#pragma unroll
        for (int i = 0; i < 2; ++i) v[i] = accm[i] + accx[i] * kChLoInv;
      }
      rv[0] = f32x4{0.f, 0.f, 0.f, 0.f}; rv[1] = rv[0];
      if (st.r_src >= 0) {
        f32x4 racm[2], racx[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) { racm[i] = f32x4{0.f, 0.f, 0.f, 0.f}; racx[i] = racm[i]; }             // R(x): a 1x1 conv of the block input, same rows, its own accumulators; its steps lie
        // behind the main reduction's in the image and (nsteps % PF == 0: checked on the host) already sit in the ring
        const u32x4* rcells = reinterpret_cast<const u32x4*>(smem + lcell[st.r_src]);
        const int rz = bt * st.lout;
        if (my_nr == 1) ch_gemm<1>(rcells, st.r_pitch, rz, 0, 1, 1, 0, 1, st.r_log2_ncell, st.r_nsteps, wrs, st.nsteps, 0, lane16, wq, rrow, rzero, rok, kg, racm, racx);
        else ch_gemm<2>(rcells, st.r_pitch, rz, 0, 1, 1, 0, 1, st.r_log2_ncell, st.r_nsteps, wrs, st.nsteps, 0, lane16, wq, rrow, rzero, rok, kg, racm, racx);
#pragma unroll
        for (int i = 0; i < 2; ++i) rv[i] = racm[i] + racx[i] * kChLoInv;
      }
      CH_STAMP(2 + 4 * si);
      // the ring is empty now: send for what this wave multiplies next -- its next channel tile of this stage, or its
      // first tile of the next stage -- so that the epilogue, the barriers and the re-split run under that latency
      if (ct + T.ct_step < n_ct) {
        ch_ring_fill(wq, ch_wrsrc(pk, st, ct + T.ct_step), lane16, st.nsteps, rot);
      } else if (si + 1 < n_stages) {
        const ChTiles tn = ch_tiles(nx, bt, wave);
        if (tn.my_nr > 0) ch_ring_fill(wq, ch_wrsrc(pk, nx, tn.ct0), lane16, nx.nsteps, ch_rot(rseed, nx.nsteps));
      }
      // ---- epilogue of this channel tile, from the accumulators: lane = (channel r16, rows 4 kg .. 4 kg + 3 of a tile) ----
      const int c = 16 * ct + r16;
      const bool cok = c < st.cout;
      const int cp = st.cout_pad;
      const float bias = par[c], gm = par[cp + c], be = par[2 * cp + c], rbias = par[3 * cp + c];   // zeros / ones where absent
      float gmean[2] = {0.f, 0.f}, gm2[2] = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] += bias;
      const int cgl = st.cg_log2;
      if (gn) {
        // GroupNorm statistics of (sample, group): 4 positions per lane -> channel lanes of the group -> row quads of
        // the sample inside the tile -> (lout = 32) the sample's second row tile through LDS
#pragma unroll
        for (int i = 0; i < 2; ++i) {
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
        if (st.lout >= 32 && my_nr == 2) {       // the wave's two tiles are the two halves of one sample
          float m = gmean[0], sq = gm2[0];
          ch_merge(m, sq, gmean[1], gm2[1], (float)(16 << cgl));
          gmean[0] = gmean[1] = m;
          gm2[0] = gm2[1] = sq;
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i >= my_nr) continue;
        const int m0 = 16 * (T.rt0 + i) + 4 * kg;
        const int sb = m0 >> st.log2_lout, l0 = m0 & (st.lout - 1);
        const int b = b0 + sb;
        const bool live = m0 < rows_out && b < batch && cok;
        f32x4 o = v[i];
        if (gn) {
          const float inv_n = 1.0f / (float)(st.lout << cgl);
          const float rstd = 1.0f / sqrtf(gm2[i] * inv_n + st.eps);
          const float sc = rstd * gm;
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] = ch_mish((v[i][k] - gmean[i]) * sc + be);
        }
        if (st.tb_col >= 0) o += par[(4 + sb) * cp + c];
        if (st.r_src >= 0) o += rv[i] + rbias;
        float* fr = F + (size_t)m0 * fp + c;
        if (st.res_identity) {
#pragma unroll
          for (int k = 0; k < 4; ++k) o[k] += fr[(size_t)k * fp];
        }
        if (m0 < rows_out) {
#pragma unroll
          for (int k = 0; k < 4; ++k) fr[(size_t)k * fp] = o[k];
        }
        if (st.out >= 0) {
          const ChainOut go = ch_from_lds<ChainOut, kChOutWords>(louts + st.out * kChOutWords, lane);
          if (live) {
          float* yp = go.p + (int64_t)b * go.sb + (int64_t)c * go.sc + (int64_t)l0 * go.sl;
          if (go.vec) {
            *reinterpret_cast<f32x4*>(yp) = o;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) yp[(int64_t)k * go.sl] = o[k];
          }
          }
        }
      }
    }
    CH_STAMP(3 + 4 * si);
    if (my_nr == 0 && si + 1 < n_stages) {         // a wave without tiles here may have some in the next stage
      const ChTiles tn = ch_tiles(nx, bt, wave);
      if (tn.my_nr > 0) ch_ring_fill(wq, ch_wrsrc(pk, nx, tn.ct0), lane16, nx.nsteps, ch_rot(rseed, nx.nsteps));
    }
    __syncthreads();
    CH_STAMP(4 + 4 * si);
    // ---- fp32 tile -> split cells of the next stage's input (one (row, 8-channel cell) per thread and trip) ----------
    if (st.dst >= 0) {
      u32x4* dcells = reinterpret_cast<u32x4*>(smem + lcell[st.dst]);
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
    CH_STAMP(5 + 4 * si);
  }
}

#ifdef ADX_CHAIN_TRACE
extern "C" int adx_debug_chain_trace(unsigned long long* host_dst, int n) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(g_chain_trace), sizeof(unsigned long long) * n);
}
extern "C" int adx_debug_chain_trace_clear() {
  static unsigned long long zeros[64 * 256];
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_chain_trace), zeros, sizeof(zeros));
}
#endif

// ---------------------------------------------------------------------------------------------------------------------
// host side

// Weight image of a chain stage: [cout_pad16 / 16][steps][2 planes][64 lanes][8 halfs] (tconv_hs.hip's short-K fragment
// order: element j of lane ln at `step` is W[n = 16 tile + (ln & 15)][flattened cell kc = 4 step + (ln >> 4): tap = kc /
// ncell, ci = 8 (kc % ncell) + j], split into hi / lo planes).  A stage with a 1x1 residual conv stores that conv's steps
// behind the main reduction's inside every tile (`tile_steps` = both, `step0` = where this conv's steps begin).
__global__ void chain_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ packed, int layout, int flip, int taps,
                                  int cin, int cout, int ncell, int nsteps, int tile_steps, int step0, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 7;
  const int ln = (idx >> 3) & 63;
  const size_t blk = idx >> 9;
  const int step = blk % nsteps;
  const int t16 = blk / nsteps;
  const int kc = 4 * step + (ln >> 4);
  const int tap = kc / ncell, ci = 8 * (kc - tap * ncell) + j;
  const int n = t16 * 16 + (ln & 15);
  float v = 0.f;
  if (tap < taps && n < cout && ci < cin) {
    const int ts = flip ? taps - 1 - tap : tap;
    v = layout == 0 ? w[((size_t)n * cin + ci) * taps + ts] : w[((size_t)ci * cout + n) * taps + ts];
  }
  const _Float16 hi = (_Float16)v;
  const _Float16 lo = (_Float16)((v - (float)hi) * kChLoScale);
  _Float16* dst = packed + ((size_t)t16 * tile_steps + step0 + step) * 1024 + ln * 8 + j;
  dst[0] = hi;
  dst[512] = lo;
}

static int ilog2_exact_ch(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

int chain_steps(const adx_tconv_desc* d) { return ceil_div(d->taps * (round_up(d->c0 + d->c1, 16) / 8), 4); }

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

bool chain_residual_ok(const adx_tconv_desc* main) { return chain_steps(main) % kChPF == 0; }

size_t chain_packed_floats(const adx_tconv_desc* d) {
  return (size_t)(round_up(d->cout, 16) / 16) * chain_steps(d) * 512;    // 2048 bytes per (tile, step)
}

// main conv (and, behind it in every tile, the 1x1 residual conv `r`, or null) -> one image
int chain_pack(const adx_tconv_desc* d, const float* w, const adx_tconv_desc* r, const float* rw, float* packed, hipStream_t s) {
  const int ns = chain_steps(d), nr = r != nullptr ? chain_steps(r) : 0;
  const int tiles = round_up(d->cout, 16) / 16;
  {
    const size_t total = (size_t)tiles * ns * 512;
    chain_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        w, reinterpret_cast<_Float16*>(packed), d->kind == 1 ? 1 - d->w_layout : d->w_layout, d->w_flip, d->taps,
        d->c0 + d->c1, d->cout, round_up(d->c0 + d->c1, 16) / 8, ns, ns + nr, 0, total);
  }
  if (r != nullptr) {
    const size_t total = (size_t)tiles * nr * 512;
    chain_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(
        rw, reinterpret_cast<_Float16*>(packed), r->w_layout, r->w_flip, r->taps, r->c0 + r->c1, r->cout,
        round_up(r->c0 + r->c1, 16) / 8, nr, ns + nr, ns, total);
  }
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

void chain_fill_stage(ChainStage* st, const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  st->kind = d->kind; st->taps = d->taps; st->stride = d->stride; st->pad = d->pad;
  st->log2_ncell = ilog2_exact_ch(cin_pad / 8);
  st->nsteps = chain_steps(d);
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
