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
// What the time of such a kernel is made of (rocprofv3 SQ counters, round 3): at one or two waves per SIMD a wave retires one
// instruction per ~12 clocks whatever it is -- the per-layer kernels show the same rate -- and neither the weight stream nor
// the matrix work is visible (zero weights or no MFMAs: same time).  A stage's time IS its instruction count per wave, so
// everything here is written for few instructions on a wave's path: one 16 x 16 tile per wave and pass (8 waves share a
// stage's tiles), the tap of a K-step wave-uniform (inputs padded to 32 channels), the 1x1 residual conv as extra steps of
// the same weight stream, the stage table and every per-channel parameter in LDS before the first stage starts.
//
// Arithmetic = tconv_hs.hip's: split-fp16 operands (x = hi + 2^-11 lo), three v_mfma_f32_16x16x32_f16 per product, fp32
// accumulation; GroupNorm statistics by pairwise (Chan) merges of (mean, M2) in a fixed order; Mish with the hardware
// exp / rcp.  A stage is  conv (+ 1x1 residual conv as a second reduction) -> bias -> [GroupNorm -> Mish] -> + time
// bias -> + residual  and leaves its result (a) as fp32 in an LDS tile (the next block's identity residual, updated in
// place), (b) re-split into 16-byte cells of 8 channels (hi cell next to lo cell) = the next stage's A operand, (c) in
// global memory where a later launch needs it (the level's skip output, the chain's result).
#include <algorithm>
#include <cstddef>
#include <vector>

#include "tconv_chain.h"
#include "tconv_pack.h"

namespace adx {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// -DADX_CHAIN_TRACE: thread 0 of every workgroup stamps the shader clock at its phase boundaries (tools/chain_trace.py);
// every stamp costs a scalar-memory round trip, so a traced run is several times slower than a real one
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
#define ADX_CHAIN_PF 8
#endif
constexpr int kChPF = ADX_CHAIN_PF;       // weight-fragment ring depth: K-steps (2 KB each) in flight per wave
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

// (mean, M2) of two equally sized sets of n elements each -> of their union (Chan et al.), written symmetrically so that
// both partners of an exchange compute the same bits
__device__ __forceinline__ void ch_merge(float& m, float& s, float mo, float so, float n_each) {
  const float d = mo - m;
  m = 0.5f * (m + mo);
  s = (s + so) + (d * d) * (0.5f * n_each);
}

// Chain input: global [B][C][L] (two sources = skip concat, arbitrary strides) -> split cells, rows = (sample, position);
// channels beyond the real ones (inputs are padded to >= 32) are zero cells
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

// buffer descriptor of one 16-channel tile's weight image ([step][plane][64 lanes] x 16 bytes; the residual conv's steps
// behind the main conv's): uniform base, the lane supplies 16 * lane, the step is an SGPR offset; reads past the image
// return zeros without traffic (the ring simply runs off the end)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ch_wrsrc(const float* pk, int w_off, int nsteps, int ct) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(pk + w_off + (size_t)ct * nsteps * 512), 0, nsteps * 2048, 0x00020000);
}
__device__ __forceinline__ void ch_ring_fill(u32x4 (&wq)[kChPF][2], const __amdgpu_buffer_rsrc_t wrs, int lane16) {
#pragma unroll
  for (int s = 0; s < kChPF; ++s) {
    wq[s][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, s * 2048, 0);
    wq[s][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, s * 2048 + 1024, 0);
  }
}

// The argument block lives in the kernarg segment, i.e. in HOST memory on this platform: every 64-byte line of it a wave
// touches for the first time is a ~2 us round trip, and the compiler reads fields where they are used.  The stage table is
// therefore copied into LDS once (all lines in flight together); a stage's descriptor is then one LDS read per lane and
// one v_readlane per field.
constexpr int kChStageWords = (int)(sizeof(ChainStage) / 4);
static_assert(sizeof(ChainStage) % 4 == 0 && kChStageWords <= 64, "ChainStage must fit one dword per lane");

__device__ __forceinline__ ChainStage ch_stage_from_lds(const int* words, int lane) {
  const int v = words[min(lane, kChStageWords - 1)];
  int raw[kChStageWords];
#pragma unroll
  for (int k = 0; k < kChStageWords; ++k) raw[k] = __builtin_amdgcn_readlane(v, k);
  ChainStage t;
  __builtin_memcpy(&t, raw, sizeof(ChainStage));
  return t;
}

__global__ void __launch_bounds__(kChNT) tconv_chain_kernel(const ChainArgs ca) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r16 = lane & 15, kg = lane >> 4;
  const int lane16 = lane * 16;
  const int bt = ca.bt, batch = ca.batch, n_stages = ca.n_stages;
  const int b0 = blockIdx.x * bt;
  const float* __restrict__ pk = ca.packed;
  CH_STAMP(0);

  // first launch of a UNet forward: the split-reduction ticket words start every forward at zero whatever the caller's
  // workspace held (an uninitialised buffer, the debris of a launch that never finished); their users come later in stream order
  if (ca.zero_words != nullptr && blockIdx.x == 0 && tid < ca.n_zero)
    ca.zero_words[tid] = (ca.epoch_ctr != nullptr && tid == ca.epoch_slot)
                             ? __hip_atomic_fetch_add(ca.epoch_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u : 0u;
  int* ltab = reinterpret_cast<int*>(smem + ca.tab_lds);
#if defined(__HIP_DEVICE_COMPILE__)
  {
    typedef const __attribute__((address_space(4))) int* kernarg_words;
    kernarg_words kraw = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr() + (int)(offsetof(ChainArgs, st) / 4);
    for (int i = tid; i < n_stages * kChStageWords; i += kChNT) ltab[i] = kraw[i];
  }
#endif
  // the first weight fragments of this wave's first tile travel while everything else is staged
  u32x4 wq[kChPF][2];
  if (wave < ca.st[0].n_tiles)
    ch_ring_fill(wq, ch_wrsrc(pk, ca.st[0].w_off, ca.st[0].ns_main + ca.st[0].ns_r, wave >> ca.st[0].log2_nrt), lane16);
  {   // per-channel parameters of every stage (one block), this workgroup's rows of the time-bias matrix
    float* lpar = smem + ca.par_lds;
    for (int i = tid; i < ca.par_floats; i += kChNT) lpar[i] = pk[ca.par_src + i];
    for (int k = 0; k < ca.n_tb; ++k) {
      const int cp = ca.tb_cout[k], lg = 31 - __builtin_clz(cp);       // padded channel count: a power of two
      float* dst = smem + ca.tb_lds[k];
      for (int e = tid; e < bt * cp; e += kChNT) {
        const int sb = e >> lg, c = e & (cp - 1);
        dst[e] = ca.tb[(int64_t)min(b0 + sb, batch - 1) * ca.tb_stride + ca.tb_col[k] + c];
      }
    }
  }
  ch_stage_input(ca, reinterpret_cast<u32x4*>(smem + ca.in_cells), b0, tid);
  __syncthreads();
  CH_STAMP(1);

  for (int si = 0; si < n_stages; ++si) {
    const ChainStage st = ch_stage_from_lds(ltab + si * kChStageWords, lane);     // in scalar registers
    const int flags = st.flags;
    const int lout = 1 << st.log2_lout, n_ct = 1 << st.log2_nct, cp = 16 << st.log2_nct, fp = cp + 4;
    (void)n_ct;
    const int kind0 = (st.conv & 255) == 0, taps = (st.conv >> 8) & 255, stride = (st.conv >> 16) & 255, pad = st.conv >> 24;
    const u32x4* cells = reinterpret_cast<const u32x4*>(smem + st.src);
    const u32x4* rcells = reinterpret_cast<const u32x4*>(smem + st.r_src);
    float* F = smem + st.f_dst;
    const float* par = smem + st.par;
    const int nsteps = st.ns_main + st.ns_r;
    const int cgl = (flags >> 8) & 15;

    for (int t = wave; t < st.n_tiles; t += kChainWaves) {
      // row tiles fastest: the two halves of a 32-position sample are tiles t, t + 1 -- same trip of this loop, so their
      // GroupNorm partials meet at one barrier
      const int rt = t & ((1 << st.log2_nrt) - 1), ct = t >> st.log2_nrt;
      const int m = 16 * rt + r16;                       // this lane's row as an A operand
      const bool rok = m < st.rows_out;
      const int l = m & (lout - 1), rb = (m >> st.log2_lout) * st.lin;
      const __amdgpu_buffer_rsrc_t wrs = ch_wrsrc(pk, st.w_off, nsteps, ct);
      // ---- one weight stream: the conv's steps (tap-major; the four cells of a step lie in one tap, so the tap is wave-
      // uniform and the lane's LDS row changes only with it), then the 1x1 residual conv's steps on the block input --------
      f32x4 am = f32x4{0.f, 0.f, 0.f, 0.f}, ax = am, rm = am, rx = am;
      auto main_row = [&](int tap) {
        const int vt = l + pad - tap;
        const int ip = kind0 ? l * stride + tap - pad : vt >> 1;
        const bool ok = rok & ((unsigned)ip < (unsigned)st.lin) & (kind0 | ((vt & 1) == 0));
        return cells + (ok ? rb + ip : st.zrow) * st.src_pitch + 2 * kg;
      };
      const u32x4* r_rowp = rcells + ((rok && (flags & kChResConv)) ? m : st.r_zrow) * st.r_pitch + 2 * kg;   // (none: a zero row)
      const u32x4* rowp = main_row(0);
      int f_sub = 0, f_tap = 0, cur_spt = 1 << st.log2_spt;
      u32x4 ah, al;
      auto fetch = [&]() {
        const u32x4* xp = rowp + 8 * f_sub;              // 4 cells = 8 sixteen-byte units per step
        ah = xp[0];
        al = xp[1];
        if (++f_sub == cur_spt) {                        // uniform: next tap, or on to the residual conv's input
          f_sub = 0;
          if (++f_tap < taps) rowp = main_row(f_tap);
          else { rowp = r_rowp; cur_spt = 1 << 30; }
        }
      };
      fetch();
      auto compute = [&](const u32x4 (&w)[2], int j) {
        const h8 ch = __builtin_bit_cast(h8, ah);
        const h8 cl = __builtin_bit_cast(h8, al);
        fetch();                                         // (one step past the end reads a valid row: unused)
        __builtin_amdgcn_sched_barrier(0);               // the next step's LDS reads stay in front of this step's MFMAs
        const h8 wh = __builtin_bit_cast(h8, w[0]);
        const h8 wl = __builtin_bit_cast(h8, w[1]);
        if (j < st.ns_main) {
          am = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wh, am, 0, 0, 0);
          ax = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wl, ax, 0, 0, 0);
          ax = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, wh, ax, 0, 0, 0);
        } else {
          rm = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wh, rm, 0, 0, 0);
          rx = __builtin_amdgcn_mfma_f32_16x16x32_f16(ch, wl, rx, 0, 0, 0);
          rx = __builtin_amdgcn_mfma_f32_16x16x32_f16(cl, wh, rx, 0, 0, 0);
        }
      };
      int j0 = 0;
      for (; j0 + kChPF <= nsteps; j0 += kChPF) {
#pragma unroll
        for (int s = 0; s < kChPF; ++s) {
          compute(wq[s], j0 + s);
          const int so = (j0 + s + kChPF) * 2048;
          wq[s][0] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so, 0);
          wq[s][1] = __builtin_amdgcn_raw_buffer_load_b128(wrs, lane16, so + 1024, 0);
        }
      }
#pragma unroll
      for (int s = 0; s < kChPF; ++s)
        if (j0 + s < nsteps) compute(wq[s], j0 + s);
      CH_STAMP(2 + 4 * si);
      // the ring is empty: send for this wave's next tile -- of this stage, or of the next one -- so that the epilogue, the
      // barriers and the re-split run under that latency
      if (t + kChainWaves < st.n_tiles) {
        ch_ring_fill(wq, ch_wrsrc(pk, st.w_off, nsteps, (t + kChainWaves) >> st.log2_nrt), lane16);
      } else if (wave < st.nx_n_tiles) {
        ch_ring_fill(wq, ch_wrsrc(pk, st.nx_w_off, st.nx_nsteps, wave >> st.nx_log2_nrt), lane16);
      }
      // ---- epilogue from the accumulators: lane = (channel r16 of the tile, rows 4 kg .. 4 kg + 3) ---------------------------
      const int c = 16 * ct + r16;
      f32x4 v = am + ax * kChLoInv;
      v += par[c];
      const int m0 = 16 * rt + 4 * kg;
      const int sb = m0 >> st.log2_lout, l0 = m0 & (lout - 1);
      f32x4 o = v;
      if (flags & kChGn) {
        // statistics of (sample, group): 4 positions per lane -> the group's channel lanes -> the row quads of the sample
        // inside the tile -> (32 positions per sample) the sample's other row tile, which another wave holds, through LDS
        float mu = 0.25f * ((v[0] + v[1]) + (v[2] + v[3]));
        const f32x4 d = v - mu;
        float sq = (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        float n = 4.f;
        if (cgl >= 1) { ch_merge(mu, sq, ch_dpp<0xB1>(mu), ch_dpp<0xB1>(sq), n); n *= 2.f; }
        if (cgl >= 2) { ch_merge(mu, sq, ch_dpp<0x4E>(mu), ch_dpp<0x4E>(sq), n); n *= 2.f; }
        if (cgl >= 3) { ch_merge(mu, sq, ch_dpp<0x141>(mu), ch_dpp<0x141>(sq), n); n *= 2.f; }
        if (cgl >= 4) { ch_merge(mu, sq, ch_dpp<0x140>(mu), ch_dpp<0x140>(sq), n); n *= 2.f; }
        if (lout >= 8) { ch_merge(mu, sq, __shfl_xor(mu, 16, 64), __shfl_xor(sq, 16, 64), n); n *= 2.f; }
        if (lout >= 16) { ch_merge(mu, sq, __shfl_xor(mu, 32, 64), __shfl_xor(sq, 32, 64), n); n *= 2.f; }
        if (lout >= 32) {              // uniform; the host gives such a stage a multiple of 8 tiles: every wave takes the barrier
          float* xch = smem + ca.xch_lds;
          const int gi = (ct << 2) + (r16 >> cgl);
          if (kg == 0 && (r16 & ((1 << cgl) - 1)) == 0) {
            xch[(rt * (n_ct << 2) + gi) * 2] = mu;
            xch[(rt * (n_ct << 2) + gi) * 2 + 1] = sq;
          }
          __syncthreads();
          const float* p0 = xch + ((rt & ~1) * (n_ct << 2) + gi) * 2;
          const float* p1 = p0 + (n_ct << 3);
          mu = p0[0]; sq = p0[1];
          ch_merge(mu, sq, p1[0], p1[1], n);
        }
        const float sc = __builtin_amdgcn_rsqf(sq * st.inv_n + st.eps) * par[cp + c];
        const float be = par[2 * cp + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = ch_mish((v[k] - mu) * sc + be);
      }
      if (flags & kChTb) o += smem[st.tbl + sb * cp + c];
      if (flags & kChResConv) o += (rm + rx * kChLoInv) + par[3 * cp + c];
      float* fr = F + m0 * fp + c;
      if (flags & kChResIdentity) {
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] += fr[k * fp];
      }
      if (m0 < st.rows_out) {
#pragma unroll
        for (int k = 0; k < 4; ++k) fr[k * fp] = o[k];
        if ((flags & kChOut) && c < st.cout && b0 + sb < batch) {
          const ChainOut& go = ca.out[(flags >> 12) & 1];
          float* yp = go.p + (int64_t)(b0 + sb) * go.sb + (int64_t)c * go.sc + (int64_t)l0 * go.sl;
          if (flags & kChOutVec) {
            *reinterpret_cast<f32x4*>(yp) = o;
          } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) yp[(int64_t)k * go.sl] = o[k];
          }
        }
      }
    }
    CH_STAMP(3 + 4 * si);
    // a wave without tiles in this stage still fetches for its tile of the next one
    if (wave >= st.n_tiles && wave < st.nx_n_tiles)
      ch_ring_fill(wq, ch_wrsrc(pk, st.nx_w_off, st.nx_nsteps, wave >> st.nx_log2_nrt), lane16);
    __syncthreads();
    CH_STAMP(4 + 4 * si);
    // ---- fp32 tile -> split cells of the next stage's input: one (row, 8-channel cell) per thread and trip ---------------
    if (flags & kChCells) {
      u32x4* dcells = reinterpret_cast<u32x4*>(smem + st.dst);
      const int lg_nc = st.log2_nct + 1, dp = st.dst_pitch;
      for (int it = tid; it < (st.rows_out << lg_nc); it += kChNT) {
        const int cell = it & ((1 << lg_nc) - 1), row = it >> lg_nc;
        const float* fr = F + row * fp + 8 * cell;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(fr);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(fr + 4);
        const float t8[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        h8 hi, lo;
        ch_split8(t8, hi, lo);
        u32x4* dst = dcells + row * dp + 2 * cell;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[1] = __builtin_bit_cast(u32x4, lo);
      }
      if (tid < dp) dcells[st.rows_out * dp + tid] = u32x4{0u, 0u, 0u, 0u};
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
// ncell, ci = 8 (kc % ncell) + j], split into hi / lo planes; ncell = padded input channels / 8, a multiple of 4, so a step
// never straddles taps).  A stage with a 1x1 residual conv stores that conv's steps behind the main reduction's inside
// every tile (`tile_steps` = both, `step0` = where this conv's steps begin).
// (tconv_pack.hip, kPackCell)

static int ilog2_exact_ch(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

int chain_cin_pad(const adx_tconv_desc* d) { return std::max(32, round_up(d->c0 + d->c1, 16)); }
int chain_steps(const adx_tconv_desc* d) { return d->taps * (chain_cin_pad(d) / 8) / 4; }

bool chain_layer_ok(const adx_tconv_desc* d) {
  if (!tconv_hs_supported(d)) return false;
  if (ilog2_exact_ch(chain_cin_pad(d) / 8) < 0) return false;
  if (d->lout < 4 || d->lout > 32 || d->lin > 32) return false;
  if (ilog2_exact_ch(round_up(d->cout, 16) / 16) < 0) return false;
  if (d->groups > 0) {
    const int cg = d->cout / d->groups;
    if (cg != 2 && cg != 4 && cg != 8 && cg != 16) return false;       // a group lies inside one 16-channel tile
  }
  return true;
}

size_t chain_packed_floats(const adx_tconv_desc* d) {
  return (size_t)(round_up(d->cout, 16) / 16) * chain_steps(d) * 512;    // 2048 bytes per (tile, step)
}

// main conv (and, behind it in every tile, the 1x1 residual conv `r`, or null) -> one image
int chain_pack(const adx_tconv_desc* d, const float* w, const adx_tconv_desc* r, const float* rw, float* packed, hipStream_t s) {
  const int ns = chain_steps(d), nr = r != nullptr ? chain_steps(r) : 0;
  const int tiles = round_up(d->cout, 16) / 16;
  PackJob j{};
  j.w = w; j.out = packed; j.total = (uint32_t)((size_t)tiles * ns * 512); j.kind = kPackCell;
  j.layout = d->kind == 1 ? 1 - d->w_layout : d->w_layout; j.flip = d->w_flip; j.taps = d->taps; j.cin = d->c0 + d->c1;
  j.cout = d->cout; j.a = chain_cin_pad(d) / 8; j.b = ns; j.tile_steps = ns + nr; j.step0 = 0;
  int rc = pack_submit(j, s);
  if (rc == ADX_OK && r != nullptr) {
    PackJob q{};
    q.w = rw; q.out = packed; q.total = (uint32_t)((size_t)tiles * nr * 512); q.kind = kPackCell;
    q.layout = r->w_layout; q.flip = r->w_flip; q.taps = r->taps; q.cin = r->c0 + r->c1; q.cout = r->cout;
    q.a = chain_cin_pad(r) / 8; q.b = nr; q.tile_steps = ns + nr; q.step0 = ns;
    rc = pack_submit(q, s);
  }
  return rc;
}

int chain_launch(const ChainArgs& ca, int grid, size_t lds_bytes, hipStream_t s) {
  static std::atomic<uint64_t> attr_set{0};
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_chain_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kChainMaxLds));
    once.commit();
  }
  ADX_REQUIRE(lds_bytes <= kChainMaxLds, "tconv_chain: %zu bytes of LDS exceed %zu", lds_bytes, kChainMaxLds);
  tconv_chain_kernel<<<dim3(grid), dim3(kChNT), lds_bytes, s>>>(ca);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

}  // namespace adx
