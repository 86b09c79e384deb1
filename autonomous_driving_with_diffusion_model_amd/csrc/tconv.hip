// Temporal (1-D) convolution as an implicit GEMM on the fp32 matrix cores of gfx950.
//
// Replaces, fused into ONE launch each (reference = one torch op per arrow):
//   Conv1dBlock        Conv1d(k5) -> GroupNorm(8) -> Mish            modeling/helpers.py:95-112
//     + time-bias add  out = block0(x) + time_mlp(t)[:, :, None]     modeling/temporal.py:53
//     + residual add   out = block1(.) + residual_conv(x)            modeling/temporal.py:55
//   Downsample1d       Conv1d(C, C, 3, 2, 1)                         modeling/helpers.py:77-83
//   Upsample1d         ConvTranspose1d(C, C, 4, 2, 1)                modeling/helpers.py:86-92
//   residual / head    Conv1d(Cin, Cout, 1)                          modeling/temporal.py:40-44,192-194
//   block time_mlp     Linear(2*dim, C) as a length-1 "conv"         modeling/temporal.py:34-38
//
// GEMM view: rows m = (sample, position), cols n = output channel, k = (tap, input channel).
// One workgroup = 4 waves owns `bt` whole samples x `ct` channels, where ct is a multiple of
// the GroupNorm group width, so every (sample, group) it touches is complete on chip:
//   1. the input tile [cin][bt][lin + halo] is staged zero-padded in LDS (coalesced along
//      the horizon axis, two sources = skip concat without materialising the cat);
//   2. the 4 waves split K; A fragments come from LDS (ds_read_b32, conflict-free pitch
//      chosen by the host), B fragments straight from the pre-packed weight image with one
//      16-byte load per lane per 16 input channels (v_mfma_f32_16x16x4_f32, exact fp32);
//   3. the 4 partial tiles are summed through LDS in the [sample][channel][pos] order of
//      the output tensor, one wave per (sample, group) computes two-pass mean/variance with
//      wavefront shuffles, applies affine + Mish + time bias + residual and stores coalesced.
#include <algorithm>
#include <array>
#include <map>
#include <mutex>

#include "tconv_internal.h"
#include "tconv_pack.h"

namespace adx {

__device__ __forceinline__ int tconv_in_pos(int kind, int l, int tap, int stride, int pad, bool& ok) {
  if (kind == 0) {
    ok = true;
    return l * stride + tap - pad;
  }
  // ConvTranspose1d, stride 2: o = 2 i - pad + tap  <=>  i = (o + pad - tap) / 2 when even
  const int v = l + pad - tap;
  ok = (v & 1) == 0;
  return v >> 1;
}

template <int NW>
struct KCursor {  // walks this wave's (tap, 16-channel block) pairs of one chunk: i = wave, wave + NW, ...
  int tap, cb;
  __device__ __forceinline__ void init(int wave, int ncbc) {
    tap = 0;
    cb = wave;
    while (cb >= ncbc) { cb -= ncbc; ++tap; }
  }
  __device__ __forceinline__ void next(int ncbc) {
    cb += NW;
    while (cb >= ncbc) { cb -= ncbc; ++tap; }
  }
};

// PF = depth of the register ring that keeps the packed-weight loads of the next PF K-blocks in
// flight while the current block's MFMAs run (the weights come from L2 / Infinity Cache).
// NW = waves per workgroup = K-split factor (4 for shallow layers, 16 where taps*cin/16 is large, so
// that a CU holds 4 waves per SIMD and the MFMA pipe stays fed while other waves wait on loads).
template <int MF, int NF, int PF, int NW>
__global__ void __launch_bounds__(64 * NW) tconv_kernel(const TConvArgs a) {
  constexpr int NT = 64 * NW;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nt = blockIdx.x % a.ntiles;  // same channel tile => same weights on blocks b, b+ntiles: stays in one XCD's L2 when ntiles | 8 or 8 | ntiles
  const int b0 = (blockIdx.x / a.ntiles) * a.bt;
  const int r = lane & 15, kk = lane >> 4;
  const int batch = a.io.batch;

  int rowoff[MF], rowl[MF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) {
    const int m = mf * 16 + r;
    const int bl = m >> a.log2_lout;
    rowl[mf] = m & (a.lout - 1);
    rowoff[mf] = bl * a.lp + a.pl;
  }
  f32x4 acc[MF][NF];
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = f32x4{0.f, 0.f, 0.f, 0.f};

  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.io.packed_w) + (size_t)nt * NF * a.nkb * 64 + lane;
  const int nq = a.rs >> 2;      // 16-byte quads per staged row (pl, lp, rs are multiples of 4)
  const int lpq = max(a.lp >> 2, 1), plq = a.pl >> 2, linq = a.lin >> 2;  // used on the dense path only
  const bool dense = a.dense != 0;
  f32x4 bq[PF][NF];

  for (int c0 = 0; c0 < a.cin_pad; c0 += a.ck) {
    const int ckc = min(a.ck, a.cin_pad - c0);
    const int ncbc = ckc >> 4;
    const int nblk = a.taps * ncbc;
    const int nbw = nblk > wave ? (nblk - wave + NW - 1) / NW : 0;  // K-blocks of this wave in this chunk
    const int cb0 = c0 >> 4;
    // ---- weight prefetch for the first PF blocks; in flight while the input tile is staged ----
    KCursor<NW> lc;
    lc.init(wave, ncbc);
#pragma unroll
    for (int s = 0; s < PF; ++s) {
      const int tp = min(lc.tap, a.taps - 1);  // past-the-end slots re-read a valid block and are never used
      const f32x4* src = wp + (size_t)(tp * a.ncb + cb0 + lc.cb) * 64;
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bq[s][nf] = src[(size_t)nf * a.nkb * 64];
      lc.next(ncbc);
    }
    if (c0 > 0) __syncthreads();
    // ---- stage [ckc channels][rs columns], zero padded, one 16-byte quad per work item ---------
    // every quad of a row is either all halo/pad (zeros) or 4 consecutive positions of one sample
    {
      const int items = ckc * nq;
#pragma unroll 4
      for (int it = tid; it < items; it += NT) {
        const int cl = it / nq, qi = it - cl * nq;
        const int ci = c0 + cl;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (ci < a.cin) {
          const bool first = ci < a.c0;
          const float* src = first ? a.io.x0 + (int64_t)ci * a.io.x0_sc : a.io.x1 + (int64_t)(ci - a.c0) * a.io.x1_sc;
          const int64_t sb = first ? a.io.x0_sb : a.io.x1_sb;
          if (dense) {
            const int bl = qi / lpq, ql = qi - bl * lpq - plq;
            const int b = b0 + bl;
            if (bl < a.bt && ql >= 0 && ql < linq && b < batch)
              v = *reinterpret_cast<const f32x4*>(src + (int64_t)b * sb + 4 * ql);
          } else {
            const int64_t sl = first ? a.io.x0_sl : a.io.x1_sl;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int col = 4 * qi + k;
              const int bl = col / a.lp, ip = col - bl * a.lp - a.pl;
              const int b = b0 + bl;
              if (bl < a.bt && ip >= 0 && ip < a.lin_valid && b < batch) v[k] = src[(int64_t)b * sb + (int64_t)ip * sl];
            }
          }
        }
        *reinterpret_cast<f32x4*>(smem + 4 * it) = v;
      }
    }
    __syncthreads();
    // ---- K loop over this wave's blocks, PF-deep weight ring -----------------------------------
    KCursor<NW> cc;
    cc.init(wave, ncbc);
    auto compute = [&](const f32x4 (&bw)[NF], int tap, int cbl) {
      float av[MF][4];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        bool ok;
        const int ip = tconv_in_pos(a.kind, rowl[mf], tap, a.stride, a.pad, ok);
        const int col = ok ? rowoff[mf] + ip : 0;  // column 0 is always a zero pad
        const float* xp = smem + (cbl * 16 + kk) * a.rs + col;
#pragma unroll
        for (int j = 0; j < 4; ++j) av[mf][j] = xp[4 * j * a.rs];
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NF; ++nf)
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mf][j], bw[nf][j], acc[mf][nf], 0, 0, 0);
    };
    int j0 = 0;
    for (; j0 + PF <= nbw; j0 += PF) {  // full groups: branch-free
#pragma unroll
      for (int s = 0; s < PF; ++s) {
        compute(bq[s], cc.tap, cc.cb);
        cc.next(ncbc);
        const int tp = min(lc.tap, a.taps - 1);
        const f32x4* src = wp + (size_t)(tp * a.ncb + cb0 + lc.cb) * 64;
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) bq[s][nf] = src[(size_t)nf * a.nkb * 64];
        lc.next(ncbc);
      }
    }
#pragma unroll
    for (int s = 0; s < PF; ++s) {      // tail group: its weights are already in the ring
      if (j0 + s < nbw) {
        compute(bq[s], cc.tap, cc.cb);
        cc.next(ncbc);
      }
    }
  }

  // ---- epilogue: sum the K-partials through LDS, laid out [sample][channel][pos] ----------
  __syncthreads();
  constexpr int tile_elems = 256 * MF * NF;  // bt * ct * lout
  float* P = smem;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int c = nf * 16 + r;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int m = mf * 16 + kk * 4 + q;
        const int bl = m >> a.log2_lout, l = m & (a.lout - 1);
        P[wave * tile_elems + (((bl << a.log2_ct) + c) << a.log2_lout) + l] = acc[mf][nf][q];
      }
    }
  __syncthreads();
  tconv_epilogue<NT, NW, tile_elems>(a, smem, tid, lane, wave, nt, b0);
}

// weight image: [cout_pad/16][nkb][64 lanes][4]; element j of lane l in block (tap, cb) is
// W[n = 16*tile + (l & 15)][ci = 16*cb + 4*j + (l >> 4)][tap]  (B operand of 16x16x4, k = l >> 4): tconv_pack.hip, kPackExact

constexpr size_t kMaxTconvLds = 132 * 1024;

static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

int tconv_check(const adx_tconv_desc* d) {
  ADX_REQUIRE(d != nullptr, "tconv: null descriptor");
  ADX_REQUIRE(d->kind == 0 || d->kind == 1, "tconv: kind must be 0 (conv) or 1 (transposed), got %d", d->kind);
  ADX_REQUIRE(d->taps >= 1 && d->taps <= 8 && d->c0 >= 1 && d->c1 >= 0 && d->cout >= 1, "tconv: bad channel/tap count");
  ADX_REQUIRE(d->lin >= 1 && d->lout >= 1, "tconv: bad length");
  if (d->kind == 0) {
    ADX_REQUIRE(d->stride >= 1, "tconv: stride must be >= 1");
    ADX_REQUIRE(d->lout == (d->lin + 2 * d->pad - d->taps) / d->stride + 1, "tconv: lout %d inconsistent with lin %d",
                d->lout, d->lin);
  } else {
    ADX_REQUIRE(d->stride == 2, "tconv: transposed conv supports stride 2 only");
    const int lo0 = (d->lin - 1) * 2 - 2 * d->pad + d->taps;  // output_padding 0 or 1
    ADX_REQUIRE(d->lout == lo0 || d->lout == lo0 + 1, "tconv: transposed lout %d inconsistent", d->lout);
    ADX_REQUIRE(d->w_layout == 0 || d->w_layout == 1, "tconv: bad w_layout");
    ADX_REQUIRE((d->taps - 1 - d->pad + 1) / 2 >= 1, "tconv: transposed conv needs a left halo");
  }
  ADX_REQUIRE(ilog2_exact(d->lout) >= 0 && d->lout <= 64,
              "tconv: lout must be a power of two <= 64, got %d (other lengths: round lin / lout up and give the real ones in "
              "lin_valid / lout_valid)", d->lout);
  ADX_REQUIRE(d->lin_valid >= 0 && d->lin_valid <= d->lin && d->lout_valid >= 0 && d->lout_valid <= d->lout,
              "tconv: lin_valid %d / lout_valid %d outside [0, lin %d] / [0, lout %d]", d->lin_valid, d->lout_valid, d->lin,
              d->lout);
  if (d->groups > 0) ADX_REQUIRE(d->cout % d->groups == 0, "tconv: cout %d not divisible by groups %d", d->cout, d->groups);
  ADX_REQUIRE(tconv_hs_supported(d) || tconv_exact_supported(d) || tconv_generic_supported(d),
              "tconv: no kernel covers this layer (the general-shape kernel needs a sample's %d x %d input in 128 KB of LDS)",
              d->c0 + d->c1, d->lin);
  return ADX_OK;
}

// the tile rules of the exact-fp32 MFMA kernel below (layers outside them run on tconv_generic.hip)
bool tconv_exact_supported(const adx_tconv_desc* d) {
  if (d->groups > 0) {
    if (d->cout % d->groups != 0) return false;
    const int cg = d->cout / d->groups;
    if (ilog2_exact(cg) < 0 || cg > 128 || d->cout % 16 != 0 || (cg * d->lout) % 64 != 0) return false;
  }
  return true;
}

// LDS pitch search: the A fragment of one MFMA is read by lanes (r = row 0..15, kk = 0..3) at
// word kk*rs + sample*lp + pos; ds_read_b32 serves lanes 0-31 (kk = 0,1) in one pass when all
// 32 words fall in distinct banks (word % 32).
static int lds_conflict_cost(const adx_tconv_desc* d, int bt, int mf, int pl, int lp, int rs) {
  int cost = 0;
  for (int tap = 0; tap < d->taps; ++tap)
    for (int f = 0; f < mf; ++f) {
      int words[32], nw = 0;
      for (int kk = 0; kk < 2; ++kk)
        for (int r = 0; r < 16; ++r) {
          const int m = f * 16 + r;
          const int bl = m / d->lout, l = m % d->lout;
          int ip;
          bool ok = true;
          if (d->kind == 0) {
            ip = l * d->stride + tap - d->pad;
          } else {
            const int v = l + d->pad - tap;
            ok = (v & 1) == 0;
            ip = v >> 1;
          }
          words[nw++] = kk * rs + (ok ? bl * lp + pl + ip : 0);
        }
      int worst = 1;
      for (int bank = 0; bank < 32; ++bank) {
        int distinct = 0;
        for (int i = 0; i < nw; ++i) {
          if (words[i] % 32 != bank) continue;
          bool seen = false;
          for (int j = 0; j < i; ++j) seen |= (words[j] == words[i]);
          distinct += !seen;
        }
        worst = distinct > worst ? distinct : worst;
      }
      cost += worst;
    }
  (void)bt;
  return cost;
}

int tconv_tile(const adx_tconv_desc* d, int batch, TConvTile* t) {
  int rc = tconv_check(d);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(batch >= 1, "tconv: batch must be >= 1");
  ADX_REQUIRE(tconv_exact_supported(d), "tconv: the exact-fp32 MFMA kernel does not tile this layer");
  t->cin = d->c0 + d->c1;
  t->cin_pad = round_up(t->cin, 16);
  t->ncb = t->cin_pad / 16;
  t->nkb = d->taps * t->ncb;
  t->cout_pad = round_up(d->cout, 16);
  if (d->lout >= 16) {
    t->bt = 1;
    t->mf = d->lout / 16;
  } else {
    t->bt = 16 / d->lout;
    t->mf = 1;
  }
  const int btiles = ceil_div(batch, t->bt);
  const int n16 = t->cout_pad / 16;
  if (d->groups > 0) {
    const int cg = d->cout / d->groups;
    t->ct = cg > 16 ? cg : 16;
    t->nf = t->ct / 16;
  } else {
    t->nf = 1;
    for (int nf = 4; nf >= 2; nf >>= 1)
      if (n16 % nf == 0 && t->mf * nf <= 8 && btiles * (n16 / nf) >= 512) {
        t->nf = nf;
        break;
      }
    t->ct = 16 * t->nf;
  }
  ADX_REQUIRE(t->mf * t->nf <= 8 && t->mf <= 4 && t->nf <= 8, "tconv: tile %dx%d fragments unsupported", t->mf, t->nf);
  ADX_REQUIRE(n16 % t->nf == 0, "tconv: cout %d not divisible by the channel tile %d", d->cout, t->ct);
  t->ntiles = n16 / t->nf;
  int pr;
  if (d->kind == 0) {
    t->pl = d->pad;
    pr = (d->lout - 1) * d->stride + d->taps - 1 - d->pad - (d->lin - 1);
  } else {
    t->pl = (d->taps - 1 - d->pad + 1) / 2;
    pr = (d->lout - 1 + d->pad) / 2 - (d->lin - 1);
  }
  if (pr < 0) pr = 0;
  // 16-byte staging: when lin is a multiple of 4 keep every sample's data on quad boundaries
  const bool quads = d->lin % 4 == 0;
  const int step = quads ? 4 : 1;
  if (quads) t->pl = round_up(t->pl, 4);
  const int lp_min = round_up(t->pl + d->lin + pr, step);
  // the pitch search costs ~1 ms of host time: memoise it per geometry (thread-safe)
  {
    const std::array<int, 8> key{d->kind, d->taps, d->stride, d->pad, d->lin, d->lout, t->bt, t->mf};
    static std::mutex mu;
    static std::map<std::array<int, 8>, std::pair<int, int>> memo;
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(key);
    if (it == memo.end()) {
      int best_cost = 1 << 30, best_lp = lp_min, best_rs = round_up(t->bt * lp_min, 4);
      for (int lp = lp_min; lp <= lp_min + 16; lp += step)
        for (int rs = round_up(t->bt * lp, 4); rs <= t->bt * lp + 36; rs += 4) {
          const int cost = lds_conflict_cost(d, t->bt, t->mf, t->pl, lp, rs);
          if (cost < best_cost || (cost == best_cost && rs < best_rs)) {
            best_cost = cost;
            best_lp = lp;
            best_rs = rs;
          }
        }
      it = memo.emplace(key, std::make_pair(best_lp, best_rs)).first;
    }
    t->lp = it->second.first;
    t->rs = it->second.second;
  }
  // Staged tile: <= 64 KB when there are enough workgroups to co-schedule two per CU, up to 128 KB
  // when the whole grid is a single wave of workgroups anyway (then one chunk = one staging phase);
  // the channels are split into equal chunks beyond that.
  const bool single_wave = (long)btiles * t->ntiles <= 320;
  int ck_max = ((single_wave ? 32768 : 16384) / t->rs) / 16 * 16;
  if (ck_max < 16) ck_max = 16;
  const int nchunks = ceil_div(t->cin_pad, ck_max);
  t->ck = round_up(ceil_div(t->cin_pad, nchunks), 16);
  const size_t stage = (size_t)t->ck * t->rs;
  t->nw = t->nkb >= 64 ? 16 : (t->nkb >= 32 ? 8 : 4);   // waves per workgroup = K split
  const size_t epi = (size_t)t->nw * 256 * t->mf * t->nf + 2 * 4 * t->mf * t->nf;  // K-partials + GN wave partials
  t->lds_bytes = sizeof(float) * (stage > epi ? stage : epi);
  ADX_REQUIRE(t->lds_bytes <= kMaxTconvLds, "tconv: LDS tile of %zu bytes exceeds %zu", t->lds_bytes, kMaxTconvLds);
  return ADX_OK;
}

size_t tconv_packed_floats(const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  const size_t exact = (size_t)(round_up(d->cout, 16) / 16) * d->taps * (cin_pad / 16) * 256;
  const size_t hs = tconv_hs_packed_floats(d);     // both images fit: the `exact` flag may flip between pack calls
  const size_t gen = tconv_generic_packed_floats(d);
  return std::max(std::max(exact, hs), gen);
}

int tconv_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s) {
  int rc = tconv_check(d);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(w != nullptr && packed != nullptr, "tconv_pack: null pointer");
  if (tconv_hs_supported(d)) return tconv_hs_pack(d, w, packed, s);
  if (!tconv_exact_supported(d)) return tconv_generic_pack(d, w, packed, s);
  const int cin = d->c0 + d->c1;
  const int ncb = round_up(cin, 16) / 16;
  const size_t total = tconv_packed_floats(d);
  PackJob j{};
  j.w = w; j.out = packed; j.total = (uint32_t)total; j.kind = kPackExact;
  j.layout = d->kind == 1 ? 1 - d->w_layout : d->w_layout; j.flip = d->w_flip; j.taps = d->taps; j.cin = cin; j.cout = d->cout;
  j.a = ncb; j.b = d->taps * ncb;
  return pack_submit(j, s);
}

template <int MF, int NF>
static int launch(const TConvArgs& a, int grid, size_t lds, int nw, hipStream_t s) {
  // ring depth: ~16 MFMAs (512 cycles) of work per slot x PF slots covers an L2 / Infinity Cache round trip
  constexpr int PF = NF >= 8 ? 2 : ((MF * NF >= 4) ? 4 : 8);
  constexpr int PF16 = NF >= 8 ? 2 : (MF * NF >= 8 ? 2 : 4);  // 1024-thread workgroups: <= 128 VGPRs per lane
  static std::atomic<uint64_t> attr_set{0};  // dynamic LDS above 64 KB must be opted into once per kernel
  if (DeviceOnce once{attr_set}; once) {
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_kernel<MF, NF, PF16, 16>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxTconvLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_kernel<MF, NF, PF, 8>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxTconvLds));
    ADX_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&tconv_kernel<MF, NF, PF, 4>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)kMaxTconvLds));
    once.commit();
  }
  if (nw == 16) tconv_kernel<MF, NF, PF16, 16><<<dim3(grid), dim3(1024), lds, s>>>(a);
  else if (nw == 8) tconv_kernel<MF, NF, PF, 8><<<dim3(grid), dim3(512), lds, s>>>(a);
  else tconv_kernel<MF, NF, PF, 4><<<dim3(grid), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int tconv_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s) {
  ADX_REQUIRE(io != nullptr, "tconv_forward: null io");
  int rc = tconv_check(d);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(io->batch >= 1, "tconv: batch must be >= 1");
  ADX_REQUIRE(io->x0 != nullptr && io->packed_w != nullptr && io->y != nullptr, "tconv_forward: null tensor");
  ADX_REQUIRE(d->c1 == 0 || io->x1 != nullptr, "tconv_forward: c1 > 0 but x1 is null");
  ADX_REQUIRE(d->groups == 0 || (io->gamma != nullptr && io->beta != nullptr), "tconv_forward: GroupNorm affine missing");
  if (tconv_hs_supported(d)) return tconv_hs_forward(d, io, s);   // split-fp16 MFMA path (tconv_hs.hip)
  if (!tconv_exact_supported(d)) return tconv_generic_forward(d, io, s);   // any other shape (tconv_generic.hip)
  TConvTile t;
  rc = tconv_tile(d, io->batch, &t);
  if (rc != ADX_OK) return rc;
  TConvArgs a;
  a.io = *io;
  a.kind = d->kind; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
  a.c0 = d->c0; a.cin = t.cin; a.cout = d->cout; a.lin = d->lin; a.lout = d->lout;
  a.lin_valid = d->lin_valid > 0 ? d->lin_valid : d->lin;
  a.lout_valid = d->lout_valid > 0 ? d->lout_valid : d->lout;
  a.log2_lout = ilog2_exact(d->lout);
  a.groups = d->groups; a.cg = d->groups > 0 ? d->cout / d->groups : 1; a.eps = d->eps;
  a.ncb = t.ncb; a.nkb = t.nkb;
  a.bt = t.bt; a.ct = t.ct; a.log2_ct = ilog2_exact(t.ct); a.pl = t.pl; a.lp = t.lp; a.rs = t.rs; a.ck = t.ck;
  a.ntiles = t.ntiles; a.cin_pad = t.cin_pad;
  auto dense_src = [&](const float* p, int64_t sb, int64_t sc, int64_t sl) {
    return sl == 1 && sc == d->lin && sb % 4 == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
  };
  a.dense = a.lin_valid == d->lin && d->lin % 4 == 0 && t.pl % 4 == 0 && t.lp % 4 == 0 && dense_src(io->x0, io->x0_sb, io->x0_sc, io->x0_sl) &&
            (d->c1 == 0 || dense_src(io->x1, io->x1_sb, io->x1_sc, io->x1_sl));
  const int grid = ceil_div(io->batch, t.bt) * t.ntiles;
  const int key = t.mf * 16 + t.nf;
  switch (key) {
    case 1 * 16 + 1: return launch<1, 1>(a, grid, t.lds_bytes, t.nw, s);
    case 1 * 16 + 2: return launch<1, 2>(a, grid, t.lds_bytes, t.nw, s);
    case 1 * 16 + 4: return launch<1, 4>(a, grid, t.lds_bytes, t.nw, s);
    case 1 * 16 + 8: return launch<1, 8>(a, grid, t.lds_bytes, t.nw, s);
    case 2 * 16 + 1: return launch<2, 1>(a, grid, t.lds_bytes, t.nw, s);
    case 2 * 16 + 2: return launch<2, 2>(a, grid, t.lds_bytes, t.nw, s);
    case 2 * 16 + 4: return launch<2, 4>(a, grid, t.lds_bytes, t.nw, s);
    case 4 * 16 + 1: return launch<4, 1>(a, grid, t.lds_bytes, t.nw, s);
    case 4 * 16 + 2: return launch<4, 2>(a, grid, t.lds_bytes, t.nw, s);
    default: break;
  }
  adx::set_error("tconv_forward: no kernel for %dx%d fragments", t.mf, t.nf);
  return ADX_ERR_INVALID;
}

}  // namespace adx
