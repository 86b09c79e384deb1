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
#include <array>
#include <map>
#include <mutex>

#include "tconv.h"

namespace adx {

struct TConvArgs {
  adx_tconv_io io;
  int kind, taps, stride, pad;
  int c0, cin, cout, lin, lout, log2_lout;
  int groups, cg;
  float eps;
  int ncb, nkb;
  int bt, ct, log2_ct, pl, lp, rs, ck, ntiles, cin_pad;
};

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

template <int MF, int NF>
__global__ void __launch_bounds__(256) tconv_kernel(const TConvArgs a) {
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

  const f32x4* __restrict__ wp = reinterpret_cast<const f32x4*>(a.io.packed_w);

  for (int c0 = 0; c0 < a.cin_pad; c0 += a.ck) {
    const int ckc = min(a.ck, a.cin_pad - c0);
    if (c0 > 0) __syncthreads();
    // ---- stage [ckc channels][rs columns] of the input, zero padded ------------------------
    for (int col = lane; col < a.rs; col += 64) {
      const int bl = col / a.lp;
      const int ip = col - bl * a.lp - a.pl;
      const int b = b0 + bl;
      const bool ok = bl < a.bt && ip >= 0 && ip < a.lin && b < batch;
      const int64_t off0 = (int64_t)b * a.io.x0_sb + (int64_t)ip * a.io.x0_sl;
      const int64_t off1 = (int64_t)b * a.io.x1_sb + (int64_t)ip * a.io.x1_sl;
      for (int cl = wave; cl < ckc; cl += 4) {
        const int ci = c0 + cl;
        float v = 0.f;
        if (ok && ci < a.cin) {
          v = ci < a.c0 ? a.io.x0[off0 + (int64_t)ci * a.io.x0_sc]
                        : a.io.x1[off1 + (int64_t)(ci - a.c0) * a.io.x1_sc];
        }
        smem[cl * a.rs + col] = v;
      }
    }
    __syncthreads();
    // ---- K loop: this wave takes every 4th (tap, 16-channel block) -------------------------
    const int ncbc = ckc >> 4;
    const int nblk = a.taps * ncbc;
    int tap = 0, cbl = wave;
    while (cbl >= ncbc && tap < a.taps) { cbl -= ncbc; ++tap; }
    for (int i = wave; i < nblk; i += 4) {
      const int kb = tap * a.ncb + (c0 >> 4) + cbl;
      f32x4 bv[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bv[nf] = wp[((size_t)(nt * NF + nf) * a.nkb + kb) * 64 + lane];
      float av[MF][4];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        bool ok;
        const int ip = tconv_in_pos(a.kind, rowl[mf], tap, a.stride, a.pad, ok);
        const int col = ok ? rowoff[mf] + ip : 0;  // column 0 is always a zero pad when a parity can be invalid
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
            acc[mf][nf] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mf][j], bv[nf][j], acc[mf][nf], 0, 0, 0);
      cbl += 4;
      while (cbl >= ncbc) { cbl -= ncbc; ++tap; }
    }
  }

  // ---- epilogue: sum the 4 K-partials through LDS, laid out [sample][channel][pos] ----------
  __syncthreads();
  const int tile_elems = 256 * MF * NF;  // bt * ct * lout
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
  const int n0 = nt * a.ct;
  float* T = smem;  // the sum overwrites partial 0 element by element
  for (int e = tid; e < tile_elems; e += 256) {
    float v = ((P[e] + P[e + tile_elems]) + P[e + 2 * tile_elems]) + P[e + 3 * tile_elems];
    const int c = n0 + ((e >> a.log2_lout) & (a.ct - 1));
    if (a.io.bias != nullptr && c < a.cout) v += a.io.bias[c];
    T[e] = v;
  }
  __syncthreads();

  if (a.groups > 0) {
    // one wave per (sample, group): cg*lout contiguous floats of T
    const int gpt = a.ct / a.cg;
    const int npairs = a.bt * gpt;
    const int n = a.cg << a.log2_lout;
    const float inv_n = 1.0f / (float)n;
    for (int pidx = wave; pidx < npairs; pidx += 4) {
      const int bl = pidx / gpt, g = pidx - bl * gpt;
      const int b = b0 + bl;
      if (b >= batch) continue;
      const float* tp = T + (((bl << a.log2_ct) + g * a.cg) << a.log2_lout);
      float s = 0.f;
      for (int e = lane; e < n; e += 64) s += tp[e];
      const float mean = wave_sum(s) * inv_n;
      float q = 0.f;
      for (int e = lane; e < n; e += 64) {
        const float d = tp[e] - mean;
        q += d * d;
      }
      const float var = wave_sum(q) * inv_n;
      const float rstd = 1.0f / sqrtf(var + a.eps);
      for (int e = lane; e < n; e += 64) {
        const int c = n0 + g * a.cg + (e >> a.log2_lout);
        const int l = e & (a.lout - 1);
        const float sc = rstd * a.io.gamma[c];
        float v = (tp[e] - mean) * sc + a.io.beta[c];
        v = mish_f(v);
        if (a.io.tbias != nullptr) v += a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
        if (a.io.res != nullptr)
          v += a.io.res[(int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)l * a.io.res_sl];
        a.io.y[(int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)l * a.io.y_sl] = v;
      }
    }
  } else {
    for (int e = tid; e < tile_elems; e += 256) {
      const int l = e & (a.lout - 1);
      const int c = n0 + ((e >> a.log2_lout) & (a.ct - 1));
      const int b = b0 + (e >> (a.log2_lout + a.log2_ct));
      if (b < batch && c < a.cout) {
        float v = T[e];
        if (a.io.tbias != nullptr) v += a.io.tbias[(int64_t)b * a.io.tbias_stride + c];
        if (a.io.res != nullptr)
          v += a.io.res[(int64_t)b * a.io.res_sb + (int64_t)c * a.io.res_sc + (int64_t)l * a.io.res_sl];
        a.io.y[(int64_t)b * a.io.y_sb + (int64_t)c * a.io.y_sc + (int64_t)l * a.io.y_sl] = v;
      }
    }
  }
}

// weight image: [cout_pad/16][nkb][64 lanes][4]; element j of lane l in block (tap, cb) is
// W[n = 16*tile + (l & 15)][ci = 16*cb + 4*j + (l >> 4)][tap]  (B operand of 16x16x4, k = l >> 4)
__global__ void tconv_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int kind, int taps,
                                  int cin, int cout, int ncb, int nkb, size_t total) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int lane = (idx >> 2) & 63;
  const size_t blk = idx >> 8;
  const int kb = blk % nkb;
  const int t16 = blk / nkb;
  const int tap = kb / ncb, cb = kb - tap * ncb;
  const int n = t16 * 16 + (lane & 15);
  const int ci = cb * 16 + 4 * j + (lane >> 4);
  float v = 0.f;
  if (n < cout && ci < cin)
    v = kind == 0 ? w[((size_t)n * cin + ci) * taps + tap] : w[((size_t)ci * cout + n) * taps + tap];
  packed[idx] = v;
}

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
    ADX_REQUIRE(d->lout == (d->lin - 1) * 2 - 2 * d->pad + d->taps, "tconv: transposed lout %d inconsistent", d->lout);
    ADX_REQUIRE((d->taps - 1 - d->pad + 1) / 2 >= 1, "tconv: transposed conv needs a left halo");
  }
  ADX_REQUIRE(ilog2_exact(d->lout) >= 0 && d->lout <= 64, "tconv: lout must be a power of two <= 64, got %d", d->lout);
  if (d->groups > 0) {
    ADX_REQUIRE(d->cout % d->groups == 0, "tconv: cout %d not divisible by groups %d", d->cout, d->groups);
    const int cg = d->cout / d->groups;
    ADX_REQUIRE(ilog2_exact(cg) >= 0 && cg <= 128, "tconv: group width %d must be a power of two <= 128", cg);
    ADX_REQUIRE(d->cout % 16 == 0, "tconv: GroupNorm convs need cout %% 16 == 0");
  }
  return ADX_OK;
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
  const int lp_min = t->pl + d->lin + pr;
  // the pitch search costs ~1 ms of host time: memoise it per geometry (thread-safe)
  {
    const std::array<int, 8> key{d->kind, d->taps, d->stride, d->pad, d->lin, d->lout, t->bt, t->mf};
    static std::mutex mu;
    static std::map<std::array<int, 8>, std::pair<int, int>> memo;
    std::lock_guard<std::mutex> lock(mu);
    auto it = memo.find(key);
    if (it == memo.end()) {
      int best_cost = 1 << 30, best_lp = lp_min, best_rs = t->bt * lp_min;
      for (int lp = lp_min; lp <= lp_min + 16; ++lp)
        for (int rs = t->bt * lp; rs <= t->bt * lp + 32; ++rs) {
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
  // keep the staged tile <= 64 KB so two workgroups fit a CU; chunk the channels otherwise
  int ck = (16384 / t->rs) / 16 * 16;
  if (ck < 16) ck = 16;
  t->ck = ck < t->cin_pad ? ck : t->cin_pad;
  const size_t stage = (size_t)t->ck * t->rs;
  const size_t epi = (size_t)4 * 256 * t->mf * t->nf;
  t->lds_bytes = sizeof(float) * (stage > epi ? stage : epi);
  ADX_REQUIRE(t->lds_bytes <= 64 * 1024, "tconv: LDS tile of %zu bytes exceeds 64 KB", t->lds_bytes);
  return ADX_OK;
}

size_t tconv_packed_floats(const adx_tconv_desc* d) {
  const int cin_pad = round_up(d->c0 + d->c1, 16);
  return (size_t)(round_up(d->cout, 16) / 16) * d->taps * (cin_pad / 16) * 256;
}

int tconv_pack(const adx_tconv_desc* d, const float* w, float* packed, hipStream_t s) {
  int rc = tconv_check(d);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(w != nullptr && packed != nullptr, "tconv_pack: null pointer");
  const int cin = d->c0 + d->c1;
  const int ncb = round_up(cin, 16) / 16;
  const size_t total = tconv_packed_floats(d);
  tconv_pack_kernel<<<dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s>>>(w, packed, d->kind, d->taps, cin,
                                                                             d->cout, ncb, d->taps * ncb, total);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

template <int MF, int NF>
static int launch(const TConvArgs& a, int grid, size_t lds, hipStream_t s) {
  tconv_kernel<MF, NF><<<dim3(grid), dim3(256), lds, s>>>(a);
  ADX_LAUNCH_CHECK();
  return ADX_OK;
}

int tconv_forward(const adx_tconv_desc* d, const adx_tconv_io* io, hipStream_t s) {
  ADX_REQUIRE(io != nullptr, "tconv_forward: null io");
  TConvTile t;
  int rc = tconv_tile(d, io->batch, &t);
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(io->x0 != nullptr && io->packed_w != nullptr && io->y != nullptr, "tconv_forward: null tensor");
  ADX_REQUIRE(d->c1 == 0 || io->x1 != nullptr, "tconv_forward: c1 > 0 but x1 is null");
  ADX_REQUIRE(d->groups == 0 || (io->gamma != nullptr && io->beta != nullptr), "tconv_forward: GroupNorm affine missing");
  TConvArgs a;
  a.io = *io;
  a.kind = d->kind; a.taps = d->taps; a.stride = d->stride; a.pad = d->pad;
  a.c0 = d->c0; a.cin = t.cin; a.cout = d->cout; a.lin = d->lin; a.lout = d->lout;
  a.log2_lout = ilog2_exact(d->lout);
  a.groups = d->groups; a.cg = d->groups > 0 ? d->cout / d->groups : 1; a.eps = d->eps;
  a.ncb = t.ncb; a.nkb = t.nkb;
  a.bt = t.bt; a.ct = t.ct; a.log2_ct = ilog2_exact(t.ct); a.pl = t.pl; a.lp = t.lp; a.rs = t.rs; a.ck = t.ck;
  a.ntiles = t.ntiles; a.cin_pad = t.cin_pad;
  const int grid = ceil_div(io->batch, t.bt) * t.ntiles;
  const int key = t.mf * 16 + t.nf;
  switch (key) {
    case 1 * 16 + 1: return launch<1, 1>(a, grid, t.lds_bytes, s);
    case 1 * 16 + 2: return launch<1, 2>(a, grid, t.lds_bytes, s);
    case 1 * 16 + 4: return launch<1, 4>(a, grid, t.lds_bytes, s);
    case 1 * 16 + 8: return launch<1, 8>(a, grid, t.lds_bytes, s);
    case 2 * 16 + 1: return launch<2, 1>(a, grid, t.lds_bytes, s);
    case 2 * 16 + 2: return launch<2, 2>(a, grid, t.lds_bytes, s);
    case 2 * 16 + 4: return launch<2, 4>(a, grid, t.lds_bytes, s);
    case 4 * 16 + 1: return launch<4, 1>(a, grid, t.lds_bytes, s);
    case 4 * 16 + 2: return launch<4, 2>(a, grid, t.lds_bytes, s);
    default: break;
  }
  adx::set_error("tconv_forward: no kernel for %dx%d fragments", t.mf, t.nf);
  return ADX_ERR_INVALID;
}

}  // namespace adx
