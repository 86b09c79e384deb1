// Native executor for TemporalMapUnet.forward minus the perception pass
// (modeling/temporal.py:204-245): one C call issues the whole launch sequence on a stream, so a
// denoising loop has no Python between kernels and can be captured in a HIP graph.
//
// Topology (modeling/temporal.py:59-195): n levels of [ResBlock, ResBlock, Downsample (not on
// the last)], two mid ResBlocks, n-1 levels of [cat(skip), ResBlock, ResBlock, Upsample (on ALL
// of them)], then Conv1dBlock + 1x1 head.  A ResBlock (temporal.py:23-55) is
//   h = Mish(GN(conv5(x))) + Linear(Mish(cond))[:, :, None];  y = Mish(GN(conv5(h))) + R(x)
// and costs 2 launches (+1 when R is a 1x1 conv): GN, Mish, bias and both adds live in the
// conv epilogues.  The 16 per-block Linears share their input, so they run up front as ONE
// GEMM [rows, 2*dim] x [2*dim, sum(C)] through the same MFMA kernel (a length-1 "conv").
#include <algorithm>
#include <vector>

#include "tconv.h"

#include "batch_ops.h"
#include "unet_internal.h"
#include "tconv_pack.h"

namespace adx {

static int pow2_ceil(int v) { int p = 1; while (p < v) p <<= 1; return p; }

static adx_tconv_desc conv_desc(int kind, int taps, int stride, int pad, int c0, int c1, int cout, int lin, int lout,
                                int groups) {
  // lin / lout are the real lengths; a horizon that is not a power of two (24 -> 24, 12, 6, 3) runs on the next power
  // of two with the real lengths as masks (adx_tconv_desc::lin_valid)
  adx_tconv_desc d{};
  d.kind = kind; d.taps = taps; d.stride = stride; d.pad = pad;
  d.c0 = c0; d.c1 = c1; d.cout = cout; d.lin = pow2_ceil(lin); d.lout = pow2_ceil(lout);
  if (d.lin != lin || d.lout != lout) { d.lin_valid = lin; d.lout_valid = lout; }
  d.groups = groups; d.eps = 1e-5f;
  return d;
}

static size_t align64(size_t v) { return (v + 63) / 64 * 64; }

struct Builder {
  adx_unet* u;
  int next_param = 0;
  size_t off = 0;
  size_t take(size_t n) {
    const size_t o = off;
    off = align64(off + n);
    return o;
  }
  // Conv1dBlock: conv weight, conv bias, GN weight, GN bias (modeling/helpers.py:103-109)
  ConvLayer conv_block(int c0, int c1, int cout, int len) {
    ConvLayer L;
    L.d = conv_desc(0, 5, 1, 2, c0, c1, cout, len, len, 8);
    L.p_w = next_param++; L.p_b = next_param++; L.p_g = next_param++; L.p_be = next_param++;
    L.o_w = take(tconv_packed_floats(&L.d));
    L.o_b = take(cout); L.o_g = take(cout); L.o_be = take(cout);
    return L;
  }
  ConvLayer plain(int kind, int taps, int stride, int pad, int c0, int c1, int cout, int lin, int lout) {
    ConvLayer L;
    L.d = conv_desc(kind, taps, stride, pad, c0, c1, cout, lin, lout, 0);
    L.p_w = next_param++; L.p_b = next_param++;
    L.o_w = take(tconv_packed_floats(&L.d));
    L.o_b = take(cout);
    return L;
  }
  ResBlock res_block(int c0, int c1, int cout, int len) {
    ResBlock B;
    B.c0 = c0; B.c1 = c1; B.cout = cout; B.len = pow2_ceil(len);     // pitch of the activation buffers
    B.a = conv_block(c0, c1, cout, len);
    B.b = conv_block(cout, 0, cout, len);
    B.p_tw = next_param++; B.p_tb = next_param++;
    B.tb_off = u->sum_c;
    u->sum_c += cout;
    B.has_r = (c0 + c1) != cout;
    if (B.has_r) B.r = plain(0, 1, 1, 0, c0, c1, cout, len, len);
    return B;
  }
};

// ---- chains (tconv_chain.hip): which runs of layers go into one launch ---------------------------------------------------
constexpr int kChainMaxChannels = 128;     // all channels of a layer in one workgroup: every workgroup streams every weight

// ADX_UNET_CHAIN=0 turns the chains off (every level layer by layer: the A/B)
static bool chains_enabled() { return debug_switches().unet_chain; }

static int ilog2_floor(int v) { int l = 0; while ((2 << l) <= v) ++l; return l; }

// The kernel's argument block for `rows` samples, `bt` per workgroup: LDS layout + every stage descriptor.  Returns the
// dynamic LDS bytes (0: this bt does not fit / is not allowed).  Pointers and strides are filled by the caller.
static size_t chain_args(const ChainPlan& cp, int bt, ChainArgs& a) {
  memset(&a, 0, sizeof(a));
  const int n = (int)cp.st.size();
  a.n_stages = n; a.bt = bt;
  a.in_c0 = cp.in_c0; a.in_c1 = cp.in_c1; a.in_len = cp.len;
  a.in_cpad = chain_cin_pad(&cp.st[0].L->d);
  // cell buffers (16-byte units), fp32 tiles, exchange area
  size_t cell16[kChainMaxCells] = {0, 0, 0, 0}, f[2] = {0, 0}, xch = 0, tb_floats = 0;
  auto need = [&](int buf, size_t rows, int pitch) { if (buf >= 0) cell16[buf] = std::max(cell16[buf], (rows + 1) * (size_t)pitch); };
  for (int i = 0; i < n; ++i) {
    const ChainStagePlan& sp = cp.st[i];
    const adx_tconv_desc& d = sp.L->d;
    const int cout_pad = round_up(d.cout, 16);
    need(sp.src, (size_t)bt * d.lin, 2 * (chain_cin_pad(&d) / 8) + 1);
    need(sp.dst, (size_t)bt * d.lout, 2 * (cout_pad / 8) + 1);
    if (sp.R != nullptr) need(sp.r_src, (size_t)bt * d.lout, 2 * (chain_cin_pad(&sp.R->d) / 8) + 1);
    const size_t rows_pad = (size_t)round_up(bt * d.lout, 16);
    if (rows_pad > 64) return 0;
    f[sp.f] = std::max(f[sp.f], rows_pad * (size_t)(cout_pad + 4));
    if (d.groups > 0 && d.lout >= 32) {
      const size_t tiles = (rows_pad / 16) * (cout_pad / 16);
      if (tiles % kChainWaves != 0) return 0;            // every wave must reach the exchange barrier the same number of times
      xch = std::max(xch, tiles * 8);
    }
    if (sp.tb_col >= 0) tb_floats += (size_t)bt * cout_pad;
  }
  size_t off = 0;     // floats
  int cell_off[kChainMaxCells], f_off[2];
  for (int k = 0; k < kChainMaxCells; ++k) { cell_off[k] = (int)off; off += cell16[k] * 4; }
  for (int k = 0; k < 2; ++k) { f_off[k] = (int)off; off += (f[k] + 3) / 4 * 4; }
  a.par_lds = (int)off; a.par_floats = (int)cp.par_floats; a.par_src = (int)cp.o_par;
  off += (cp.par_floats + 3) / 4 * 4;
  const int tb_base = (int)off;
  off += tb_floats;
  a.xch_lds = (int)off;
  off += xch;
  a.tab_lds = (int)off;
  off += (size_t)n * (sizeof(ChainStage) / 4);
  a.in_cells = cell_off[cp.st[0].src];
  int tb_next = tb_base;
  for (int i = 0; i < n; ++i) {
    const ChainStagePlan& sp = cp.st[i];
    const adx_tconv_desc& d = sp.L->d;
    ChainStage& st = a.st[i];
    const int cin_pad = chain_cin_pad(&d), cout_pad = round_up(d.cout, 16);
    const int cg = d.groups > 0 ? d.cout / d.groups : 1;
    st.flags = (d.groups > 0 ? kChGn : 0) | (sp.tb_col >= 0 ? kChTb : 0) | (sp.res_identity ? kChResIdentity : 0) |
               (sp.R != nullptr ? kChResConv : 0) | (sp.out >= 0 ? kChOut : 0) | (sp.dst >= 0 ? kChCells : 0) |
               (ilog2_floor(cg) << 8) | ((sp.out >= 0 ? sp.out : 0) << 12);
    st.conv = d.kind | (d.taps << 8) | (d.stride << 16) | (d.pad << 24);
    st.log2_spt = ilog2_floor(cin_pad / 32);
    st.ns_main = chain_steps(&d);
    st.ns_r = sp.R != nullptr ? chain_steps(&sp.R->d) : 0;
    st.lin = d.lin; st.log2_lout = ilog2_floor(d.lout);
    st.cout = d.cout; st.log2_nct = ilog2_floor(cout_pad / 16);
    st.w_off = (int)sp.L->o_cw;
    st.src = cell_off[sp.src]; st.src_pitch = 2 * (cin_pad / 8) + 1;
    if (sp.R != nullptr) {
      const int rpad = chain_cin_pad(&sp.R->d);
      st.r_src = cell_off[sp.r_src]; st.r_pitch = 2 * (rpad / 8) + 1; st.r_log2_spt = ilog2_floor(rpad / 32);
      st.r_zrow = bt * d.lout;
    } else {
      st.r_src = st.src; st.r_pitch = st.src_pitch; st.r_log2_spt = 0; st.r_zrow = bt * d.lin;
    }
    st.dst = sp.dst >= 0 ? cell_off[sp.dst] : 0; st.dst_pitch = 2 * (cout_pad / 8) + 1;
    st.f_dst = f_off[sp.f];
    st.par = a.par_lds + (int)sp.par_off;
    if (sp.tb_col >= 0) {
      st.tbl = tb_next;
      a.tb_col[a.n_tb] = sp.tb_col; a.tb_cout[a.n_tb] = cout_pad; a.tb_lds[a.n_tb] = tb_next;
      ++a.n_tb;
      tb_next += bt * cout_pad;
    }
    st.rows_out = bt * d.lout; st.zrow = bt * d.lin;
    st.inv_n = 1.0f / (float)(cg * d.lout); st.eps = d.eps;
    const int nrt = ceil_div(bt * d.lout, 16);
    if ((nrt & (nrt - 1)) != 0) return 0;                    // row tiles per stage: a power of two (tile index arithmetic)
    st.log2_nrt = ilog2_floor(nrt);
    st.n_tiles = (cout_pad / 16) * nrt;
  }
  for (int i = 0; i < n; ++i) {
    ChainStage& st = a.st[i];
    if (i + 1 < n) {
      const ChainStage& nx = a.st[i + 1];
      st.nx_w_off = nx.w_off; st.nx_nsteps = nx.ns_main + nx.ns_r; st.nx_log2_nrt = nx.log2_nrt; st.nx_n_tiles = nx.n_tiles;
    }
  }
  return off * sizeof(float);
}

// b0 / b1: the level's two residual blocks; tail: its down / up conv (may be null); h0 / h1: final_conv behind the last up level
static void plan_chain(ChainPlan* cp, const ResBlock& b0, const ResBlock& b1, const ConvLayer* tail, const ConvLayer* h0,
                       const ConvLayer* h1, bool up_level, int level) {
  cp->valid = false;
  cp->st.clear();
  // ADX_CHAIN_MASK: bit i = down level i, bit 8 + i = up level i (diagnostic: which levels are chained; default all)
  const unsigned mask = debug_switches().chain_mask;
  if (((mask >> (up_level ? 8 + level : level)) & 1u) == 0) return;
  if (!chains_enabled() || !b0.has_r || b1.has_r) return;      // block 0 changes the channel count (R = 1x1 conv), block 1 keeps it
  if (b0.cout > kChainMaxChannels || b1.cout != b0.cout) return;
  const ConvLayer* all[] = {&b0.a, &b0.b, &b0.r, &b1.a, &b1.b, tail, h0, h1};
  constexpr int max_steps = 48;
  for (const ConvLayer* L : all) {
    if (L == nullptr) continue;
    if (!chain_layer_ok(&L->d)) return;
    // One wave walks ALL K-steps of its tile: a reduction of more than ~48 steps (the 1024 -> 256 / 512 -> 128 convs on the
    // concatenated input of the deeper up levels: 80) is faster as its own launch with the reduction split over waves and
    // workgroups (measured: that level as a chain +1.3 us at 128 rows, +4.1 us at 2 rows)
    if (chain_steps(&L->d) > max_steps) return;
  }
  cp->in_c0 = b0.c0; cp->in_c1 = b0.c1; cp->len = b0.a.d.lin;
  auto add = [&](const ConvLayer& L, int src, int dst, int f, int out, int tb_col) -> ChainStagePlan& {
    ChainStagePlan sp;
    sp.L = &L; sp.src = src; sp.dst = dst; sp.f = f; sp.out = out; sp.tb_col = tb_col;
    cp->st.push_back(sp);
    return cp->st.back();
  };
  // cell buffers: 0 = block input (and, later, block 1's output), 1 = a block's inner activation, 2 = block 0's output
  add(b0.a, 0, 1, 0, -1, b0.tb_off);
  {
    ChainStagePlan& s1 = add(b0.b, 1, 2, 1, -1, -1);
    s1.R = &b0.r; s1.r_src = 0;
  }
  add(b1.a, 2, 1, 0, -1, b1.tb_off);
  {
    ChainStagePlan& s3 = add(b1.b, 1, tail != nullptr ? 0 : -1, 1, up_level ? (tail != nullptr ? -1 : 0) : 0, -1);
    s3.res_identity = true;
  }
  cp->with_head = false;
  if (tail != nullptr) {
    const bool head = up_level && h0 != nullptr && h1 != nullptr;
    add(*tail, 0, head ? 1 : -1, 0, head ? -1 : (up_level ? 0 : 1), -1);
    if (head) {
      add(*h0, 1, 2, 1, -1, -1);
      add(*h1, 2, -1, 0, 0, -1);
      cp->with_head = true;
    }
  }
  // the chain's parameter block: [bias | gamma | beta | residual bias] x cout_pad per stage
  cp->par_floats = 0;
  cp->max_len = 0;
  for (auto& sp : cp->st) {
    sp.par_off = cp->par_floats;
    cp->par_floats += (size_t)4 * round_up(sp.L->d.cout, 16);
    cp->max_len = std::max(cp->max_len, sp.L->d.lout);
  }
  // must run with the fewest samples per workgroup that fill a 16-row tile
  ChainArgs a;
  const size_t lds = chain_args(*cp, std::max(1, 16 / cp->len), a);
  if (lds == 0 || lds > kChainMaxLds) { cp->st.clear(); return; }
  cp->valid = true;
}

static int build(adx_unet* u) {
  const adx_unet_config& c = u->cfg;
  ADX_REQUIRE(c.n_mults >= 1 && c.n_mults <= 8, "unet: n_mults %d out of range", c.n_mults);
  ADX_REQUIRE(c.guidance >= 0 && c.guidance <= 2, "unet: guidance %d out of range", c.guidance);
  ADX_REQUIRE(c.dim >= 16 && c.dim % 16 == 0, "unet: dim %d must be a multiple of 16", c.dim);
  ADX_REQUIRE(c.transition_dim >= 4, "unet: transition_dim %d too small", c.transition_dim);
  const int n = c.n_mults;
  u->n_levels = n;
  ADX_REQUIRE(c.horizon % (1 << (n - 1)) == 0, "unet: horizon %d not divisible by %d", c.horizon, 1 << (n - 1));
  Builder B{u};
  // parameter order == TemporalMapUnet.__init__ registration order (temporal.py:87-194)
  if (c.guidance == 1) {
    u->p_c0w = B.next_param++; u->p_c0b = B.next_param++; u->p_c2w = B.next_param++; u->p_c2b = B.next_param++;
  }
  u->p_t1w = B.next_param++; u->p_t1b = B.next_param++; u->p_t3w = B.next_param++; u->p_t3b = B.next_param++;
  std::vector<int> dims(n + 1);
  dims[0] = c.transition_dim;
  for (int i = 0; i < n; ++i) dims[i + 1] = c.dim * c.dim_mults[i];
  int len = c.horizon;
  std::vector<int> level_len(n);
  for (int i = 0; i < n; ++i) {
    const int ci = dims[i], co = dims[i + 1];
    level_len[i] = len;
    u->blocks.push_back(B.res_block(ci, 0, co, len));
    u->blocks.push_back(B.res_block(co, 0, co, len));
    if (i < n - 1) {
      u->downs.push_back(B.plain(0, 3, 2, 1, co, 0, co, len, len / 2));
      len /= 2;
    }
  }
  // ups are registered before the mid blocks (temporal.py:104-105,152-178) but run after them
  std::vector<ResBlock> up_blocks;
  std::vector<ConvLayer> up_convs;
  int ulen = len;
  for (int i = 0; i < n - 1; ++i) {
    const int ci = dims[n - 1 - i], co = dims[n - i];  // reversed(in_out[1:])
    up_blocks.push_back(B.res_block(co, co, ci, ulen));
    up_blocks.push_back(B.res_block(ci, 0, ci, ulen));
    up_convs.push_back(B.plain(1, 4, 2, 1, ci, 0, ci, ulen, ulen * 2));
    ulen *= 2;
  }
  const int mid = dims[n];
  // the time-bias column offsets follow execution order only by convention; keep the
  // registration-order offsets assigned by res_block() above
  ResBlock m1 = B.res_block(mid, 0, mid, len);
  ResBlock m2 = B.res_block(mid, 0, mid, len);
  u->blocks.push_back(m1);
  u->blocks.push_back(m2);
  for (auto& b : up_blocks) u->blocks.push_back(b);
  u->ups = up_convs;
  const int fin = n > 1 ? dims[1] : dims[n];
  ADX_REQUIRE(ulen == c.horizon || n == 1, "unet: up path ends at length %d, expected %d", ulen, c.horizon);
  u->head0 = B.conv_block(fin, 0, fin, ulen);
  u->out_ch = c.guidance == 2 ? 3 : c.transition_dim;
  u->head1 = B.plain(0, 1, 1, 0, fin, 0, u->out_ch, ulen, ulen);
  u->n_params = B.next_param;
  // the fused block-Linear: [rows, 2 dim] -> [rows, sum_c]
  u->tlin.d = conv_desc(0, 1, 1, 0, 2 * c.dim, 0, u->sum_c, 1, 1, 0);
  u->tlin.o_w = B.take(tconv_packed_floats(&u->tlin.d));
  u->o_tlin_b = B.take(u->sum_c);
  u->o_tlin_raw = B.take((size_t)u->sum_c * 2 * c.dim);
  u->o_freqs = B.take(c.dim / 2);
  u->o_t1w = B.take((size_t)4 * c.dim * c.dim); u->o_t1b = B.take(4 * c.dim);
  u->o_t3w = B.take((size_t)4 * c.dim * c.dim); u->o_t3b = B.take(c.dim);
  if (c.guidance == 1) {
    u->o_c0w = B.take(2 * c.dim); u->o_c0b = B.take(c.dim);
    u->o_c2w = B.take((size_t)c.dim * c.dim); u->o_c2b = B.take(c.dim);
  }
  // chains: one per level where one workgroup can hold every channel (plan_chain decides); their layers get a second
  // weight image in the chain kernel's layout
  u->down_chains.assign(n, ChainPlan{});
  u->up_chains.assign(n > 1 ? n - 1 : 0, ChainPlan{});
  auto give_images = [&](ChainPlan& cp) {      // called once the level qualifies; plan_chain is then run again (offsets)
    for (auto& sp : cp.st) {
      ConvLayer* L = const_cast<ConvLayer*>(sp.L);
      if (!L->chained) {
        L->chained = true;
        L->o_cw = B.take(chain_packed_floats(&L->d) + (sp.R != nullptr ? chain_packed_floats(&sp.R->d) : 0));
      }
    }
    cp.o_par = B.take(cp.par_floats);
  };
  for (int i = 0; i < n; ++i) {
    ResBlock& b0 = u->blocks[2 * i];
    ResBlock& b1 = u->blocks[2 * i + 1];
    ConvLayer* dn = i < n - 1 ? &u->downs[i] : nullptr;
    plan_chain(&u->down_chains[i], b0, b1, dn, nullptr, nullptr, false, i);
    if (u->down_chains[i].valid) give_images(u->down_chains[i]);
  }
  for (int i = 0; i < n - 1; ++i) {
    ResBlock& b0 = u->blocks[2 * n + 2 + 2 * i];
    ResBlock& b1 = u->blocks[2 * n + 2 + 2 * i + 1];
    const bool last = i == n - 2;
    plan_chain(&u->up_chains[i], b0, b1, &u->ups[i], last ? &u->head0 : nullptr, last ? &u->head1 : nullptr, true, i);
    if (u->up_chains[i].valid) give_images(u->up_chains[i]);
  }
  // the pipeline run (tconv_pipe.hip): the last level's block 0 second conv, block 1 and both mid blocks when they are seven
  // convs of ONE shape whose per-workgroup weight share fits the LDS (three live taps at 512 channels: MODEL.HORIZON = 16)
  {
    ResBlock& b0 = u->blocks[2 * (n - 1)];
    ResBlock& b1 = u->blocks[2 * (n - 1) + 1];
    ResBlock& m1 = u->blocks[2 * n];
    ResBlock& m2 = u->blocks[2 * n + 1];
    ConvLayer* run[7] = {&b0.b, &b1.a, &b1.b, &m1.a, &m1.b, &m2.a, &m2.b};
    const adx_tconv_desc& d0 = b0.b.d;
    bool ok = debug_switches().unet_pipe && !u->down_chains[n - 1].valid && b0.has_r && !b1.has_r && !m1.has_r && !m2.has_r &&
              d0.lin == b0.len;
    for (ConvLayer* L : run) {
      const adx_tconv_desc& d = L->d;
      ok = ok && d.kind == 0 && d.stride == 1 && d.c1 == 0 && d.c0 == d0.cout && d.cout == d0.cout && d.taps == d0.taps &&
           d.pad == d0.pad && d.lin == d0.lin && d.lout == d0.lin && d.groups == d0.groups && d.groups > 0 && d.eps == d0.eps &&
           d.w_layout == 0 && d.w_flip == 0;
    }
    ok = ok && pipe_shape_ok(d0.cout, d0.lin, 1, d0.taps, d0.pad, d0.groups) && (d0.cout / d0.groups) % 4 == 0 &&
         tconv_hs_kernel_image(&d0);      // the pipeline's images are re-laid from the K-split kernel's (not with ADX_TCONV_EXACT=1)
    u->pipe_ok = ok;
    if (ok)
      for (ConvLayer* L : run) {
        L->piped = true;
        L->o_pw = B.take(pipe_packed_floats(d0.cout, d0.taps, d0.pad, d0.lin));
      }
  }
  u->packed_floats = B.off;
  // validate every layer's geometry now so that forward() cannot fail on shape grounds (any layer one of the three
  // temporal kernels covers: tconv_check)
  for (auto& b : u->blocks) {
    int rc = tconv_check(&b.a.d);
    if (rc == ADX_OK) rc = tconv_check(&b.b.d);
    if (rc == ADX_OK && b.has_r) rc = tconv_check(&b.r.d);
    if (rc != ADX_OK) return rc;
  }
  for (auto& l : u->downs) { int rc = tconv_check(&l.d); if (rc != ADX_OK) return rc; }
  for (auto& l : u->ups) { int rc = tconv_check(&l.d); if (rc != ADX_OK) return rc; }
  int rc = tconv_check(&u->head0.d);
  if (rc == ADX_OK) rc = tconv_check(&u->head1.d);
  if (rc == ADX_OK) rc = tconv_check(&u->tlin.d);
  return rc;
}

// queued: adx_unet_pack flushes the whole list as a handful of launches (batch_ops.h)
static int copy_f(float* dst, const float* src, size_t n, hipStream_t) {
  batch_copy_add(dst, src, n);
  return ADX_OK;
}

static int pack_layer(const ConvLayer& L, const float* const* P, float* base, hipStream_t s, const ConvLayer* rider = nullptr) {
  int rc = tconv_pack(&L.d, P[L.p_w], base + L.o_w, s);
  if (rc == ADX_OK && L.chained)       // chain image; `rider` = the block's 1x1 residual conv, stored behind this conv's steps
    rc = chain_pack(&L.d, P[L.p_w], rider != nullptr ? &rider->d : nullptr, rider != nullptr ? P[rider->p_w] : nullptr, base + L.o_cw, s);
  if (rc == ADX_OK && L.p_b >= 0) rc = copy_f(base + L.o_b, P[L.p_b], L.d.cout, s);
  if (rc == ADX_OK && L.p_g >= 0) rc = copy_f(base + L.o_g, P[L.p_g], L.d.cout, s);
  if (rc == ADX_OK && L.p_be >= 0) rc = copy_f(base + L.o_be, P[L.p_be], L.d.cout, s);
  return rc;
}

static void fill_io(adx_tconv_io& io, const ConvLayer& L, const float* base) {
  io.packed_w = base + L.o_w;
  io.bias = L.p_b >= 0 ? base + L.o_b : nullptr;
  io.gamma = L.p_g >= 0 ? base + L.o_g : nullptr;
  io.beta = L.p_be >= 0 ? base + L.o_be : nullptr;
}

struct Act {  // an activation tensor [rows][c][len] with explicit strides
  const float* p = nullptr;
  int64_t sb = 0, sc = 0, sl = 0;
};

static Act dense(float* p, int c, int len) { return Act{p, (int64_t)c * len, (int64_t)len, 1}; }

static adx_tconv_io make_io(const ConvLayer& L, const float* base, const Act& x0, const Act* x1, const float* tbias,
                            int64_t tb_stride, const Act* res, float* y, int64_t y_sb, int64_t y_sc, int64_t y_sl, int rows);

// Split-reduction scratch of the forward call being enqueued on this thread (a region of ITS workspace; consumed in stream
// order by the launches it is handed to, so two calls on different streams never share it).
constexpr size_t kSplitScratchFloats = (size_t)2 << 20;
// ... of which the LAST kPipeTailFloats belong to the pipeline launch alone (tconv_pipe.hip: its epoch-tagged records must
// not be written by anything else -- a split reduction's partial tile in a tag slot would be data posing as a signal)
constexpr size_t kPipeTailFloats = (size_t)192 << 10;
static thread_local float* t_split_scratch = nullptr;
constexpr size_t kTicketWords = 256;
constexpr int kEpochSlot = 240;          // ticket word that receives the forward's number (the pipeline's stage tags)
static thread_local uint32_t* t_split_tickets = nullptr;     // kTicketWords words (adx_tconv_io::tickets), cleared by every forward

static int run_conv(const ConvLayer& L, const float* base, const Act& x0, const Act* x1, const float* tbias,
                    int64_t tb_stride, const Act* res, float* y, int64_t y_sb, int64_t y_sc, int64_t y_sl, int rows,
                    hipStream_t s) {
  const adx_tconv_io io = make_io(L, base, x0, x1, tbias, tb_stride, res, y, y_sb, y_sc, y_sl, rows);
  return tconv_forward(&L.d, &io, s);
}

static adx_tconv_io make_io(const ConvLayer& L, const float* base, const Act& x0, const Act* x1, const float* tbias,
                            int64_t tb_stride, const Act* res, float* y, int64_t y_sb, int64_t y_sc, int64_t y_sl, int rows) {
  adx_tconv_io io;
  memset(&io, 0, sizeof(io));
  io.x0 = x0.p; io.x0_sb = x0.sb; io.x0_sc = x0.sc; io.x0_sl = x0.sl;
  if (x1 != nullptr) { io.x1 = x1->p; io.x1_sb = x1->sb; io.x1_sc = x1->sc; io.x1_sl = x1->sl; }
  fill_io(io, L, base);
  io.tbias = tbias; io.tbias_stride = tb_stride;
  if (res != nullptr) { io.res = res->p; io.res_sb = res->sb; io.res_sc = res->sc; io.res_sl = res->sl; }
  io.y = y; io.y_sb = y_sb; io.y_sc = y_sc; io.y_sl = y_sl;
  io.batch = rows;
  io.scratch = t_split_scratch;
  io.scratch_floats = t_split_scratch != nullptr ? (int64_t)(kSplitScratchFloats - kPipeTailFloats) : 0;
  io.tickets = t_split_tickets;
  return io;
}

// time_embed [rows][dim], mish_cond [rows][2 dim] and the time-bias matrix tb[rows][sum_c] = all 16 block Linears at once
// (temporal.py:206-216 + the `time_mlp` of every ResidualTemporalMapBlock, helpers.py:121-123)
static int time_conditioning(adx_unet* u, const float* base, const adx_unet_io* io, int rows, float* te, float* mc, float* tb,
                             hipStream_t s) {
  const int dim = u->cfg.dim;
  adx_embed_weights ew;
  memset(&ew, 0, sizeof(ew));
  ew.freqs = base + u->o_freqs;
  ew.w1 = base + u->o_t1w; ew.b1 = base + u->o_t1b; ew.w3 = base + u->o_t3w; ew.b3 = base + u->o_t3b;
  if (u->cfg.guidance == 1) {
    ew.cw0 = base + u->o_c0w; ew.cb0 = base + u->o_c0b; ew.cw2 = base + u->o_c2w; ew.cb2 = base + u->o_c2b;
  }
  int rc = embed_forward(&ew, dim, io->t, io->t_rows, u->cfg.guidance == 1 ? io->cond : nullptr, io->img_feature,
                         io->feat_rows, rows, te, mc, s);
  if (rc != ADX_OK) return rc;
  adx_tconv_io lio;
  memset(&lio, 0, sizeof(lio));
  lio.x0 = mc; lio.x0_sb = 2 * dim; lio.x0_sc = 1; lio.x0_sl = 0;
  lio.packed_w = base + u->tlin.o_w; lio.bias = base + u->o_tlin_b;
  lio.y = tb; lio.y_sb = u->sum_c; lio.y_sc = 1; lio.y_sl = 0;
  lio.batch = rows;
  return tconv_forward(&u->tlin.d, &lio, s);
}

static int check_conditioning_io(const adx_unet* u, const adx_unet_io* io, const char* who) {
  const int rows = io->rows;
  ADX_REQUIRE(io->img_feature && io->t, "%s: null tensor", who);
  ADX_REQUIRE(io->t_rows >= 1 && rows % io->t_rows == 0 && io->feat_rows >= 1 && rows % io->feat_rows == 0,
              "%s: rows %d must be a multiple of t_rows %d and feat_rows %d", who, rows, io->t_rows, io->feat_rows);
  if (u->cfg.guidance != 1)
    ADX_REQUIRE(io->t_rows == rows && io->feat_rows == rows,
                "%s: time/img batch must equal the trajectory batch unless FREE_GUIDANCE "
                "(the reference's torch.cat fails otherwise, temporal.py:213)", who);
  return ADX_OK;
}

}  // namespace adx

using namespace adx;

extern "C" {

int adx_unet_create(const adx_unet_config* cfg, adx_unet** out) {
  ADX_REQUIRE(cfg != nullptr && out != nullptr, "adx_unet_create: null argument");
  adx_unet* u = new adx_unet();
  u->cfg = *cfg;
  const int rc = build(u);
  if (rc != ADX_OK) {
    delete u;
    return rc;
  }
  *out = u;
  return ADX_OK;
}

void adx_unet_destroy(adx_unet* u) {
  delete u;
}

int adx_unet_num_params(const adx_unet* u) { return u ? u->n_params : 0; }

size_t adx_unet_packed_bytes(const adx_unet* u) { return u ? u->packed_floats * sizeof(float) : 0; }

int adx_unet_pack(adx_unet* u, const float* const* P, int32_t n_params, const float* freqs, void* packed,
                  adx_stream stream) {
  ADX_REQUIRE(u && P && freqs && packed, "adx_unet_pack: null argument");
  // CLASSIFIER_GUIDANCE models carry TrajPredict's parameters after the head; they are not ours
  ADX_REQUIRE(n_params >= u->n_params, "adx_unet_pack: expected at least %d parameter tensors, got %d", u->n_params,
              n_params);
  for (int i = 0; i < u->n_params; ++i) ADX_REQUIRE(P[i] != nullptr, "adx_unet_pack: parameter %d is null", i);
  hipStream_t s = (hipStream_t)stream;
  float* base = (float*)packed;
  const int dim = u->cfg.dim;
  int rc = ADX_OK;
  PackQueueScope pack_scope;   // every weight image of the stack in one table-driven launch (tconv_pack.h), flushed below
  for (auto& b : u->blocks) {
    if (rc == ADX_OK) rc = pack_layer(b.a, P, base, s);
    if (rc == ADX_OK) rc = pack_layer(b.b, P, base, s, (b.has_r && b.b.chained) ? &b.r : nullptr);
    if (rc == ADX_OK && b.has_r) rc = pack_layer(b.r, P, base, s);
    // concatenate the block's time_mlp Linear into the fused [sum_c][2 dim] matrix
    if (rc == ADX_OK) rc = copy_f(base + u->o_tlin_raw + (size_t)b.tb_off * 2 * dim, P[b.p_tw], (size_t)b.cout * 2 * dim, s);
    if (rc == ADX_OK) rc = copy_f(base + u->o_tlin_b + b.tb_off, P[b.p_tb], b.cout, s);
  }
  for (auto& l : u->downs) if (rc == ADX_OK) rc = pack_layer(l, P, base, s);
  for (auto& l : u->ups) if (rc == ADX_OK) rc = pack_layer(l, P, base, s);
  if (rc == ADX_OK) rc = pack_layer(u->head0, P, base, s);
  if (rc == ADX_OK) rc = pack_layer(u->head1, P, base, s);
  // the chains' parameter blocks: [bias | gamma | beta | residual bias] x cout_pad per stage (zero where a stage has none)
  for (const std::vector<ChainPlan>* cps : {&u->down_chains, &u->up_chains})
    for (const ChainPlan& cp : *cps) {
      if (!cp.valid || rc != ADX_OK) continue;
      batch_fill_add(base + cp.o_par, cp.par_floats);
      rc = batch_fill_flush(s);            // stream order: the zeros land before the copies queued below
      for (const ChainStagePlan& sp : cp.st) {
        float* dst = base + cp.o_par + sp.par_off;
        const int cpad = round_up(sp.L->d.cout, 16), co = sp.L->d.cout;
        if (rc == ADX_OK && sp.L->p_b >= 0) rc = copy_f(dst, P[sp.L->p_b], co, s);
        if (rc == ADX_OK && sp.L->p_g >= 0) rc = copy_f(dst + cpad, P[sp.L->p_g], co, s);
        if (rc == ADX_OK && sp.L->p_be >= 0) rc = copy_f(dst + 2 * cpad, P[sp.L->p_be], co, s);
        if (rc == ADX_OK && sp.R != nullptr && sp.R->p_b >= 0) rc = copy_f(dst + 3 * cpad, P[sp.R->p_b], co, s);
      }
    }
  if (rc == ADX_OK) rc = batch_copy_flush(s);      // the fused Linear below is packed from the concatenated copy
  if (rc == ADX_OK) rc = tconv_pack(&u->tlin.d, base + u->o_tlin_raw, base + u->tlin.o_w, s);
  {
    const int rp = pack_flush(s);                  // behind the copies above (the fused Linear reads one); always closes the queue
    if (rc == ADX_OK) rc = rp;
  }
  if (rc == ADX_OK && u->pipe_ok) {
    // the pipeline run's seven images (tconv_pipe.hip), re-laid from the K-split images the flush above wrote: one launch.  They are
    // part of THIS packed buffer's contents like every other image -- a forward never writes `packed`, and a caller may keep
    // several packed buffers (live and EMA weights) on one handle
    const int n = u->cfg.n_mults;
    ResBlock& b0 = u->blocks[2 * (n - 1)];
    ResBlock& b1 = u->blocks[2 * (n - 1) + 1];
    ResBlock& m1 = u->blocks[2 * n];
    ResBlock& m2 = u->blocks[2 * n + 1];
    const ConvLayer* run[7] = {&b0.b, &b1.a, &b1.b, &m1.a, &m1.b, &m2.a, &m2.b};
    const float* src[7];
    float* dst[7];
    for (int k = 0; k < 7; ++k) {
      ADX_REQUIRE(run[k]->piped && tconv_hs_kernel_image(&run[k]->d), "adx_unet_pack: piped layer without a K-split weight image");
      src[k] = base + run[k]->o_w;
      dst[k] = base + run[k]->o_pw;
    }
    const adx_tconv_desc& d0 = b0.b.d;
    rc = pipe_repack_from_hs_many(src, dst, 7, d0.cout, d0.taps, d0.pad, d0.lin, s);
  }
  if (rc == ADX_OK) rc = copy_f(base + u->o_freqs, freqs, dim / 2, s);
  if (rc == ADX_OK) rc = copy_f(base + u->o_t1w, P[u->p_t1w], (size_t)4 * dim * dim, s);
  if (rc == ADX_OK) rc = copy_f(base + u->o_t1b, P[u->p_t1b], 4 * dim, s);
  if (rc == ADX_OK) rc = copy_f(base + u->o_t3w, P[u->p_t3w], (size_t)4 * dim * dim, s);
  if (rc == ADX_OK) rc = copy_f(base + u->o_t3b, P[u->p_t3b], dim, s);
  if (rc == ADX_OK && u->cfg.guidance == 1) {
    rc = copy_f(base + u->o_c0w, P[u->p_c0w], 2 * dim, s);
    if (rc == ADX_OK) rc = copy_f(base + u->o_c0b, P[u->p_c0b], dim, s);
    if (rc == ADX_OK) rc = copy_f(base + u->o_c2w, P[u->p_c2w], (size_t)dim * dim, s);
    if (rc == ADX_OK) rc = copy_f(base + u->o_c2b, P[u->p_c2b], dim, s);
  }
  {
    const int rf = batch_copy_flush(s);            // always drain the queue, also after an error
    if (rc == ADX_OK) rc = rf;
  }
  if (rc == ADX_OK) u->packed_once = true;
  return rc;
}

constexpr int kRing = 6;

// workspace: time_embed, mish_cond, time-bias matrix, then kRing rotating activation buffers
// and one skip per level, each rows*dim*horizon floats
// (C*L is the same at every level: channels double as the length halves).
static size_t act_floats(const adx_unet* u, int rows) {
  size_t m = 0;
  for (auto& b : u->blocks) m = std::max(m, (size_t)b.cout * b.len);
  for (auto& l : u->ups) m = std::max(m, (size_t)l.d.cout * l.d.lout);
  m = std::max(m, (size_t)u->head0.d.cout * u->head0.d.lout);
  return align64(m * rows);
}

size_t adx_unet_workspace_bytes(const adx_unet* u, int32_t rows) {
  if (!u || rows < 1) return 0;
  const int dim = u->cfg.dim;
  size_t f = align64((size_t)rows * dim) + align64((size_t)rows * 2 * dim) + align64((size_t)rows * u->sum_c);
  f += act_floats(u, rows) * (size_t)(kRing + u->n_levels);
  f += kSplitScratchFloats + kTicketWords;
  return f * sizeof(float);
}

int adx_unet_forward(adx_unet* u, const void* packed, void* workspace, const adx_unet_io* io, adx_stream stream) {
  ADX_REQUIRE(u && packed && workspace && io, "adx_unet_forward: null argument");
  if (!u->packed_once) {
    set_error("adx_unet_forward: weights were never packed (call adx_unet_pack first)");
    return ADX_ERR_STATE;
  }
  ADX_REQUIRE(io->x && io->out, "adx_unet_forward: null tensor");
  if (pipe_fault_take() != 0) {
    set_error("adx_unet_forward: a stage of an earlier pipeline launch (tconv_pipe) timed out waiting for its producers; that run's "
              "output was set to NaN");
    return ADX_ERR_STATE;
  }
  const int rows = io->rows, dim = u->cfg.dim, H = u->cfg.horizon, D = u->cfg.transition_dim;
  ADX_REQUIRE(rows >= 1, "adx_unet_forward: rows must be >= 1");
  if (io->time_bias == nullptr) {
    const int rc0 = check_conditioning_io(u, io, "adx_unet_forward");
    if (rc0 != ADX_OK) return rc0;
  } else {
    ADX_REQUIRE(io->time_embed == nullptr, "adx_unet_forward: with a precomputed time_bias the caller already holds time_embed");
  }
  const int x_rows = io->x_rows > 0 ? io->x_rows : rows;
  ADX_REQUIRE(x_rows == rows || x_rows == 1, "adx_unet_forward: x_rows must be rows (%d) or 1, got %d", rows, x_rows);
  hipStream_t s = (hipStream_t)stream;
  const float* base = (const float*)packed;
  float* ws = (float*)workspace;
  size_t off = 0;
  auto take = [&](size_t n) { float* p = ws + off; off += align64(n); return p; };
  // the ticket words come first: their place does not depend on `rows`.  Every forward CLEARS them before their first user
  // runs -- nothing rests on what the caller's buffer held (uninitialised memory, a launch that never finished): where the
  // forward opens with a chained level that launch's workgroup 0 does it (ChainArgs::zero_words, no extra node in a captured
  // step), otherwise a 1 KB memset node goes first
  uint32_t* const split_tickets = reinterpret_cast<uint32_t*>(take(kTicketWords));
  float* te = take((size_t)rows * dim);
  float* mc = take((size_t)rows * 2 * dim);
  float* tb = take((size_t)rows * u->sum_c);
  const size_t af = act_floats(u, rows);
  float* bufs[kRing];
  for (auto& b : bufs) b = take(af);
  std::vector<float*> skips(u->n_levels);
  for (auto& p : skips) p = take(af);
  struct ScratchScope {   // handed to every conv of this call through make_io
    ScratchScope(float* p, uint32_t* t) { t_split_scratch = p; t_split_tickets = t; }
    ~ScratchScope() { t_split_scratch = nullptr; t_split_tickets = nullptr; }
  };
  float* const split_scratch = take(kSplitScratchFloats);
  constexpr bool tickets_on = true;
  ScratchScope scratch_scope(split_scratch, tickets_on ? split_tickets : nullptr);
  bool tickets_pending = tickets_on;      // still to be cleared by this call
  if (tickets_pending && !(io->time_bias != nullptr && u->down_chains[0].valid)) {
    const int rc0 = pipe_tickets_reset(split_tickets, (int)kTicketWords, kEpochSlot, s);     // zeroes them and draws this forward's number
    if (rc0 != ADX_OK) return rc0;
    tickets_pending = false;
  }

  int rc = ADX_OK;
  if (io->time_bias != nullptr) {
    tb = const_cast<float*>(io->time_bias);     // one step's rows of adx_unet_time_conditioning's table (read only)
  } else {
    rc = time_conditioning(u, base, io, rows, te, mc, tb, s);
    if (rc != ADX_OK) return rc;
  }

  // x arrives as [rows][H][D]; the UNet works on [rows][D][H] (temporal.py:204): read with strides
  // x_rows == 1: every row reads the one trajectory (the CFG pair torch.cat([x, x]) of interact.py:131, never built)
  Act cur{io->x, x_rows == rows ? (int64_t)H * D : 0, 1, (int64_t)D};
  // Six activation buffers used strictly round-robin.  A buffer is overwritten six takes after it
  // was handed out; the longest any tensor stays live is four takes (a block input is read by
  // the last conv of the block, after h and the 1x1-residual buffers of that block were taken).
  int nb = 0;
  auto next_buf = [&]() { float* p = bufs[nb]; nb = (nb + 1) % kRing; return p; };
  // R(x) of a block beside its first conv on a side stream was measured and removed (round 2: fork/join edges cost more
  // than the overlap gains, 1664 -> 1244 steps/s eager, 1603 -> 1423 as a graph): everything is issued in stream order.
  auto run_block = [&](const ResBlock& B, const Act& x0, const Act* x1, float* dst) -> int {
    float* h = next_buf();
    Act res = x0;  // identity residual (cin == cout, never a concat)
    if (B.has_r) {
      // R(x) and block[0] read the same input and nothing of each other: ONE launch where both run on the short-K
      // kernel (tconv_hs_forward_pair), two otherwise
      float* rb = next_buf();
      const adx_tconv_io io_a = make_io(B.a, base, x0, x1, tb + B.tb_off, u->sum_c, nullptr, h, (int64_t)B.cout * B.len, B.len, 1, rows);
      const adx_tconv_io io_r = make_io(B.r, base, x0, x1, nullptr, 0, nullptr, rb, (int64_t)B.cout * B.len, B.len, 1, rows);
      const int r = tconv_hs_forward_pair(&B.a.d, &io_a, &B.r.d, &io_r, s);
      if (r != ADX_OK) return r;
      res = dense(rb, B.cout, B.len);
    } else {
      const int r = run_conv(B.a, base, x0, x1, tb + B.tb_off, u->sum_c, nullptr, h, (int64_t)B.cout * B.len, B.len, 1, rows, s);
      if (r != ADX_OK) return r;
    }
    const Act hin = dense(h, B.cout, B.len);
    return run_conv(B.b, base, hin, nullptr, nullptr, 0, &res, dst, (int64_t)B.cout * B.len, B.len, 1, rows, s);
  };

  // one launch for a whole level (tconv_chain.hip) where the plan allows it
  constexpr int chain_rows = 0;     // 16 / 32 pin the rows per workgroup (measured: the rule below wins)
  auto run_chain = [&](const ChainPlan& cp, const Act& in0, const Act* in1, const Act& o0, const Act* o1) -> int {
    // samples per workgroup: as many as keep its rows (bt x the chain's longest length) within 32 where that still gives
    // the chip ~a hundred workgroups, else the fewest that fill a 16-row tile (more, smaller workgroups)
    const int bt_min = std::max(1, 16 / cp.len), bt_max = std::max(bt_min, 32 / cp.max_len);
    int bt = (chain_rows == 16 || ceil_div(rows, bt_max) < 96) ? bt_min : bt_max;
    if (chain_rows == 32) bt = bt_max;
    ChainArgs a;
    size_t lds = chain_args(cp, bt, a);
    if (lds == 0 || lds > kChainMaxLds) {
      bt = bt_min;
      lds = chain_args(cp, bt, a);
    }
    a.packed = base; a.tb = tb; a.tb_stride = u->sum_c;
    a.in0 = in0.p; a.in0_sb = in0.sb; a.in0_sc = in0.sc; a.in0_sl = in0.sl;
    if (in1 != nullptr) { a.in1 = in1->p; a.in1_sb = in1->sb; a.in1_sc = in1->sc; a.in1_sl = in1->sl; }
    auto dense4 = [](const Act& t) {
      return t.sl == 1 && t.sc % 4 == 0 && t.sb % 4 == 0 && (reinterpret_cast<uintptr_t>(t.p) & 15) == 0;
    };
    a.in_vec = cp.len % 4 == 0 && dense4(in0) && (in1 == nullptr || dense4(*in1));
    const Act* outs[2] = {&o0, o1};
    for (int k = 0; k < 2; ++k)
      if (outs[k] != nullptr) a.out[k] = ChainOut{const_cast<float*>(outs[k]->p), outs[k]->sb, outs[k]->sc, outs[k]->sl};
    for (int k = 0; k < a.n_stages; ++k)
      if ((a.st[k].flags & kChOut) && dense4(*outs[(a.st[k].flags >> 12) & 1])) a.st[k].flags |= kChOutVec;
    a.batch = rows;
    if (tickets_pending) {
      a.zero_words = split_tickets; a.n_zero = (int)kTicketWords;
      a.epoch_ctr = pipe_epoch_counter(false); a.epoch_slot = kEpochSlot;
      tickets_pending = false;
    }
    return chain_launch(a, ceil_div(rows, bt), lds, s);
  };

  size_t bi = 0;
  const int n = u->n_levels;
  bool pipe_done = false;
  for (int i = 0; i < n; ++i) {
    const ResBlock& B0 = u->blocks[bi++];
    const ResBlock& B1 = u->blocks[bi++];
    if (u->down_chains[i].valid) {
      const Act skip = dense(skips[i], B1.cout, B1.len);
      if (i < n - 1) {
        const ConvLayer& dn = u->downs[i];
        const Act y = dense(next_buf(), dn.d.cout, dn.d.lout);
        rc = run_chain(u->down_chains[i], cur, nullptr, skip, &y);
        cur = y;
      } else {
        rc = run_chain(u->down_chains[i], cur, nullptr, skip, nullptr);
        cur = skip;
      }
      if (rc != ADX_OK) return rc;
      continue;
    }
    if (i == n - 1 && u->pipe_ok && tickets_on && split_scratch != nullptr && pipe_epoch_counter(false) != nullptr &&
        pipe_shape_ok(B0.cout, B0.len, rows, B0.b.d.taps, B0.b.d.pad, B0.b.d.groups)) {
      // Small batch: block 0's second conv, block 1 and both mid blocks -- seven same-shaped convs -- as ONE pipeline launch
      // (tconv_pipe.hip).  Block 0's first conv + 1x1 residual conv stay the pair / mixed launch they were.
      const int C = B0.cout, Lp = B0.len;
      float* h = next_buf();
      float* rb = next_buf();
      {
        const adx_tconv_io io_a = make_io(B0.a, base, cur, nullptr, tb + B0.tb_off, u->sum_c, nullptr, h, (int64_t)C * Lp, Lp, 1, rows);
        const adx_tconv_io io_r = make_io(B0.r, base, cur, nullptr, nullptr, 0, nullptr, rb, (int64_t)C * Lp, Lp, 1, rows);
        rc = tconv_hs_forward_pair(&B0.a.d, &io_a, &B0.r.d, &io_r, s);
        if (rc != ADX_OK) return rc;
      }
      const ResBlock& M1 = u->blocks[bi];
      const ResBlock& M2 = u->blocks[bi + 1];
      float* out = next_buf();
      const ConvLayer* run[7] = {&B0.b, &B1.a, &B1.b, &M1.a, &M1.b, &M2.a, &M2.b};
      PipeArgs pa;
      memset(&pa, 0, sizeof(pa));
      pa.n_conv = 7; pa.C = C; pa.L = Lp; pa.rows = rows; pa.P = C / kPipeCh;
      pa.groups = B0.b.d.groups; pa.taps = B0.b.d.taps; pa.pad = B0.b.d.pad; pa.eps = B0.b.d.eps;
      pipe_live_taps(pa.taps, pa.pad, Lp, &pa.tap0, &pa.ntap);
      // the tail of this call's scratch: records of the seven stages, then the three block outputs only later residuals read
      const size_t rec_floats = pipe_record_floats(7, pa.P);
      ADX_REQUIRE(align64(rec_floats) + 3 * align64((size_t)rows * Lp * C) <= kPipeTailFloats, "adx_unet_forward: pipeline tail too small");
      pa.records = split_scratch + (kSplitScratchFloats - kPipeTailFloats);
      float* ya = pa.records + align64(rec_floats);
      float* yb = ya + align64((size_t)rows * Lp * C);
      float* yc = yb + align64((size_t)rows * Lp * C);
      pa.fault = pipe_fault_word();
      pa.epoch = split_tickets + kEpochSlot;                      // this forward's number, left there by the launch that cleared the tickets
      for (int k = 0; k < 7; ++k) {
        pa.st[k].w = base + run[k]->o_pw;
        pa.st[k].bias = base + run[k]->o_b;
      }
      for (int k = 1; k <= 7; ++k) {                               // stage k forms its input from conv k - 1's records
        pa.st[k].gamma = base + run[k - 1]->o_g;
        pa.st[k].beta = base + run[k - 1]->o_be;
      }
      pa.st[0].in = h;                                             // h0 = block0(x) + time bias, finished by the launch above
      // conv 0 = B0.b: y0 = act + R(x);       conv 1 = B1.a: h1 = act + tb(B1);   conv 2 = B1.b: y2 = act + y0 (the level's output, the skip)
      // conv 3 = M1.a: h3 = act + tb(M1);     conv 4 = M1.b: y4 = act + y2;      conv 5 = M2.a: h5 = act + tb(M2);  conv 6 = M2.b: y6 = act + y4
      auto resid = [&](int k, const float* p, int kind) { pa.st[k].add = p; pa.st[k].add_kind = kind; };
      auto tbias = [&](int k, const ResBlock& Bk) { pa.st[k].add = tb + Bk.tb_off; pa.st[k].add_stride = u->sum_c; pa.st[k].add_kind = 1; };
      auto publish = [&](int k, float* p, int kind) { pa.st[k].pub = p; pa.st[k].pub_kind = kind; };
      resid(1, rb, 2);       publish(1, ya, 2);        pa.st[1].add_early = 1;      // R(x): finished by the launch above
      tbias(2, B1);
      resid(3, ya, 3);       publish(3, skips[i], 1);  pa.st[3].pub2 = yc;          // the level's output: the skip, and y4's residual
      tbias(4, M1);
      resid(5, yc, 3);       publish(5, yb, 2);
      tbias(6, M2);
      resid(7, yb, 3);       publish(7, out, 1);
      rc = pipe_launch(pa, s);
      if (rc != ADX_OK) return rc;
      cur = dense(out, C, Lp);
      bi += 2;                // the mid blocks ran inside the pipeline
      pipe_done = true;
      continue;
    }
    float* y0 = next_buf();
    rc = run_block(B0, cur, nullptr, y0);
    if (rc != ADX_OK) return rc;
    const Act a0 = dense(y0, B0.cout, B0.len);
    rc = run_block(B1, a0, nullptr, skips[i]);  // the level output doubles as the skip (temporal.py:219)
    if (rc != ADX_OK) return rc;
    cur = dense(skips[i], B1.cout, B1.len);
    if (i < n - 1) {
      const ConvLayer& dn = u->downs[i];
      float* y = next_buf();
      rc = run_conv(dn, base, cur, nullptr, nullptr, 0, nullptr, y, (int64_t)dn.d.cout * dn.d.lout, dn.d.lout, 1, rows, s);
      if (rc != ADX_OK) return rc;
      cur = dense(y, dn.d.cout, dn.d.lout);
    }
  }
  for (int k = 0; k < 2 && !pipe_done; ++k) {  // mid_block1, mid_block2
    const ResBlock& B = u->blocks[bi++];
    float* y = next_buf();
    rc = run_block(B, cur, nullptr, y);
    if (rc != ADX_OK) return rc;
    cur = dense(y, B.cout, B.len);
  }
  bool head_done = false;
  for (int i = 0; i < n - 1; ++i) {
    const ResBlock& B0 = u->blocks[bi++];
    const ResBlock& B1 = u->blocks[bi++];
    // h.pop(): the deepest skip first; h[0] is pushed but never popped (temporal.py:226-227)
    const Act skip = dense(skips[n - 1 - i], B0.c1, B0.len);
    if (u->up_chains[i].valid) {
      const ChainPlan& cp = u->up_chains[i];
      if (cp.with_head) {      // ... Upsample1d, final_conv: the chain writes the model output itself
        const Act o{io->out, (int64_t)H * u->out_ch, 1, (int64_t)u->out_ch};
        rc = run_chain(cp, cur, &skip, o, nullptr);
        head_done = true;
      } else {
        const ConvLayer& up = u->ups[i];
        const Act y = dense(next_buf(), up.d.cout, up.d.lout);
        rc = run_chain(cp, cur, &skip, y, nullptr);
        cur = y;
      }
      if (rc != ADX_OK) return rc;
      continue;
    }
    float* y0 = next_buf();
    rc = run_block(B0, cur, &skip, y0);
    if (rc != ADX_OK) return rc;
    const Act a0 = dense(y0, B0.cout, B0.len);
    float* y1 = next_buf();
    rc = run_block(B1, a0, nullptr, y1);
    if (rc != ADX_OK) return rc;
    const Act a1 = dense(y1, B1.cout, B1.len);
    const ConvLayer& up = u->ups[i];
    float* y2 = next_buf();
    rc = run_conv(up, base, a1, nullptr, nullptr, 0, nullptr, y2, (int64_t)up.d.cout * up.d.lout, up.d.lout, 1, rows, s);
    if (rc != ADX_OK) return rc;
    cur = dense(y2, up.d.cout, up.d.lout);
  }
  if (!head_done) {  // final_conv / act_conv: Conv1dBlock + 1x1, written back as [rows][H][out_ch] (temporal.py:233-235,243-244)
    float* y = next_buf();
    const ConvLayer& h0 = u->head0;
    rc = run_conv(h0, base, cur, nullptr, nullptr, 0, nullptr, y, (int64_t)h0.d.cout * h0.d.lout, h0.d.lout, 1, rows, s);
    if (rc != ADX_OK) return rc;
    const Act a = dense(y, h0.d.cout, h0.d.lout);
    rc = run_conv(u->head1, base, a, nullptr, nullptr, 0, nullptr, io->out, (int64_t)H * u->out_ch, 1, u->out_ch, rows, s);
    if (rc != ADX_OK) return rc;
  }
  if (io->time_embed != nullptr) {
    ADX_CHECK_HIP(hipMemcpyAsync(io->time_embed, te, (size_t)rows * dim * sizeof(float), hipMemcpyDeviceToDevice, s));
  }
  (void)D;
  return ADX_OK;
}

int32_t adx_unet_time_bias_width(const adx_unet* u) { return u ? u->sum_c : 0; }

size_t adx_unet_time_conditioning_workspace_bytes(const adx_unet* u, int32_t rows) {
  if (!u || rows < 1) return 0;
  const int dim = u->cfg.dim;        // time_embed + Mish(cond) rows only: no activation ring
  return (align64((size_t)rows * dim) + align64((size_t)rows * 2 * dim)) * sizeof(float);
}

int adx_unet_time_conditioning(adx_unet* u, const void* packed, void* workspace, const adx_unet_io* io, float* time_embed,
                               float* time_bias, adx_stream stream) {
  ADX_REQUIRE(u && packed && workspace && io && time_bias, "adx_unet_time_conditioning: null argument");
  if (!u->packed_once) {
    set_error("adx_unet_time_conditioning: weights were never packed (call adx_unet_pack first)");
    return ADX_ERR_STATE;
  }
  const int rows = io->rows, dim = u->cfg.dim;
  ADX_REQUIRE(rows >= 1, "adx_unet_time_conditioning: rows must be >= 1");
  int rc = check_conditioning_io(u, io, "adx_unet_time_conditioning");
  if (rc != ADX_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  float* ws = (float*)workspace;                    // te, mc (adx_unet_time_conditioning_workspace_bytes)
  float* te = ws;
  float* mc = ws + align64((size_t)rows * dim);
  rc = time_conditioning(u, (const float*)packed, io, rows, te, mc, time_bias, s);
  if (rc != ADX_OK) return rc;
  if (time_embed != nullptr)
    ADX_CHECK_HIP(hipMemcpyAsync(time_embed, te, (size_t)rows * dim * sizeof(float), hipMemcpyDeviceToDevice, s));
  return ADX_OK;
}

}  // extern "C"
