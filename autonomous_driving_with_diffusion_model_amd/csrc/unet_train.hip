// Training executor for the temporal stack: forward that keeps what the backward pass needs (every
// activation, the pre-GroupNorm conv outputs and the group statistics) on a tape, and a backward
// that walks the tape in reverse (T1, train.py:242-251; reference = torch autograd).
//
// Per taped conv launch  y = [Mish(GN(.))](conv(x0|x1) + b) [+ tb] [+ res]  the backward issues
//   gn_mish_bwd   dy -> dc, d gamma, d beta, d bias, d tb-slice      (Conv1dBlocks only)
//   tconv_wgrad   (x, dc) -> dW                                      (MFMA, batch-split + atomics)
//   tconv_forward dc -> dx with the same weight re-read via w_layout/w_flip (the data gradient of a
//                 conv is a conv; of a strided conv a transposed conv and vice versa)
// and routes dy to the residual operand.  Gradients of tensors with several consumers (block
// inputs, skips) are accumulated through the conv epilogue's `res` input, so no extra passes.
#include <algorithm>
#include <unordered_map>
#include <vector>

#include "batch_ops.h"
#include "unet_internal.h"
#include "tconv_pack.h"

namespace adx {

int gn_mish_backward_raw(const float* dy, int64_t sb, int64_t sc, int64_t sl, const float* pre, const float* stats,
                         const float* gamma, const float* beta, float* dc, float* dgamma, float* dbeta, float* dbias,
                         float* dtb, int64_t dtb_stride, int B, int C, int L, int groups, hipStream_t s, int L_valid = 0);
int tconv_wgrad(const adx_tconv_desc* d, const adx_tconv_io* io, const float* dc, float* dw, hipStream_t s, bool zero);
int bias_grad(const float* dc, int64_t sb, int64_t sc, int64_t sl, float* db, int B, int C, int L, hipStream_t s, int L_valid = 0);
int add_strided(float* dst, const float* src, int64_t sb, int64_t sc, int64_t sl, int B, int C, int L, hipStream_t s, int L_valid = 0);
int embed_backward(const adx_embed_weights* w, int dim, const int64_t* t, int t_rows, const float* cond,
                   const float* feat, int feat_rows, int rows, const float* time_embed, const float* d_mish_cond,
                   const float* d_time_embed_extra, float* d_feat, float* const* grads /* w1,b1,w3,b3,cw0,cb0,cw2,cb2 */,
                   hipStream_t s);

struct TAct {  // activation view [rows][c][len]
  const float* p = nullptr;
  int64_t sb = 0, sc = 0, sl = 0;
  int c = 0, len = 0;
  bool dense() const { return sl == 1 && sc == len && sb == (int64_t)c * len; }
};

struct TapeOp {
  const ConvLayer* L = nullptr;
  TAct x0, x1, res, y;
  bool has_x1 = false, has_res = false;
  float* pre = nullptr;
  float* stats = nullptr;
  int tb_off = -1;
  bool need_dx = true;
};

}  // namespace adx

struct adx_unet_tape {
  std::vector<adx::TapeOp> ops;
  int rows = 0, t_rows = 0, feat_rows = 0;
  const float* img_feature = nullptr;
  const int64_t* t = nullptr;
  const float* cond = nullptr;
  float* te = nullptr; float* mc = nullptr; float* tb = nullptr;
  const float* x_in = nullptr;
  float* out = nullptr;
  size_t fwd_floats = 0;     // workspace floats consumed by the forward
};

namespace adx {

static size_t align64t(size_t v) { return (v + 63) / 64 * 64; }

struct Bump {
  float* base;
  size_t off, cap;
  bool ok = true;
  float* take(size_t n) {
    const size_t o = off;
    off = align64t(off + n);
    if (off > cap) { ok = false; return base; }
    return base + o;
  }
};

static size_t max_act(const adx_unet* u, int rows) {
  size_t m = 0;
  for (auto& b : u->blocks) m = std::max(m, (size_t)(b.c0 + b.c1) * b.len), m = std::max(m, (size_t)b.cout * b.len);
  for (auto& l : u->ups) m = std::max(m, (size_t)l.d.cout * l.d.lout);
  m = std::max(m, (size_t)u->head0.d.cout * u->head0.d.lout);
  return align64t(m * rows);
}

static void fill_w(adx_tconv_io& io, const ConvLayer& L, const float* base) {
  io.packed_w = base + L.o_w;
  io.bias = L.p_b >= 0 ? base + L.o_b : nullptr;
  io.gamma = L.p_g >= 0 ? base + L.o_g : nullptr;
  io.beta = L.p_be >= 0 ? base + L.o_be : nullptr;
}

}  // namespace adx

using namespace adx;

extern "C" {

// Workspace: forward tape (activations, pre-GN, stats) + backward gradients and scratch.
size_t adx_unet_train_workspace_bytes(const adx_unet* u, int32_t rows) {
  if (!u || rows < 1) return 0;
  const size_t a = max_act(u, rows);
  const size_t n_conv = u->blocks.size() * 3 + u->downs.size() + u->ups.size() + 2;
  size_t f = a * (n_conv * 4 + 16);                                   // y, pre, grad, dc/cat scratch per launch
  f += align64t((size_t)rows * u->sum_c) * 2 + align64t((size_t)rows * 2 * u->cfg.dim) * 2 + align64t((size_t)rows * u->cfg.dim) * 2;
  f += n_conv * align64t((size_t)rows * 8 * 2);                       // GN stats
  f += align64t((size_t)u->sum_c * 2 * u->cfg.dim) + 2 * align64t(tconv_packed_floats(&u->blocks[u->blocks.size() / 2].a.d) * 2);
  // data-gradient weight images: one slot per conv (they are all re-laid in ONE launch before the backward loop); a gradient's
  // image has the roles of c0 + c1 and cout swapped, which the padded sizes bound by the forward image's x 2
  size_t wall = 0;
  auto img = [&](const adx_tconv_desc& d) { wall += align64t(tconv_packed_floats(&d) * 2); };
  for (auto& b : u->blocks) { img(b.a.d); img(b.b.d); if (b.has_r) img(b.r.d); }
  for (auto& l : u->downs) img(l.d);
  for (auto& l : u->ups) img(l.d);
  img(u->head0.d); img(u->head1.d);
  f += wall + align64t((size_t)u->sum_c * 2 * u->cfg.dim * 2);
  return f * sizeof(float);
}

int adx_unet_tape_create(adx_unet_tape** out) {
  ADX_REQUIRE(out != nullptr, "adx_unet_tape_create: null argument");
  *out = new adx_unet_tape();
  return ADX_OK;
}
void adx_unet_tape_destroy(adx_unet_tape* t) { delete t; }

int adx_unet_forward_train(adx_unet* u, const void* packed, void* workspace, size_t workspace_bytes,
                           const adx_unet_io* io, adx_unet_tape* tape, adx_stream stream) {
  ADX_REQUIRE(u && packed && workspace && io && tape, "adx_unet_forward_train: null argument");
  if (!u->packed_once) {
    set_error("adx_unet_forward_train: weights were never packed (call adx_unet_pack first)");
    return ADX_ERR_STATE;
  }
  ADX_REQUIRE(io->x && io->img_feature && io->t && io->out, "adx_unet_forward_train: null tensor");
  {
    // the backward kernels (tbwd.hip) tile like the MFMA forward kernels: a model with layers only the general-shape
    // kernel covers (GroupNorm widths that are not powers of two, groups of fewer than 64 elements) samples but does not train
    auto trainable = [](const adx_tconv_desc& d) { return tconv_exact_supported(&d); };
    bool ok = trainable(u->head0.d) && trainable(u->head1.d);
    for (auto& b : u->blocks) ok = ok && trainable(b.a.d) && trainable(b.b.d) && (!b.has_r || trainable(b.r.d));
    for (auto& l : u->downs) ok = ok && trainable(l.d);
    for (auto& l : u->ups) ok = ok && trainable(l.d);
    ADX_REQUIRE(ok, "adx_unet_forward_train: this configuration has GroupNorm groups that are not a power of two wide or hold "
                    "fewer than 64 elements; such layers run in sampling only (csrc/tconv_generic.hip), training needs "
                    "MODEL.DIM in {32, 64, 128, ...} and a horizon of at least 16");
  }
  const int rows = io->rows, dim = u->cfg.dim, H = u->cfg.horizon, D = u->cfg.transition_dim;
  ADX_REQUIRE(rows >= 1 && io->t_rows == rows && io->feat_rows == rows,
              "adx_unet_forward_train: time / image batch must equal the trajectory batch (%d)", rows);
  hipStream_t s = (hipStream_t)stream;
  const float* base = (const float*)packed;
  Bump ws{(float*)workspace, 0, workspace_bytes / sizeof(float)};
  tape->ops.clear();
  tape->rows = rows; tape->t_rows = io->t_rows; tape->feat_rows = io->feat_rows;
  tape->img_feature = io->img_feature; tape->t = io->t; tape->cond = u->cfg.guidance == 1 ? io->cond : nullptr;
  tape->x_in = io->x; tape->out = io->out;
  float* te = ws.take((size_t)rows * dim);
  float* mc = ws.take((size_t)rows * 2 * dim);
  float* tb = ws.take((size_t)rows * u->sum_c);
  tape->te = te; tape->mc = mc; tape->tb = tb;

  adx_embed_weights ew;
  memset(&ew, 0, sizeof(ew));
  ew.freqs = base + u->o_freqs;
  ew.w1 = base + u->o_t1w; ew.b1 = base + u->o_t1b; ew.w3 = base + u->o_t3w; ew.b3 = base + u->o_t3b;
  if (u->cfg.guidance == 1) {
    ew.cw0 = base + u->o_c0w; ew.cb0 = base + u->o_c0b; ew.cw2 = base + u->o_c2w; ew.cb2 = base + u->o_c2b;
  }
  int rc = embed_forward(&ew, dim, io->t, io->t_rows, tape->cond, io->img_feature, io->feat_rows, rows, te, mc, s);
  if (rc != ADX_OK) return rc;
  {
    adx_tconv_io lio;
    memset(&lio, 0, sizeof(lio));
    lio.x0 = mc; lio.x0_sb = 2 * dim; lio.x0_sc = 1; lio.x0_sl = 0;
    lio.packed_w = base + u->tlin.o_w; lio.bias = base + u->o_tlin_b;
    lio.y = tb; lio.y_sb = u->sum_c; lio.y_sc = 1; lio.y_sl = 0;
    lio.batch = rows;
    rc = tconv_forward(&u->tlin.d, &lio, s);
    if (rc != ADX_OK) return rc;
  }

  auto dense_act = [](const float* p, int c, int len) {
    TAct a; a.p = p; a.sb = (int64_t)c * len; a.sc = len; a.sl = 1; a.c = c; a.len = len; return a;
  };
  // one taped conv launch
  auto conv = [&](const ConvLayer& L, const TAct& x0, const TAct* x1, int tb_off, const TAct* res, const TAct* yview,
                  bool need_dx) -> TAct {
    TapeOp op;
    op.L = &L; op.x0 = x0; op.need_dx = need_dx;
    if (x1) { op.x1 = *x1; op.has_x1 = true; }
    if (res) { op.res = *res; op.has_res = true; }
    op.tb_off = tb_off;
    TAct y = yview ? *yview : dense_act(ws.take((size_t)rows * L.d.cout * L.d.lout), L.d.cout, L.d.lout);
    op.y = y;
    adx_tconv_io cio;
    memset(&cio, 0, sizeof(cio));
    cio.x0 = x0.p; cio.x0_sb = x0.sb; cio.x0_sc = x0.sc; cio.x0_sl = x0.sl;
    if (x1) { cio.x1 = x1->p; cio.x1_sb = x1->sb; cio.x1_sc = x1->sc; cio.x1_sl = x1->sl; }
    fill_w(cio, L, base);
    if (tb_off >= 0) { cio.tbias = tb + tb_off; cio.tbias_stride = u->sum_c; }
    if (res) { cio.res = res->p; cio.res_sb = res->sb; cio.res_sc = res->sc; cio.res_sl = res->sl; }
    cio.y = const_cast<float*>(y.p); cio.y_sb = y.sb; cio.y_sc = y.sc; cio.y_sl = y.sl;
    cio.batch = rows;
    if (L.d.groups > 0) {
      op.pre = ws.take((size_t)rows * L.d.cout * L.d.lout);
      op.stats = ws.take((size_t)rows * L.d.groups * 2);
      cio.pre = op.pre; cio.stats = op.stats;
    }
    if (rc == ADX_OK && ws.ok) rc = tconv_forward(&L.d, &cio, s);
    tape->ops.push_back(op);
    return y;
  };
  auto block = [&](const ResBlock& B, const TAct& x0, const TAct* x1, bool x_needs_grad) -> TAct {
    TAct h = conv(B.a, x0, x1, B.tb_off, nullptr, nullptr, x_needs_grad);
    TAct res = x0;
    if (B.has_r) res = conv(B.r, x0, x1, -1, nullptr, nullptr, x_needs_grad);
    return conv(B.b, h, nullptr, -1, &res, nullptr, true);
  };

  TAct cur;
  cur.p = io->x; cur.sb = (int64_t)H * D; cur.sc = 1; cur.sl = D; cur.c = D; cur.len = H;
  const int n = u->n_levels;
  std::vector<TAct> skips(n);
  size_t bi = 0;
  for (int i = 0; i < n; ++i) {
    const TAct a0 = block(u->blocks[bi++], cur, nullptr, i > 0);  // the noisy trajectory needs no gradient
    cur = block(u->blocks[bi++], a0, nullptr, true);
    skips[i] = cur;
    if (i < n - 1) cur = conv(u->downs[i], cur, nullptr, -1, nullptr, nullptr, true);
  }
  for (int k = 0; k < 2; ++k) cur = block(u->blocks[bi++], cur, nullptr, true);
  for (int i = 0; i < n - 1; ++i) {
    const TAct a0 = block(u->blocks[bi++], cur, &skips[n - 1 - i], true);
    const TAct a1 = block(u->blocks[bi++], a0, nullptr, true);
    cur = conv(u->ups[i], a1, nullptr, -1, nullptr, nullptr, true);
  }
  const TAct hh = conv(u->head0, cur, nullptr, -1, nullptr, nullptr, true);
  TAct outv;
  outv.p = io->out; outv.sb = (int64_t)H * u->out_ch; outv.sc = 1; outv.sl = u->out_ch; outv.c = u->out_ch; outv.len = H;
  conv(u->head1, hh, nullptr, -1, nullptr, &outv, true);
  if (rc != ADX_OK) return rc;
  if (!ws.ok) {
    set_error("adx_unet_forward_train: workspace of %zu bytes too small", workspace_bytes);
    return ADX_ERR_INVALID;
  }
  tape->fwd_floats = ws.off;
  if (io->time_embed != nullptr)
    ADX_CHECK_HIP(hipMemcpyAsync(io->time_embed, te, (size_t)rows * dim * sizeof(float), hipMemcpyDeviceToDevice, s));
  return ADX_OK;
}

// grads: one pointer per parameter of adx_unet_pack's list (PyTorch layouts), written (not accumulated).
// d_out: gradient of the forward's `out` ([rows][H][out_ch]); d_time_embed: optional extra gradient w.r.t. the
// returned time_embed (CLASSIFIER_GUIDANCE: from TrajPredict); d_img_feature: [rows][dim] written.
int adx_unet_backward(adx_unet* u, const void* packed, void* workspace, size_t workspace_bytes, adx_unet_tape* tape,
                      const float* d_out, const float* d_time_embed, float* d_img_feature, const float* const* params,
                      float* const* grads, int32_t n_grads, adx_stream stream) {
  ADX_REQUIRE(u && packed && workspace && tape && d_out && d_img_feature && params && grads,
              "adx_unet_backward: null argument");
  ADX_REQUIRE(n_grads >= u->n_params, "adx_unet_backward: expected %d gradient tensors, got %d", u->n_params, n_grads);
  ADX_REQUIRE(!tape->ops.empty(), "adx_unet_backward: empty tape (run adx_unet_forward_train first)");
  hipStream_t s = (hipStream_t)stream;
  const float* base = (const float*)packed;
  const int rows = tape->rows, dim = u->cfg.dim;
  Bump ws{(float*)workspace, tape->fwd_floats, workspace_bytes / sizeof(float)};
  const size_t amax = max_act(u, rows);

  struct Slot { float* g = nullptr; bool has = false; };
  std::unordered_map<const float*, Slot> gmap;
  auto slot = [&](const TAct& a) -> Slot& {
    Slot& sl = gmap[a.p];
    if (sl.g == nullptr) sl.g = ws.take((size_t)rows * a.c * a.len);
    return sl;
  };
  // gradient of the model output arrives strided exactly like the output itself
  const TapeOp& last = tape->ops.back();
  TAct dlast = last.y;
  dlast.p = d_out;

  float* dtb = ws.take((size_t)rows * u->sum_c);
  float* dc_buf = ws.take(amax);
  float* cat_buf = ws.take(amax);
  // the data-gradient conv of a launch: the same conv with the channel roles swapped, on the exact-fp32 kernel
  auto dgrad_desc = [](const adx_tconv_desc& d) {
    adx_tconv_desc g{};
    g.groups = 0; g.eps = d.eps; g.c0 = d.cout; g.c1 = 0; g.cout = d.c0 + d.c1; g.lin = d.lout; g.lout = d.lin; g.taps = d.taps;
    g.lin_valid = d.lout_valid; g.lout_valid = d.lin_valid;
    g.exact = 1;   // gradients span many binades (1e-9 .. 1): keep them off the fp16 operand path
    if (d.kind == 0 && d.stride == 1) {
      g.kind = 0; g.stride = 1; g.pad = d.taps - 1 - d.pad; g.w_layout = 1; g.w_flip = 1;
    } else if (d.kind == 0) {       // strided conv -> transposed conv with the conv's own weight
      g.kind = 1; g.stride = d.stride; g.pad = d.pad; g.w_layout = 0; g.w_flip = 0;
    } else {                        // transposed conv -> strided conv with the transposed conv's own weight
      g.kind = 0; g.stride = d.stride; g.pad = d.pad; g.w_layout = 0; g.w_flip = 0;
    }
    return g;
  };
  // every data-gradient weight image in its own slot, all of them re-laid by ONE launch here (they were ~55 launches inside the
  // loop, each in front of the conv that reads it)
  std::vector<float*> gimg(tape->ops.size(), nullptr);
  adx_tconv_desc g_tlin{};
  g_tlin.kind = 0; g_tlin.taps = 1; g_tlin.stride = 1; g_tlin.pad = 0; g_tlin.c0 = u->sum_c; g_tlin.c1 = 0; g_tlin.cout = 2 * dim;
  g_tlin.lin = 1; g_tlin.lout = 1; g_tlin.groups = 0; g_tlin.eps = 1e-5f; g_tlin.w_layout = 1; g_tlin.w_flip = 0; g_tlin.exact = 1;
  float* tlin_img = nullptr;
  {
    PackQueueScope pack_scope;
    int rq = ADX_OK;
    for (size_t oi = 0; oi < tape->ops.size() && rq == ADX_OK; ++oi) {
      const TapeOp& op = tape->ops[oi];
      if (!op.need_dx) continue;
      const adx_tconv_desc g = dgrad_desc(op.L->d);
      gimg[oi] = ws.take(tconv_packed_floats(&g));
      if (!ws.ok) break;
      rq = tconv_pack(&g, params[op.L->p_w], gimg[oi], s);
    }
    tlin_img = ws.take(tconv_packed_floats(&g_tlin));
    if (ws.ok && rq == ADX_OK) rq = tconv_pack(&g_tlin, base + u->o_tlin_raw, tlin_img, s);
    const int rf = pack_flush(s);
    ADX_REQUIRE(ws.ok, "adx_unet_backward: workspace of %zu bytes too small", workspace_bytes);
    if (rq != ADX_OK) return rq;
    if (rf != ADX_OK) return rf;
  }

  // every small gradient tensor that is accumulated atomically (conv weights, GroupNorm affine, conv bias) is zeroed
  // here in a handful of launches instead of one memset each inside the loop
  for (const TapeOp& op : tape->ops) {
    const ConvLayer& L = *op.L;
    const adx_tconv_desc& d = L.d;
    batch_fill_add(grads[L.p_w], (size_t)d.cout * (d.c0 + d.c1) * d.taps);
    if (d.groups > 0) {
      batch_fill_add(grads[L.p_g], d.cout);
      batch_fill_add(grads[L.p_be], d.cout);
      batch_fill_add(grads[L.p_b], d.cout);
    }
  }
  int rc = batch_fill_flush(s);
  for (size_t oi = tape->ops.size(); oi-- > 0 && rc == ADX_OK;) {
    const TapeOp& op = tape->ops[oi];
    const ConvLayer& L = *op.L;
    const adx_tconv_desc& d = L.d;
    const int cin = d.c0 + d.c1;
    // real lengths (a horizon that is not a power of two runs on the next one: adx_tconv_desc::lin_valid); buffers keep the
    // padded pitch, every kernel below skips the positions that do not exist
    const int lov = d.lout_valid > 0 ? d.lout_valid : d.lout, liv = d.lin_valid > 0 ? d.lin_valid : d.lin;
    // ---- dy of this launch
    TAct dy;
    if (oi + 1 == tape->ops.size()) {
      dy = dlast;
    } else {
      auto it = gmap.find(op.y.p);
      ADX_REQUIRE(it != gmap.end() && it->second.has, "adx_unet_backward: launch %zu has no output gradient", oi);
      dy = op.y;
      dy.p = it->second.g;
      dy.sb = (int64_t)op.y.c * op.y.len; dy.sc = op.y.len; dy.sl = 1;   // gradient buffers are dense
    }
    // ---- residual operand receives dy unchanged
    if (op.has_res) {
      Slot& rs = slot(op.res);
      if (!rs.has) {
        ADX_CHECK_HIP(hipMemsetAsync(rs.g, 0, sizeof(float) * (size_t)rows * op.res.c * op.res.len, s));
        rs.has = true;
      }
      rc = add_strided(rs.g, dy.p, dy.sb, dy.sc, dy.sl, rows, op.res.c, op.res.len, s, lov);
      if (rc != ADX_OK) break;
    }
    // ---- through Mish / GroupNorm (and the time-bias add) down to the conv output
    const float* dc = nullptr;
    if (d.groups > 0) {
      float* dg = grads[L.p_g]; float* dbe = grads[L.p_be]; float* dbi = grads[L.p_b];
      rc = gn_mish_backward_raw(dy.p, dy.sb, dy.sc, dy.sl, op.pre, op.stats, base + L.o_g, base + L.o_be, dc_buf, dg,
                                dbe, dbi, op.tb_off >= 0 ? dtb + op.tb_off : nullptr, u->sum_c, rows, d.cout, d.lout,
                                d.groups, s, lov);
      if (rc != ADX_OK) break;
      dc = dc_buf;
    } else {
      if (dy.sl == 1 && dy.sc == d.lout && dy.sb == (int64_t)d.cout * d.lout) {
        dc = dy.p;
      } else {  // the head writes [rows][H][D]: make the dense [rows][D][H] copy the GEMMs expect
        ADX_CHECK_HIP(hipMemsetAsync(dc_buf, 0, sizeof(float) * (size_t)rows * d.cout * d.lout, s));
        rc = add_strided(dc_buf, dy.p, dy.sb, dy.sc, dy.sl, rows, d.cout, d.lout, s, lov);
        if (rc != ADX_OK) break;
        dc = dc_buf;
      }
      if (L.p_b >= 0) {
        rc = bias_grad(dc, (int64_t)d.cout * d.lout, d.lout, 1, grads[L.p_b], rows, d.cout, d.lout, s, lov);
        if (rc != ADX_OK) break;
      }
    }
    // ---- weight gradient
    {
      adx_tconv_io wio;
      memset(&wio, 0, sizeof(wio));
      wio.batch = rows;
      if (d.kind == 0) {
        wio.x0 = op.x0.p; wio.x0_sb = op.x0.sb; wio.x0_sc = op.x0.sc; wio.x0_sl = op.x0.sl;
        if (op.has_x1) { wio.x1 = op.x1.p; wio.x1_sb = op.x1.sb; wio.x1_sc = op.x1.sc; wio.x1_sl = op.x1.sl; }
        rc = tconv_wgrad(&d, &wio, dc, grads[L.p_w], s, false);
      } else {
        // ConvTranspose1d weight [cin][cout][k]: dW = wgrad of the mirrored strided conv with x and dy swapped
        adx_tconv_desc m{};
        m.kind = 0; m.taps = d.taps; m.stride = d.stride; m.pad = d.pad;
        m.c0 = d.cout; m.c1 = 0; m.cout = cin; m.lin = d.lout; m.lout = d.lin; m.groups = 0; m.eps = d.eps;
        m.lin_valid = d.lout_valid; m.lout_valid = d.lin_valid;
        wio.x0 = dc; wio.x0_sb = (int64_t)d.cout * d.lout; wio.x0_sc = d.lout; wio.x0_sl = 1;
        ADX_REQUIRE(op.x0.dense(), "adx_unet_backward: transposed conv input must be dense");
        rc = tconv_wgrad(&m, &wio, op.x0.p, grads[L.p_w], s, false);
      }
      if (rc != ADX_OK) break;
    }
    // ---- data gradient
    if (!op.need_dx) continue;
    const adx_tconv_desc g = dgrad_desc(d);        // its weight image was re-laid before the loop
    adx_tconv_io gio;
    memset(&gio, 0, sizeof(gio));
    gio.x0 = dc; gio.x0_sb = (int64_t)d.cout * d.lout; gio.x0_sc = d.lout; gio.x0_sl = 1;
    gio.packed_w = gimg[oi];
    gio.batch = rows;
    const int64_t xsb = (int64_t)cin * d.lin;
    if (!op.has_x1) {
      Slot& xs = slot(op.x0);
      gio.y = xs.g; gio.y_sb = xsb; gio.y_sc = d.lin; gio.y_sl = 1;
      if (xs.has) { gio.res = xs.g; gio.res_sb = xsb; gio.res_sc = d.lin; gio.res_sl = 1; }  // accumulate in place
      rc = tconv_forward(&g, &gio, s);
      xs.has = true;
    } else {
      gio.y = cat_buf; gio.y_sb = xsb; gio.y_sc = d.lin; gio.y_sl = 1;
      rc = tconv_forward(&g, &gio, s);
      const TAct* parts[2] = {&op.x0, &op.x1};
      int coff = 0;
      for (int k = 0; k < 2 && rc == ADX_OK; ++k) {
        Slot& xs = slot(*parts[k]);
        if (!xs.has) {
          ADX_CHECK_HIP(hipMemsetAsync(xs.g, 0, sizeof(float) * (size_t)rows * parts[k]->c * d.lin, s));
          xs.has = true;
        }
        rc = add_strided(xs.g, cat_buf + (size_t)coff * d.lin, xsb, d.lin, 1, rows, parts[k]->c, d.lin, s, liv);
        coff += parts[k]->c;
      }
    }
  }
  if (rc != ADX_OK) return rc;
  ADX_REQUIRE(ws.ok, "adx_unet_backward: workspace of %zu bytes too small", workspace_bytes);

  // ---- the fused block Linear: tb = mish_cond @ Wcat^T + bcat
  float* dwcat = ws.take((size_t)u->sum_c * 2 * dim);
  float* dbcat = ws.take(u->sum_c);
  float* dmc = ws.take((size_t)rows * 2 * dim);
  ADX_REQUIRE(ws.ok, "adx_unet_backward: workspace of %zu bytes too small", workspace_bytes);
  {
    adx_tconv_io wio;
    memset(&wio, 0, sizeof(wio));
    wio.x0 = tape->mc; wio.x0_sb = 2 * dim; wio.x0_sc = 1; wio.x0_sl = 0;
    wio.batch = rows;
    rc = tconv_wgrad(&u->tlin.d, &wio, dtb, dwcat, s, true);
    if (rc != ADX_OK) return rc;
    rc = bias_grad(dtb, u->sum_c, 1, 0, dbcat, rows, u->sum_c, 1, s);
    if (rc != ADX_OK) return rc;
    for (auto& b : u->blocks) {          // the blocks' slices of the concatenated gradients: one launch, not 32 copies
      batch_copy_add(grads[b.p_tw], dwcat + (size_t)b.tb_off * 2 * dim, (size_t)b.cout * 2 * dim);
      batch_copy_add(grads[b.p_tb], dbcat + b.tb_off, (size_t)b.cout);
    }
    rc = batch_copy_flush(s);
    if (rc != ADX_OK) return rc;
    const adx_tconv_desc g = g_tlin;               // image re-laid before the loop, with the others
    adx_tconv_io gio;
    memset(&gio, 0, sizeof(gio));
    gio.x0 = dtb; gio.x0_sb = u->sum_c; gio.x0_sc = 1; gio.x0_sl = 0;
    gio.packed_w = tlin_img;
    gio.y = dmc; gio.y_sb = 2 * dim; gio.y_sc = 1; gio.y_sl = 0;
    gio.batch = rows;
    rc = tconv_forward(&g, &gio, s);
    if (rc != ADX_OK) return rc;
  }
  // ---- embedding MLPs and the perception feature
  adx_embed_weights ew;
  memset(&ew, 0, sizeof(ew));
  ew.freqs = base + u->o_freqs;
  ew.w1 = base + u->o_t1w; ew.b1 = base + u->o_t1b; ew.w3 = base + u->o_t3w; ew.b3 = base + u->o_t3b;
  float* eg[8] = {grads[u->p_t1w], grads[u->p_t1b], grads[u->p_t3w], grads[u->p_t3b], nullptr, nullptr, nullptr, nullptr};
  if (u->cfg.guidance == 1) {
    ew.cw0 = base + u->o_c0w; ew.cb0 = base + u->o_c0b; ew.cw2 = base + u->o_c2w; ew.cb2 = base + u->o_c2b;
    eg[4] = grads[u->p_c0w]; eg[5] = grads[u->p_c0b]; eg[6] = grads[u->p_c2w]; eg[7] = grads[u->p_c2b];
  }
  return embed_backward(&ew, dim, tape->t, tape->t_rows, tape->cond, tape->img_feature, tape->feat_rows, rows, tape->te,
                        dmc, d_time_embed, d_img_feature, eg, s);
}

}  // extern "C"
