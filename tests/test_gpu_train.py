"""GPU parity, training row (T1): gradients of the temporal stack against torch autograd through the
CPU oracle on identical inputs.  Tolerance: per-tensor relative error of the gradient norm-difference
<= 2e-4 (fp32, atomically reduced weight gradients => summation order differs run to run)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import unet as U
from autonomous_driving_with_diffusion_model_amd.modeling.spec import unet_entries
from autonomous_driving_with_diffusion_model_amd.utils import procedural as P
from helpers import close, oracle_sd, uni

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rel_err(got, ref):
    return ((got.cpu() - ref).norm() / (ref.norm() + 1e-12)).item()


def test_gn_mish_backward_and_wgrad_ops():
    import ctypes as C
    from autonomous_driving_with_diffusion_model_amd import _lib as L, ops
    for (c0, c1, cout, Lh, B) in ((64, 0, 64, 32, 3), (256, 0, 512, 4, 6), (512, 512, 256, 4, 5), (7, 0, 64, 16, 2),
                                  (128, 0, 128, 8, 7)):
        name = f"bw.{c0}.{c1}.{cout}.{Lh}"
        x0 = uni(name + ".x0", (B, c0, Lh))
        x1 = uni(name + ".x1", (B, c1, Lh)) if c1 else None
        cin = c0 + c1
        w = uni(name + ".w", (cout, cin, 5), lo=-(3.0 / (5 * cin)) ** 0.5, hi=(3.0 / (5 * cin)) ** 0.5).requires_grad_()
        b = uni(name + ".b", (cout,), lo=-.1, hi=.1).requires_grad_()
        g = uni(name + ".g", (cout,), lo=.9, hi=1.1).requires_grad_()
        be = uni(name + ".be", (cout,), lo=-.1, hi=.1).requires_grad_()
        tb = uni(name + ".tb", (B, cout)).requires_grad_()
        xin = (x0 if x1 is None else torch.cat([x0, x1], 1)).requires_grad_()
        pre_ref = F.conv1d(xin, w, b, padding=2)
        y = F.mish(F.group_norm(pre_ref, 8, g, be, 1e-5)) + tb[:, :, None]
        dy = uni(name + ".dy", (B, cout, Lh))
        y.backward(dy)
        # forward with the training outputs
        d = L.TConvDesc(0, 5, 1, 2, c0, c1, cout, Lh, Lh, 8, 1e-5, 0, 0)
        packed = torch.empty(L.lib().adx_tconv_packed_bytes(C.byref(d)) // 4, device=DEV)
        wd, bd, gd, bed = (t.detach().to(DEV) for t in (w, b, g, be))
        s = L.stream_ptr(torch.device(DEV))
        L.check(L.lib().adx_tconv_pack(C.byref(d), wd.data_ptr(), packed.data_ptr(), s))
        x0d = x0.to(DEV)
        x1d = None if x1 is None else x1.to(DEV)
        yd = torch.empty((B, cout, Lh), device=DEV)
        pre = torch.empty((B, cout, Lh), device=DEV)
        stats = torch.empty((B, 8, 2), device=DEV)
        io = L.TConvIO()
        io.x0, io.x0_sb, io.x0_sc, io.x0_sl = x0d.data_ptr(), c0 * Lh, Lh, 1
        if x1d is not None:
            io.x1, io.x1_sb, io.x1_sc, io.x1_sl = x1d.data_ptr(), c1 * Lh, Lh, 1
        io.packed_w, io.bias, io.gamma, io.beta = packed.data_ptr(), bd.data_ptr(), gd.data_ptr(), bed.data_ptr()
        io.y, io.y_sb, io.y_sc, io.y_sl, io.batch = yd.data_ptr(), cout * Lh, Lh, 1, B
        io.pre, io.stats = pre.data_ptr(), stats.data_ptr()
        L.check(L.lib().adx_tconv_forward(C.byref(d), C.byref(io), s))
        close(pre.cpu(), pre_ref.detach(), 2e-5)
        # gn/mish backward
        dyd = dy.to(DEV)
        dc = torch.empty_like(pre)
        dg, dbe, dbi = (torch.zeros(cout, device=DEV) for _ in range(3))
        dtb = torch.empty((B, cout + 3), device=DEV)
        L.check(L.lib().adx_gn_mish_backward(dyd.data_ptr(), cout * Lh, Lh, 1, pre.data_ptr(), stats.data_ptr(),
                                             gd.data_ptr(), bed.data_ptr(), dc.data_ptr(), dg.data_ptr(), dbe.data_ptr(),
                                             dbi.data_ptr(), dtb.data_ptr(), cout + 3, B, cout, Lh, 8, s))
        assert rel_err(dg, g.grad) < 2e-4 and rel_err(dbe, be.grad) < 2e-4 and rel_err(dbi, b.grad) < 2e-4
        assert rel_err(dtb[:, :cout], tb.grad) < 2e-5
        # weight gradient
        dw = torch.empty_like(wd)
        L.check(L.lib().adx_tconv_wgrad(C.byref(d), C.byref(io), dc.data_ptr(), dw.data_ptr(), s))
        assert rel_err(dw, w.grad) < 2e-4, (name, rel_err(dw, w.grad))
        # data gradient = the same conv family with the weight re-read (flip + swapped roles)
        gdsc = L.TConvDesc(0, 5, 1, 2, cout, 0, cin, Lh, Lh, 0, 1e-5, 1, 1)
        gp = torch.empty(L.lib().adx_tconv_packed_bytes(C.byref(gdsc)) // 4, device=DEV)
        L.check(L.lib().adx_tconv_pack(C.byref(gdsc), wd.data_ptr(), gp.data_ptr(), s))
        dx = torch.empty((B, cin, Lh), device=DEV)
        gio = L.TConvIO()
        gio.x0, gio.x0_sb, gio.x0_sc, gio.x0_sl = dc.data_ptr(), cout * Lh, Lh, 1
        gio.packed_w = gp.data_ptr()
        gio.y, gio.y_sb, gio.y_sc, gio.y_sl, gio.batch = dx.data_ptr(), cin * Lh, Lh, 1, B
        L.check(L.lib().adx_tconv_forward(C.byref(gdsc), C.byref(gio), s))
        assert rel_err(dx, xin.grad) < 2e-4, (name, rel_err(dx, xin.grad))


def test_traj_predict_parameter_gradients_vs_oracle_autograd():
    """adx_trajpred_backward_params: every state_pred parameter, d(action) and d(time_embed) against torch
    autograd through the oracle's TrajPredict (dropout off on both sides)."""
    from test_gpu_model import make_model
    m, _ = make_model("CLASSIFIER_GUIDANCE", 16)
    m.train()
    m.state_pred.dropout_p = 0.0          # the masked variant is test_traj_predict_dropout_* below
    sd = oracle_sd("CLASSIFIER_GUIDANCE")
    keys = [k for k in sd if k.startswith("state_pred.")]
    for k in keys:
        sd[k].requires_grad_()
    for B, T in ((3, 15), (1, 31), (5, 7), (3, 32), (2, 47), (1, 63)):   # T >= 32: the 64-row kernels (scratch in global memory)
        for k in keys:
            sd[k].grad = None
        m.zero_grad()
        a = P._uniform("tp.a", 71 + T, (B, T, 3), -1.5, 1.5)
        te = P._uniform("tp.te", 72 + T, (B, 64), -1.0, 1.0)
        w = P._uniform("tp.w", 73 + T, (B, T, 4), -1.0, 1.0)
        a_ref, te_ref = a.clone().requires_grad_(), te.clone().requires_grad_()
        out_ref = U.traj_predict(sd, "state_pred.", a_ref, te_ref)
        (out_ref * w).sum().backward()
        a_d, te_d = a.to(DEV).requires_grad_(), te.to(DEV).requires_grad_()
        out = m.state_pred(a_d, te_d)
        close(out.detach().cpu(), out_ref.detach(), 2e-5)
        (out * w.to(DEV)).sum().backward()
        assert rel_err(a_d.grad, a_ref.grad) < 2e-4, rel_err(a_d.grad, a_ref.grad)
        assert rel_err(te_d.grad, te_ref.grad) < 2e-4, rel_err(te_d.grad, te_ref.grad)
        named = dict(m.named_parameters())
        worst = max(((rel_err(named[k].grad, sd[k].grad), k) for k in keys))
        assert worst[0] < 5e-4, (B, T, worst)


def _lowbias32(x):
    x = x.astype(np.uint64) & 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def _dropout_masks(seed, p, B, T):
    """The masks adx_trajpred_forward_train(dropout_p = p, seed) uses (csrc/trajpred.hip: Drop), as multipliers."""
    thresh = int(float(np.float32(p)) * 4294967296.0)
    scale = float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))
    lo, hi = seed & 0xFFFFFFFF, seed >> 32
    b = np.arange(B, dtype=np.uint64)
    base = _lowbias32(_lowbias32(lo ^ ((b * 0x9E3779B9) & 0xFFFFFFFF)) ^ hi)          # [B]
    TP, E, FF, NH = (32 if T < 32 else 64), 64, 256, 4      # attention-site indices use the kernel's row pitch
    masks = {}
    for li in range(2):
        for site, shape in ((0, (NH, T, T)), (1, (T, E)), (2, (T, FF)), (3, (T, E))):
            key = _lowbias32(base ^ (((4 * li + site) * 0x85EBCA6B) & 0xFFFFFFFF))   # [B]
            if site == 0:
                h, t, s_ = np.meshgrid(np.arange(NH), np.arange(T), np.arange(T), indexing="ij")
                idx = (h * TP + t) * TP + s_
            else:
                t, n = np.meshgrid(np.arange(T), np.arange(shape[1]), indexing="ij")
                idx = t * shape[1] + n
            r = _lowbias32(key[:, None] ^ idx.reshape(1, -1).astype(np.uint64))
            masks[(li, site)] = torch.from_numpy(np.where(r >= thresh, scale, 0.0).reshape((B,) + shape))
    return masks


@pytest.mark.parametrize("B,T", [(3, 15), (2, 31), (2, 40)])
def test_traj_predict_dropout_matches_oracle_with_the_same_masks(B, T):
    """Train-mode dropout (p = 0.1 like nn.TransformerEncoderLayer): the kernel's hash masks are rebuilt on the host
    and fed to the oracle; output, d(action), d(time_embed) and every parameter gradient must agree -- this pins
    the mask placement in the forward pass and in every backward formula."""
    from test_gpu_model import make_model
    m, _ = make_model("CLASSIFIER_GUIDANCE", 16)
    m.train()
    sp = m.state_pred
    assert sp.dropout_p == pytest.approx(0.1)
    sd = {k: v.double() for k, v in oracle_sd("CLASSIFIER_GUIDANCE").items() if v.is_floating_point()}
    keys = [k for k in sd if k.startswith("state_pred.")]
    for k in keys:
        sd[k].requires_grad_()
    a = P._uniform("tpd.a", 81 + T, (B, T, 3), -1.5, 1.5)
    te = P._uniform("tpd.te", 82 + T, (B, 64), -1.0, 1.0)
    w = P._uniform("tpd.w", 83 + T, (B, T, 4), -1.0, 1.0)
    torch.manual_seed(1234)
    sp._calls = 41
    probe = type(sp)._next_seed
    seed = probe(sp)            # the seed the next call would use ...
    sp._calls = 41              # ... rewound so that the real call draws the same one
    masks = _dropout_masks(seed, sp.dropout_p, B, T)
    kept = torch.cat([v.flatten() for v in masks.values()])
    assert 0.85 < (kept > 0).double().mean().item() < 0.95          # about 10 % dropped
    a_ref, te_ref = a.double().requires_grad_(), te.double().requires_grad_()
    out_ref = U.traj_predict(sd, "state_pred.", a_ref, te_ref, masks=masks)
    (out_ref * w.double()).sum().backward()
    a_d, te_d = a.to(DEV).requires_grad_(), te.to(DEV).requires_grad_()
    out = sp(a_d, te_d)
    close(out.detach().cpu(), out_ref.detach().float(), 3e-5)
    (out * w.to(DEV)).sum().backward()
    assert rel_err(a_d.grad, a_ref.grad.float()) < 2e-4
    assert rel_err(te_d.grad, te_ref.grad.float()) < 2e-4
    named = dict(m.named_parameters())
    worst = max(((rel_err(named[k].grad, sd[k].grad.float()), k) for k in keys))
    assert worst[0] < 5e-4, worst
    # a different call draws different masks; eval mode ignores dropout
    out2 = sp(a.to(DEV), te.to(DEV))
    assert (out2 - out.detach()).abs().max().item() > 1e-3
    m.eval()
    with torch.no_grad():
        e1, e2 = sp(a.to(DEV), te.to(DEV)), sp(a.to(DEV), te.to(DEV))
    assert torch.equal(e1, e2)


@pytest.mark.parametrize("use_cond,H,B", [("NO_GUIDANCE", 16, 2), ("FREE_GUIDANCE", 32, 5),
                                          ("CLASSIFIER_GUIDANCE", 16, 3),
                                          # horizons that are not powers of two (24 -> 24, 12, 6, 3): every backward kernel
                                          # runs on the padded length and skips the positions that do not exist
                                          ("NO_GUIDANCE", 24, 3), ("FREE_GUIDANCE", 40, 2), ("CLASSIFIER_GUIDANCE", 24, 2),
                                          ("NO_GUIDANCE", 56, 2),
                                          # train.py:236-242: with probability 0.3 a FREE_GUIDANCE batch trains with cond=None
                                          # (the null embedding cond_mlp(0) of modeling/temporal.py:207)
                                          ("FREE_GUIDANCE-drop", 32, 5), ("FREE_GUIDANCE-drop", 16, 1)])
def test_unet_gradients_vs_oracle_autograd(use_cond, H, B):
    from test_gpu_model import make_model
    use_cond, _, drop = use_cond.partition("-")
    m, _ = make_model(use_cond, H)
    m.train()
    if hasattr(m, "state_pred"):
        m.state_pred.dropout_p = 0.0
    sd = oracle_sd(use_cond)
    keys = [e.key for e in unet_entries(use_cond) if not e.is_buffer and not e.key.startswith("perception.")]
    for k in keys:
        sd[k].requires_grad_()
    d = P.synthetic_batch(B, H, image_hw=(32, 32), seed=51)
    feat = P._uniform("train.feat", 51, (B, 64), -2.0, 2.0)
    feat_ref = feat.clone().requires_grad_()
    cond = d["target"] if (use_cond == "FREE_GUIDANCE" and not drop) else None
    pred_ref = U.unet_forward(sd, d["trajs"], None, d["t"], cond, use_cond=use_cond, img_feature=feat_ref)
    loss_ref = F.mse_loss(pred_ref, d["noise"])
    loss_ref.backward()
    feat_d = feat.to(DEV).requires_grad_()
    pred = m.unet_forward_train(d["trajs"].to(DEV), feat_d, d["t"].to(DEV), None if cond is None else cond.to(DEV))
    close(pred.detach().cpu(), pred_ref.detach(), 1e-4)
    loss = F.mse_loss(pred, d["noise"].to(DEV))
    loss.backward()
    assert rel_err(feat_d.grad, feat_ref.grad) < 5e-4, rel_err(feat_d.grad, feat_ref.grad)
    named = dict(m.named_parameters())
    worst = max(((rel_err(named[k].grad, sd[k].grad), k) for k in keys))
    assert worst[0] < 1e-3, worst
    assert all(named[k].grad is not None for k in keys)
    if drop:
        # cond = zeros: d(cond_mlp.0.weight) = d(pre-activation) x 0 is an all-zero tensor that must still EXIST (DDP's reducer
        # waits for every parameter, accelerate's find_unused_parameters=False), the rest of cond_mlp sees the null embedding
        assert sd["cond_mlp.0.weight"].grad.abs().max().item() == 0.0
        assert named["cond_mlp.0.weight"].grad.abs().max().item() == 0.0
        for k in ("cond_mlp.0.bias", "cond_mlp.2.weight", "cond_mlp.2.bias"):
            assert named[k].grad.abs().max().item() > 0.0, k


def test_training_cell_tensors(tmp_path):
    """The training executor's activation layouts (csrc/resnet_train.hip: bn_apply_groups_kernel).
    ADX_TRAIN_CELLS=1: the map between a BasicBlock's two convs as a pre-split cell tensor (the second conv's forward and its weight
    gradient read the halves their staging would have computed) and the ReLU mask of a block's output as one bit per element.  With
    the weight gradients reduced in index order a train-mode forward + backward must then come out BIT FOR BIT as with fp32 NCHW
    activations everywhere (ADX_TRAIN_CELLS=0): feature, every gradient, the running statistics.
    Level 2: the blocks' outputs are cell tensors too, and the identity the next block adds is hi + lo / 2^11 -- 22 bits of the
    fp32 value, as in the inference executor; 4: + the 16x16x32 forward launches and the conv-output gradients as cell tensors under a
    scale taken from a bound (bn_bwd_apply_groups_kernel); default (5): + the 16x16x32 data gradients: the feature within 2e-6 of its scale, gradients within what a handful of ReLU units
    flipping costs on these small batches (the oracle-referenced bars are test_perception_train_mode_vs_oracle_autograd's and the full-size tests')."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for cells in ("0", "1", "2", "4", "5"):
        out = str(tmp_path / f"cells{cells}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "train_cells_worker.py"), out],
                           env=dict(os.environ, ADX_TRAIN_CELLS=cells, ADX_WGRAD_DETERMINISTIC="1"), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[cells] = torch.load(out)
    assert outs["0"].keys() == outs["1"].keys() == outs["2"].keys() == outs["4"].keys() == outs["5"].keys() and len(outs["0"]) == 4
    for case, a in outs["0"].items():
        b = outs["1"][case]
        assert a.keys() == b.keys() and len(a) > 100
        for k in a:
            if k in ("grad.conv1.weight", "grad.fc.weight", "grad.fc.bias"):       # reduced with float atomics in either mode
                assert (a[k] - b[k]).abs().max().item() <= 2e-6 * a[k].abs().max().item(), (case, k)
            else:
                assert torch.equal(a[k], b[k]), (case, k, (a[k].double() - b[k].double()).abs().max().item())
        for level in ("2", "4", "5"):
            c = outs[level][case]
            assert a.keys() == c.keys()
            for k in a:
                if not a[k].is_floating_point():
                    assert torch.equal(a[k], c[k]), (case, level, k)
                    continue
                scale = a[k].abs().max().item() + 1e-30
                err = (a[k].double() - c[k].double()).abs().max().item() / scale
                if k == "feature" or k.startswith("buffer."):
                    assert err <= 2e-6, (case, level, k, err)
                else:
                    rel = ((a[k].double() - c[k].double()).norm() / (a[k].double().norm() + 1e-30)).item()
                    assert rel <= 3e-2, (case, level, k, rel)      # (a batch of 1..8 small images: a flipped ReLU unit is percents of a gradient)
            assert any(not torch.equal(a[k], c[k]) for k in a), (case, level)       # (the level did run: something is rounded differently)


def test_perception_backward_is_exactly_linear_in_the_incoming_gradient(tmp_path):
    """The perception backward is linear in d(loss)/d(feature), and every scale the split-fp16 kernels choose for a gradient tensor is
    a power of two taken from that tensor's own range (exact maxima for the fp32-layout launches, the bound of
    bn_bwd_consts_kernel for the cell-layout ones): multiplying the incoming gradient by 2^k must multiply EVERY parameter gradient by
    exactly 2^k -- same mantissas, bit for bit -- for tiny and for large gradients alike (k = -24, +10), with the weight gradients
    reduced in index order.  A scale that ignored the tensor's range (or an underflowing intermediate) breaks this."""
    import subprocess
    import sys
    if os.environ.get("ADX_CONV_EXACT") == "1" or os.environ.get("ADX_WGRAD_EXACT") == "1":
        pytest.skip("the exact-fp32 weight-gradient kernel reduces with float atomics: no bit-for-bit statement")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for k in ("0", "-24", "10"):
        out = str(tmp_path / f"scale{k}.pt")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "train_cells_worker.py"), out],
                           env=dict(os.environ, ADX_WGRAD_DETERMINISTIC="1", ADX_TEST_GRAD_SCALE_LOG2=k), capture_output=True, text=True,
                           timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[k] = torch.load(out)
    for case, a in outs["0"].items():
        for k in ("-24", "10"):
            b, f = outs[k][case], 2.0 ** float(k)
            assert torch.equal(a["feature"], b["feature"]), (case, k)
            for name in a:
                if not name.startswith("grad."):
                    continue
                if name in ("grad.conv1.weight", "grad.fc.weight", "grad.fc.bias"):       # reduced with float atomics
                    assert (a[name] * f - b[name]).abs().max().item() <= 2e-6 * f * a[name].abs().max().item(), (case, k, name)
                else:
                    assert torch.equal(a[name] * f, b[name]), (case, k, name, ((a[name] * f).double() - b[name].double()).abs().max().item())


@pytest.mark.parametrize("hw,small_gamma", [((64, 96), False), ((70, 102), False), ((64, 96), True)])
def test_perception_train_mode_vs_oracle_autograd(hw, small_gamma):
    """Batch-statistics BatchNorm forward, running-buffer update and every ResNet-34 parameter gradient ((70, 102): odd map
    widths, i.e. the generic paths of the pooling / BatchNorm passes).

    These gradients are not a smooth function of the arithmetic: with 18..72 samples per channel the deep
    batch-norms amplify fp32 rounding to ~1e-5 in the activations, and every ReLU unit whose pre-activation
    lies that close to zero flips its mask.  torch's own fp32 CPU path differs from an fp64 evaluation of the
    same graph by one flipped unit of the final map here (0.7 % of the gradient norm, tests/diagnostics/dbg_pgrad.py);
    which units flip depends on the summation order of each implementation.  The bar is therefore "as close
    to the fp64 oracle as the reference's fp32 arithmetic is" with room for a different set of flips (x3),
    per parameter tensor; the smooth part is pinned by the per-op gradient tests above."""
    from oracle import resnet as R
    from test_gpu_model import make_model
    m, _ = make_model("NO_GUIDANCE", 16)
    m.train()
    sd = oracle_sd("NO_GUIDANCE")
    if small_gamma:
        # bn1 channels whose |gamma| is far below |beta|: the stem's BatchNorm-backward sums may not take xhat from the pooled
        # value there ((pooled - beta) / gamma cancels; csrc/resnet_train.hip: stem_pool_bn_bwd_kernel<0> falls back to the
        # gather pass for such a channel).  One channel below the gate, one just above it.
        sd = {k: v.clone() for k, v in sd.items()}
        for ch, (ga, be) in {3: (1e-6, 0.1), 17: (-2e-5, -0.3), 40: (0.02, 1.0)}.items():
            sd["perception.bn1.weight"][ch], sd["perception.bn1.bias"][ch] = ga, be
        with torch.no_grad():
            m.perception.bn1.weight.copy_(sd["perception.bn1.weight"])
            m.perception.bn1.bias.copy_(sd["perception.bn1.bias"])
    pkeys = [e.key for e in unet_entries("NO_GUIDANCE") if e.key.startswith("perception.") and not e.is_buffer]
    img = P.synthetic_batch(3, 16, image_hw=hw, seed=61)["imgs"]
    w = P._uniform("perc.w", 61, (3, 64), -1.0, 1.0)

    def oracle_grads(dtype):
        s_ = {k: (v.detach().to(dtype).requires_grad_(k in pkeys) if v.is_floating_point() else v) for k, v in sd.items()}
        f = R.resnet34_forward(s_, "perception.", img.to(dtype), training=True)
        (f * w.to(dtype)).sum().backward()
        return f.detach(), {k: s_[k].grad for k in pkeys}

    f64, g64 = oracle_grads(torch.float64)
    f32, g32 = oracle_grads(torch.float32)
    rm0 = m.perception.bn1.running_mean.detach().clone()
    feat = m.perception(img.to(DEV))
    close(feat.detach().cpu(), f64.float(), 2e-4, rtol=1e-4)
    (feat * w.to(DEV)).sum().backward()
    named = dict(m.named_parameters())
    rel = lambda a, b: ((a.double() - b).norm() / (b.norm() + 1e-30)).item()  # noqa: E731
    for k in pkeys:
        if small_gamma and not k.startswith(("perception.conv1.", "perception.bn1.")):
            continue       # this case pins the stem's BatchNorm backward; which deep ReLU units flip is the other two cases' bar
        e_hip, e_ref = rel(named[k].grad.cpu(), g64[k]), rel(g32[k], g64[k])
        assert e_hip <= 3 * e_ref + 1e-3, (k, e_hip, e_ref)
    if small_gamma:        # the tweaked channels' own affine gradients, element by element (a tensor norm would hide one channel)
        for k in ("perception.bn1.weight", "perception.bn1.bias"):
            for ch in (3, 17, 40):
                a, b, c32 = named[k].grad[ch].item(), g64[k][ch].item(), g32[k][ch].item()
                assert abs(a - b) <= 3 * abs(c32 - b) + 1e-4 * max(1.0, abs(b)), (k, ch, a, b, c32)
    # running statistics moved like nn.BatchNorm2d(momentum=0.1): new = 0.9 old + 0.1 batch
    x1 = F.conv2d(img, sd["perception.conv1.weight"], None, stride=2, padding=3)
    want = 0.9 * rm0.cpu() + 0.1 * x1.mean(dim=(0, 2, 3))
    close(m.perception.bn1.running_mean.cpu(), want, 1e-5)
    assert int(m.perception.bn1.num_batches_tracked) == 1


@pytest.mark.parametrize("use_cond", ["NO_GUIDANCE", "FREE_GUIDANCE", "CLASSIFIER_GUIDANCE", "FREE_GUIDANCE_DROP"])
def test_training_step_vs_golden(golden, use_cond):
    """T1 end to end at the fixture's shape (B = 2, H = 16, 64x96 image): loss and gradient norms of the
    REAL reference (tests/golden/train.npz).  FREE_GUIDANCE_DROP: the same step with cond=None, the branch
    train.py:236-242 takes with probability 1 - USE_FREE_COND_PROB per batch."""
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from helpers import SCHED_KW
    from test_gpu_model import make_model
    g = golden("train")
    tag, use_cond, drop = use_cond, use_cond.replace("_DROP", ""), use_cond.endswith("_DROP")
    m, _ = make_model(use_cond, 16)
    m.train()
    if hasattr(m, "state_pred"):
        m.state_pred.dropout_p = 0.0      # the fixture was generated with every dropout set to p = 0
    d = {k: v.to(DEV) for k, v in P.synthetic_batch(2, 16, image_hw=(64, 96), seed=41).items()}
    sch = S.DDPMScheduler(**SCHED_KW)
    noisy = sch.add_noise(d["trajs"], d["noise"], d["t"], zero_first=True)
    pred = m(noisy, d["imgs"], d["t"], cond=d["target"] if (use_cond == "FREE_GUIDANCE" and not drop) else None)
    loss = F.mse_loss(pred, d["trajs"])
    assert abs(loss.item() - float(g[f"train.{tag}.loss"])) < 2e-5
    loss.backward()
    named = dict(m.named_parameters())
    if drop:
        w0 = named["cond_mlp.0.weight"].grad        # exists (DDP needs every gradient) and is exactly zero, as in the reference
        assert w0 is not None and w0.abs().max().item() == 0.0 and float(g[f"train.{tag}.gradnorm.cond_mlp.0.weight"]) == 0.0
    pre = f"train.{tag}.gradnorm."
    for k in g.files:
        if k.startswith(pre):
            ref = float(g[k])
            got = named[k[len(pre):]].grad.norm().item()
            assert abs(got - ref) <= 2e-3 * max(1.0, abs(ref)), (k, got, ref)
    assert all(p.grad is not None for p in m.parameters())
    # whole gradient tensors of the REAL reference (leading 70,000 elements of the large ones): element-wise, so a
    # permuted, transposed or sign-flipped gradient cannot hide behind a matching norm.
    #  * temporal stack (no ReLU: Mish, GroupNorm): relative L2 <= 1e-3 + 4 e_ref, where e_ref is the distance of the
    #    reference's own fp32 gradient from the fp64 truth (the oracle run in fp64 on the same inputs).
    #  * perception: the ReLU gradient is discontinuous.  A pre-activation within forward rounding (1e-6..5e-5 after 30
    #    layers) of zero takes a different mask in two fp32 implementations; ONE such element in a layer4 map of this
    #    fixture (2 x 3 pixels x batch 2 = 12 elements per channel) moves that layer's bias gradient by ~5e-3 of its norm
    #    and everything upstream of it by ~2e-3 (measured: tools/dbg_tape_cmp.py finds 10 flipped masks among 7.6 M taped
    #    activations between the split-fp16 and the exact-fp32 forward; tests/diagnostics/dbg_block_grad.py shows the incoming gradient
    #    itself is right to 3e-6).  torch's CPU fp32 backward shows the same effect against fp64 (e_ref up to 3e-2 on other
    #    seeds).  So the bar for perception tensors at this size is 3e-2; the tight per-tensor bar for the perception
    #    backward is in test_gpu_fullsize.py (16 x 8 x 29 elements per channel: a flip weighs 1e-4 there) and the kernels
    #    themselves are held to fp64 in test_gpu_conv2d.py.
    from oracle import sampling as OS
    pkeys = [e.key for e in unet_entries(use_cond) if not e.is_buffer]
    dc = {k: v.cpu() for k, v in d.items()}
    sd64 = {k: (v.double().requires_grad_(k in pkeys) if v.is_floating_point() else v) for k, v in oracle_sd(use_cond).items()}
    c64 = lambda t: t.double() if t.is_floating_point() else t  # noqa: E731
    OS.training_loss(sd64, c64(dc["imgs"]), c64(dc["trajs"]), c64(dc["target"]), dc["t"], c64(dc["noise"]),
                     use_cond=use_cond, drop_cond=drop).backward()
    pre = f"train.{tag}.gradfull."
    checked = 0
    for k in g.files:
        if not k.startswith(pre):
            continue
        name = k[len(pre):]
        ref = torch.from_numpy(g[k]).float()
        cut = lambda t: t if t.numel() <= 70000 else t.reshape(-1)[:70000]  # noqa: E731
        got = cut(named[name].grad.detach().cpu())
        truth = cut(sd64[name].grad)
        assert got.shape == ref.shape, (k, got.shape, ref.shape)
        e_ref = ((ref.double() - truth).norm() / (truth.norm() + 1e-300)).item()
        e = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
        e64 = ((got.double() - truth).norm() / (truth.norm() + 1e-300)).item()
        bar = max(1e-3 + 4 * e_ref, 3e-2 if name.startswith("perception.") else 0.0)
        assert e <= bar and e64 <= bar, (k, e, e64, e_ref)
        # a permuted / transposed / sign-flipped tensor has relative error ~1.4: far beyond either bar
        checked += 1
    assert checked >= 12


def test_two_optimizer_steps_and_checkpoint_vs_reference_written_fixture(golden, tmp_path):
    """tests/golden/ckpt.npz + ckpt_spec.json describe a checkpoint the REFERENCE's objects wrote (reference model, torch
    AdamW, the LambdaLR of get_constant_schedule_with_warmup, EMA rule of diffusers) after two iterations of
    train.py:221-261.  The same two iterations run here on the HIP path with FusedAdamWEMA; the file save_checkpoint
    writes must have the same layout and the same numbers, and must load into the reference's readers."""
    import json
    import os
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from autonomous_driving_with_diffusion_model_amd.checkpoint import save_checkpoint
    from autonomous_driving_with_diffusion_model_amd.optim import FusedAdamWEMA
    from helpers import SCHED_KW
    from test_gpu_model import make_model
    g = golden("ckpt")
    spec = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt_spec.json")))
    m, _ = make_model("NO_GUIDANCE", 16)
    m.train()
    kw = spec["ema_kw"]
    opt = FusedAdamWEMA(m.parameters(), lr=1e-4, warmup_steps=spec["warmup"], ema_update_after_step=kw["update_after_step"],
                        ema_power=kw["power"], ema_inv_gamma=kw["inv_gamma"], ema_max_decay=kw["max_decay"])
    d = {k: v.to(DEV) for k, v in P.synthetic_batch(2, 16, image_hw=(64, 96), seed=51).items()}
    sch = S.DDPMScheduler(**SCHED_KW)
    for it in range(spec["iter"]):
        noisy = sch.add_noise(d["trajs"], d["noise"], d["t"], zero_first=True)
        loss = F.mse_loss(m(noisy, d["imgs"], d["t"]), d["trajs"])
        assert abs(loss.item() - float(g[f"ckpt.loss.{it}"])) < 2e-5, it
        loss.backward()
        opt.step()
        opt.zero_grad()
    path = str(tmp_path / "checkpoint_2.pth")
    save_checkpoint(path, m, opt, iteration=spec["iter"])
    ck = torch.load(path, map_location="cpu", weights_only=True)
    # layout
    assert list(ck) == spec["keys"] and ck["iter"] == spec["iter"]
    assert list(ck["state_dict"]) == spec["state_dict_keys"]
    assert abs(os.path.getsize(path) - spec["file_bytes"]) < 0.01 * spec["file_bytes"]
    assert len(ck["optimizer"]["state"]) == spec["optimizer_state_len"]
    want_group = spec["optimizer_param_groups"][0]
    got_group = ck["optimizer"]["param_groups"][0]
    assert set(got_group) == set(want_group)
    for k, v in want_group.items():
        gv = got_group[k]
        assert (list(gv) if isinstance(gv, tuple) else gv) == (pytest.approx(v) if isinstance(v, float) else v), k
    assert {k: [str(v.dtype), list(v.shape)] for k, v in ck["optimizer"]["state"][0].items()} == spec["optimizer_state_entry"]
    assert set(ck["lr_scheduler"]) == set(spec["lr_scheduler"])
    for k, v in spec["lr_scheduler"].items():
        assert ck["lr_scheduler"][k] == (pytest.approx(v) if k in ("_last_lr", "base_lrs") else v), k
    assert {k: (v if not isinstance(v, list) else len(v)) for k, v in ck["ema_state_dict"].items()} == spec["ema_keys"]
    # numbers: weights after two AdamW steps, both moments, EMA shadow, BatchNorm buffers
    names = spec["parameter_names"]
    assert float(ck["optimizer"]["state"][0]["step"]) == float(g["ckpt.step"])
    for k in g.files:
        kind, _, name = k.partition(".")[2].partition(".")
        if kind not in ("param", "exp_avg", "exp_avg_sq", "shadow"):
            continue
        ref = torch.from_numpy(g[k])
        i = names.index(name)
        got = {"param": lambda: ck["state_dict"][name], "exp_avg": lambda: ck["optimizer"]["state"][i]["exp_avg"],
               "exp_avg_sq": lambda: ck["optimizer"]["state"][i]["exp_avg_sq"],
               "shadow": lambda: ck["ema_state_dict"]["shadow_params"][i]}[kind]()
        if kind in ("param", "shadow"):
            # warm-up: the first step runs at lr = 0, the second at 1e-5, and Adam's normalised update m / sqrt(v) is
            # O(1) whatever the gradient's size -- an element whose gradient is ~0 can take either sign in two fp32
            # implementations, so individual weights may differ by up to ~2 lr = 2e-5 (measured: 4e-6)
            close(got, ref, 2e-5, rtol=1e-6)
        else:
            e = ((got - ref).norm() / (ref.norm() + 1e-30)).item()
            assert e <= (2e-3 if kind == "exp_avg_sq" else 1e-3), (k, e)
    close(ck["state_dict"]["perception.bn1.running_mean"], g["ckpt.bn_running_mean"], 1e-5)
    assert int(ck["state_dict"]["perception.bn1.num_batches_tracked"]) == int(g["ckpt.bn_num_batches"])
    # the reference's readers accept the file (train.py:197-201): AdamW, LambdaLR
    cpu = [torch.nn.Parameter(p.detach().cpu()) for p in m.parameters()]
    ref_opt = torch.optim.AdamW(cpu, lr=1e-4, betas=(0.95, 0.999), eps=1e-7)
    ref_lrs = torch.optim.lr_scheduler.LambdaLR(ref_opt, lambda k_: min(1.0, k_ / spec["warmup"]))
    ref_opt.load_state_dict(ck["optimizer"])
    ref_lrs.load_state_dict(ck["lr_scheduler"])
    assert ref_lrs.last_epoch == spec["iter"] and ref_opt.param_groups[0]["lr"] == pytest.approx(2e-5)


@pytest.mark.parametrize("use_cond", ["FREE_GUIDANCE", "CLASSIFIER_GUIDANCE"])
def test_model_under_torch_ddp_world1_nccl(use_cond):
    """train.py:176-178: accelerate wraps the model in DistributedDataParallel with torch's defaults
    (find_unused_parameters=False, broadcast_buffers=True).  The package's model is custom autograd nodes over a
    parameter-holder module tree, which is exactly where DDP's reducer breaks if a parameter's gradient does not arrive
    through its AccumulateGrad node: wrap it (RCCL, world_size 1), run two iterations, and require every parameter's
    gradient to equal the unwrapped model's."""
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from autonomous_driving_with_diffusion_model_amd import scheduler as S
    from helpers import SCHED_KW
    from test_gpu_model import make_model
    m, _ = make_model(use_cond, 16)
    m.train()
    if hasattr(m, "state_pred"):
        m.state_pred.dropout_p = 0.0
    d = {k: v.to(DEV) for k, v in P.synthetic_batch(3, 16, image_hw=(64, 96), seed=43).items()}
    sch = S.DDPMScheduler(**SCHED_KW)
    noisy = sch.add_noise(d["trajs"], d["noise"], d["t"], zero_first=True)
    cond = d["target"] if use_cond == "FREE_GUIDANCE" else None

    def grads_of(model):
        for p in m.parameters():
            p.grad = None
        loss = F.mse_loss(model(noisy, d["imgs"], d["t"], cond=cond), d["trajs"])
        loss.backward()
        assert all(p.grad is not None for p in m.parameters())
        return loss.item(), [p.grad.detach().clone() for p in m.parameters()]

    loss0, g0 = grads_of(m)
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device(DEV))
    try:
        ddp = DDP(m, device_ids=[0])
        for it in range(2):                 # a parameter left without a gradient makes DDP raise in the NEXT forward
            loss1, g1 = grads_of(ddp)
            assert abs(loss1 - loss0) < 1e-6
            for (k, _), a, b in zip(m.named_parameters(), g1, g0):
                # weight gradients are reduced with float atomics: run-to-run order differs in the last bits
                assert rel_err(a, b.cpu()) <= 2e-5, (it, k)
        assert int(m.perception.bn1.num_batches_tracked) == 3
    finally:
        dist.destroy_process_group()


def test_fused_adamw_ema_matches_torch_adamw():
    """adx_adamw_ema_step vs torch.optim.AdamW(betas=(0.95, 0.999), eps=1e-7) + nan_to_num + EMA (train.py:252-261)."""
    from autonomous_driving_with_diffusion_model_amd.optim import FusedAdamWEMA, ema_decay
    shapes = [(64, 7, 5), (3840, 128), (513,), (2049,), (1,)]
    ps = [torch.nn.Parameter(uni(f"opt.p{i}", s).to(DEV)) for i, s in enumerate(shapes)]
    qs = [torch.nn.Parameter(p.detach().clone()) for p in ps]
    opt = FusedAdamWEMA(ps, lr=1e-3, warmup_steps=2, ema_update_after_step=0, ema_power=0.75)
    ref = torch.optim.AdamW(qs, lr=1e-3, betas=(0.95, 0.999), eps=1e-7)
    shadow = [q.detach().clone() for q in qs]
    for step in range(1, 5):
        for i, (p, q) in enumerate(zip(ps, qs)):
            g = uni(f"opt.g{i}.{step}", p.shape).to(DEV)
            if i == 2 and step == 2:
                g[0], g[1], g[2] = float("nan"), float("inf"), float("-inf")
            p.grad, q.grad = g.clone(), g.clone()
        for q in qs:
            torch.nan_to_num(q.grad, nan=0, posinf=1e5, neginf=-1e5, out=q.grad)
        for grp in ref.param_groups:
            grp["lr"] = 1e-3 * min(1.0, (step - 1) / 2)
        ref.step()
        d = ema_decay(step, update_after_step=0, inv_gamma=1.0, power=0.75, max_decay=0.9999)
        for s_, q in zip(shadow, qs):
            s_.sub_((1 - d) * (s_ - q.detach()))
        opt.step()
        for p, q, s_, e in zip(ps, qs, shadow, opt.shadow_params):
            close(p.detach().cpu(), q.detach().cpu(), 2e-6, rtol=1e-5)
            close(e.cpu(), s_.cpu(), 2e-6, rtol=1e-5)
    assert ps[0]._version > 0


def test_training_conv_launches_persistent_vs_one_tile_per_workgroup_bit_identical(tmp_path):
    """csrc/conv2d_hs16.hip (round 6): the training forward (`conv2d_hs3x3q_kernel<DMA, 1>`: fp32 output + BatchNorm partial sums)
    and data-gradient (`<DMA, 2>`: dx + the consumer BatchNorm's backward sums through mask bits) launches are persistent like the
    inference ones -- a workgroup walks several tiles and uses the dead patch buffer as its statistics scratch while the other
    already holds the next tile's first chunk.  Full-size NO_GUIDANCE step (B = 64, 3 x 256 x 900: 3.6 / 1.8 tiles per workgroup on
    the 128 / 256-channel layers) with ADX_HS_PERSIST=0 against the default, both with ADX_WGRAD_DETERMINISTIC=1: every gradient
    tensor that two DEFAULT runs reproduce bit for bit (all of the encoder's 3x3 conv weights and BatchNorm affines: the weight
    gradients that reduce with float atomics -- the stem's, the temporal stack's -- drop out) has the same bits."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    runs = {}
    for tag, env in (("a", {}), ("b", {}), ("one_tile", {"ADX_HS_PERSIST": "0"})):
        out = str(tmp_path / f"{tag}.json")
        r = subprocess.run([sys.executable, os.path.join(root, "tests", "train_checksum_worker.py"), out, "64", "32", "256", "900", "7"],
                           env=dict(os.environ, ADX_WGRAD_DETERMINISTIC="1", **env), capture_output=True, text=True, timeout=1200)
        assert r.returncode == 0, r.stderr[-3000:]
        runs[tag] = json.load(open(out))
    stable = [k for k, v in runs["a"]["sums"].items() if runs["b"]["sums"][k] == v]
    enc = [k for k in stable if k.startswith("perception.layer") and (k.endswith("conv1.weight") or k.endswith("conv2.weight"))]
    assert len(enc) >= 28, (len(stable), len(enc))        # the 3x3 conv weights behind the persistent launches are among them
    assert runs["a"]["loss"] == runs["one_tile"]["loss"]
    diff = [k for k in stable if runs["one_tile"]["sums"][k] != runs["a"]["sums"][k]]
    assert diff == [], diff[:8]
