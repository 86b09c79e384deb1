"""conv2d kernels of the perception encoder, one launch at a time, against an fp64 evaluation of the same op.

The 3x3 stride-1 convs run on the fp16 matrix cores with hi/lo split operands (csrc/conv2d_hs.hip); the bar for
them is the bar of fp32 arithmetic: the max error against fp64 may not exceed 1.5x what torch's own fp32 convs
(CPU oneDNN and ROCm MIOpen, whichever is worse; they differ only in summation order) show on the same inputs,
+ 1e-7 relative slack.  Measured (tools/conv_err.py, K = 4608): split-fp16 rms 4.2e-7, exact-fp32 MFMA chain
1.2e-6, MIOpen fp32 5.8e-7, oneDNN 2.3e-7."""
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda:0"
# ADX_CONV_EXACT=1 routes every conv to the exact-fp32 MFMA kernels (one sequential fp32 chain over K): their bar is
# the looser one of test_exact_fp32_mfma_conv_shapes
EXACT = os.environ.get("ADX_CONV_EXACT") == "1" or os.environ.get("ADX_WGRAD_EXACT") == "1"
BAR = 4.0 if EXACT else 1.5
Q_KERNEL = os.environ.get("ADX_HS_MODE") is None and not EXACT      # the 16x16x32 kernel is on (csrc/conv2d_hs16.hip)
split_only = pytest.mark.skipif(EXACT, reason="property of the split-fp16 kernels; ADX_CONV_EXACT=1 selects the exact ones")


def _ops():
    from autonomous_driving_with_diffusion_model_amd import ops
    return ops


def _case(cin, cout, k, h, w, n, seed, xscale=1.0, wscale=None):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, cin, h, w, generator=g) * xscale
    ws = (2.0 / (k * k * cin)) ** 0.5 if wscale is None else wscale
    wt = torch.randn(cout, cin, k, k, generator=g) * ws
    return x, wt


def _errs(y_hip, x, wt, stride, pad, post=None):
    ref = F.conv2d(x.double(), wt.double(), stride=stride, padding=pad)
    f32 = F.conv2d(x, wt, stride=stride, padding=pad)
    g32 = F.conv2d(x.to(DEV), wt.to(DEV), stride=stride, padding=pad).cpu()
    if post is not None:
        ref, f32, g32 = post(ref), post(f32), post(g32)
    den = ref.abs().max().item() + 1e-300
    err = lambda t: (t.double().cpu() - ref).abs().max().item() / den  # noqa: E731
    return err(y_hip), max(err(f32), err(g32))


@pytest.mark.parametrize("cin,cout,h,w,n", [(64, 64, 64, 225, 2), (128, 128, 32, 113, 2), (256, 256, 16, 57, 3),
                                            (512, 512, 8, 29, 2), (64, 128, 13, 37, 1), (16, 64, 5, 3, 2),
                                            (32, 192, 9, 70, 1), (48, 64, 11, 33, 2), (80, 128, 17, 20, 1)])
def test_conv3x3_s1_split_fp16_is_fp32_grade(cin, cout, h, w, n):
    x, wt = _case(cin, cout, 3, h, w, n, seed=cin + h)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1)
    e_hip, e_f32 = _errs(y, x, wt, 1, 1)
    assert e_hip <= BAR * e_f32 + 1e-7, (e_hip, e_f32)


@split_only
@pytest.mark.parametrize("xscale,wscale", [(2e-3, 1e-2), (300.0, 0.05), (1.0, 30.0), (0.05, 0.002)])
def test_conv3x3_split_keeps_relative_accuracy_across_magnitudes(xscale, wscale):
    """hi + 2^-11 lo keeps 22 significant bits for every operand with 2^-14 <= |x| < 65504 (fp16's normal range);
    smaller elements keep an ABSOLUTE accuracy of 2^-36, which is far below fp32's own rounding of anything they
    are added to as long as the tensor has elements above ~1e-4."""
    x, wt = _case(64, 64, 3, 16, 40, 1, seed=5, xscale=xscale, wscale=wscale)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1)
    e_hip, e_f32 = _errs(y, x, wt, 1, 1)
    assert torch.isfinite(y).all()
    assert e_hip <= BAR * e_f32 + 1e-7, (e_hip, e_f32)


@split_only
def test_conv3x3_split_degrades_gracefully_below_fp16_normal_range():
    """A tensor whose EVERY element is below fp16's normal range (|x| ~ 3e-6) loses bits gradually (hi is an fp16
    subnormal), it does not flush: the result is still well inside the 1e-4 parity bar."""
    x, wt = _case(64, 64, 3, 16, 40, 1, seed=5, xscale=3e-6, wscale=1.0)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1)
    e_hip, _ = _errs(y, x, wt, 1, 1)
    assert e_hip < 2e-5, e_hip


def test_conv3x3_fused_bn_residual_relu_epilogue():
    x, wt = _case(128, 128, 3, 20, 45, 2, seed=9)
    g = torch.Generator().manual_seed(10)
    scale, shift = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    res = torch.randn(2, 128, 20, 45, generator=g)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1, scale=scale.to(DEV), shift=shift.to(DEV),
                         res=res.to(DEV), relu=True)
    post = lambda c: torch.relu(c * scale.to(c.dtype)[None, :, None, None] + shift.to(c.dtype)[None, :, None, None]  # noqa: E731
                                + res.to(c.dtype))
    e_hip, e_f32 = _errs(y, x, wt, 1, 1, post)
    assert e_hip <= BAR * e_f32 + 2e-7, (e_hip, e_f32)


@split_only
@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 64, 2, 64, 225), (64, 64, 1, 16, 16), (128, 128, 3, 32, 113), (64, 128, 5, 13, 37),
                                            (256, 256, 20, 16, 57), (512, 512, 40, 8, 29), (512, 512, 12, 8, 29)])
def test_conv3x3_cell_layout_operands_match_the_fp32_layout_bit_for_bit(cin, cout, n, h, w):
    """The pipelined 3x3 kernel reading / writing CELL tensors (csrc/conv2d_hs.hip: per image [C / 8][hi, lo][H][W] cells of
    eight fp16 channels; adx_conv2d_forward_cells): a cell output is exactly the split of the fp32-layout launch's output
    (same products, same epilogue arithmetic), whether the input came as fp32 or as cells; a residual read from cells is
    hi + lo / 2^11 instead of the fp32 value, so that case agrees to 2^-22 of the magnitudes involved.  The shapes take all
    three tile modes (4 waves; 8 waves 16 rows for >= 256 channels on tall maps; 8 waves 128 channels on 8-row maps): the
    8-wave code orders its stores differently, which is where a missing wait state after the 16-byte stores showed."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin + n)
    x, wt = _case(cin, cout, 3, h, w, n, seed=cin + w)
    x, wt = x.to(DEV), wt.to(DEV)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(DEV), torch.randn(cout, generator=g).to(DEV)
    res = torch.randn(n, cout, h, w, generator=g).to(DEV)
    y0, packed = ops.conv2d(x, wt, stride=1, pad=1, scale=sc, shift=sh, res=res, relu=True)
    want = ops.to_cells(y0)
    xc, rc = ops.to_cells(x), ops.to_cells(res)
    assert torch.equal(ops.from_cells(ops.to_cells(y0), y0.shape), ops.from_cells(want, y0.shape))
    # all-cell launches with Cout % 128 == 0 and Cin % 64 == 0 run on the 16x16x32 kernel (csrc/conv2d_hs16.hip): the same
    # products in another summation order, so they agree with the 32x32x16 launch like two fp32 evaluations do, not to the bit
    # (its own fp64-referenced bar: test_conv3x3_16x16x32_kernel_is_fp32_grade)
    q = Q_KERNEL and cout % 128 == 0 and cin % 64 == 0
    mag = max(1.0, y0.abs().max().item(), res.abs().max().item())
    for rep in range(3):                 # the store hazard was a run-to-run effect
        for x_cells in (False, True):
            got = ops.conv2d_cells(xc if x_cells else x, packed, cin, cout, n, h, w, x_cells=x_cells, scale=sc, shift=sh, res=res,
                                   relu=True)
            assert torch.equal(got, want), (x_cells, rep, (ops.from_cells(got, y0.shape) - y0).abs().max().item())
            got = ops.conv2d_cells(xc if x_cells else x, packed, cin, cout, n, h, w, x_cells=x_cells, scale=sc, shift=sh, res=rc,
                                   res_cells=True, relu=True)
            err = (ops.from_cells(got, y0.shape) - y0).abs().max().item()
            assert err <= 2.0 ** (-19 if q and x_cells else -21) * mag, (x_cells, rep, err)
    # no epilogue operands at all
    y1, _ = ops.conv2d(x, wt, stride=1, pad=1, packed=packed)
    got = ops.conv2d_cells(xc, packed, cin, cout, n, h, w, x_cells=True)
    if q:
        assert (ops.from_cells(got, y1.shape) - y1).abs().max().item() <= 2.0 ** -19 * max(1.0, y1.abs().max().item())
    else:
        assert torch.equal(got, ops.to_cells(y1))


@split_only
@pytest.mark.skipif(os.environ.get("ADX_HS_MODE") is not None, reason="ADX_HS_MODE pins the 32x32x16 kernel")
@pytest.mark.parametrize("cin,cout,n,h,w", [(128, 128, 3, 32, 113), (64, 128, 5, 13, 37), (256, 256, 20, 16, 57), (512, 512, 12, 8, 29),
                                            (192, 384, 8, 9, 21), (128, 256, 2, 40, 50), (256, 128, 70, 3, 5)])
def test_conv3x3_16x16x32_kernel_is_fp32_grade(cin, cout, n, h, w):
    """csrc/conv2d_hs16.hip (v_mfma_f32_16x16x32_f16; serves the all-cell launches with Cout % 128 == 0, Cin % 64 == 0) against an
    fp64 evaluation, with the bar of the 32x32x16 kernel: plain, and with BatchNorm affine + cell residual + ReLU; odd map sizes,
    rows and columns that do not fill the 8 x 32 tile, more images than a tile has columns, three runs (store hazards)."""
    ops = _ops()
    g = torch.Generator().manual_seed(cin + n)
    x, wt = _case(cin, cout, 3, h, w, n, seed=cin + w)
    sc, sh = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn(n, cout, h, w, generator=g)
    _, packed = ops.conv2d(x[:1].to(DEV), wt.to(DEV), stride=1, pad=1)
    xc, rc = ops.to_cells(x.to(DEV)), ops.to_cells(res.to(DEV))
    y = ops.from_cells(ops.conv2d_cells(xc, packed, cin, cout, n, h, w, x_cells=True), (n, cout, h, w))
    e_hip, e_f32 = _errs(y, x, wt, 1, 1)
    assert e_hip <= BAR * e_f32 + 1e-7, (e_hip, e_f32)
    post = lambda c: torch.relu(c * sc.to(c.dtype)[None, :, None, None] + sh.to(c.dtype)[None, :, None, None] + res.to(c.dtype))  # noqa: E731
    first = None
    for rep in range(3):
        yc = ops.conv2d_cells(xc, packed, cin, cout, n, h, w, x_cells=True, scale=sc.to(DEV), shift=sh.to(DEV), res=rc, res_cells=True,
                              relu=True)
        first = yc if first is None else first
        assert torch.equal(yc, first)                      # deterministic
    e_hip, e_f32 = _errs(ops.from_cells(first, (n, cout, h, w)), x, wt, 1, 1, post)
    assert e_hip <= BAR * e_f32 + 4e-7, (e_hip, e_f32)     # + the 2^-22 of a residual held as hi + lo / 2^11 and of the split output


@pytest.mark.parametrize("cin,cout,h,w", [(64, 128, 64, 225), (256, 512, 16, 57), (128, 256, 9, 31), (64, 64, 7, 8)])
def test_conv3x3_s2_split_fp16_is_fp32_grade(cin, cout, h, w):
    x, wt = _case(cin, cout, 3, h, w, 2, seed=cin + w)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=2, pad=1)
    e_hip, e_f32 = _errs(y, x, wt, 2, 1)
    assert e_hip <= BAR * e_f32 + 1e-7, (e_hip, e_f32)


@pytest.mark.parametrize("h,w,n", [(64, 96, 2), (256, 900, 1), (37, 45, 3), (32, 32, 1)])
def test_stem_7x7_s2_split_fp16_is_fp32_grade(h, w, n):
    x, wt = _case(3, 64, 7, h, w, n, seed=h)
    g = torch.Generator().manual_seed(h + 1)
    scale, shift = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=2, pad=3, scale=scale.to(DEV), shift=shift.to(DEV), relu=True)
    post = lambda c: torch.relu(c * scale.to(c.dtype)[None, :, None, None] + shift.to(c.dtype)[None, :, None, None])  # noqa: E731
    e_hip, e_f32 = _errs(y, x, wt, 2, 3, post)
    assert e_hip <= BAR * e_f32 + 2e-7, (e_hip, e_f32)


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [(64, 128, 1, 2, 0, 64, 225), (128, 64, 1, 1, 0, 9, 33)])
def test_exact_fp32_mfma_conv_shapes(cin, cout, k, stride, pad, h, w):
    x, wt = _case(cin, cout, k, h, w, 2, seed=k + stride + cin)
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=stride, pad=pad)
    e_hip, e_f32 = _errs(y, x, wt, stride, pad)
    # exact-fp32 MFMA kernels: one sequential fp32 chain over K, so up to ~sqrt(K) * eps (4x the blocked sums)
    assert e_hip <= 4 * e_f32 + 1e-7, (e_hip, e_f32)


@split_only
def test_conv_non_finite_inputs_stay_loud():
    x, wt = _case(64, 64, 3, 8, 32, 1, seed=2)
    x[0, 3, 4, 5] = float("nan")
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1)
    ref = F.conv2d(x, wt, padding=1)
    assert torch.equal(torch.isnan(y.cpu()), torch.isnan(ref))
    x[0, 3, 4, 5] = 1e6          # beyond fp16: must not turn into a silently wrong finite number
    y, _ = _ops().conv2d(x.to(DEV), wt.to(DEV), stride=1, pad=1)
    assert not torch.isfinite(y[0, :, 3:6, 4:7]).all()


@pytest.mark.parametrize("cin,cout,h,w,n,dyscale,stride", [(64, 64, 64, 225, 2, 1e-4, 1), (128, 128, 32, 113, 2, 1.0, 1),
                                                          (256, 256, 16, 57, 3, 3e-7, 1), (512, 512, 8, 29, 4, 1e-3, 1),
                                                          (64, 128, 9, 40, 1, 1e-5, 1), (128, 64, 5, 70, 2, 20.0, 1),
                                                          (64, 128, 64, 225, 2, 1e-4, 2), (128, 256, 32, 113, 2, 1.0, 2),
                                                          (256, 512, 16, 57, 3, 1e-6, 2), (64, 64, 7, 9, 2, 1e-2, 2),
                                                          (64, 128, 33, 70, 1, 1.0, 2),
                                                          # larger batches (several units per workgroup of a split)
                                                          (64, 64, 12, 225, 9, 1e-4, 1), (128, 128, 8, 113, 10, 1.0, 1),
                                                          (256, 256, 5, 57, 16, 3e-7, 1), (512, 512, 4, 29, 33, 1e-3, 1),
                                                          (64, 128, 6, 40, 7, 1e-5, 1)])
def test_conv3x3_weight_gradient_split_fp16(cin, cout, h, w, n, dyscale, stride):
    """adx_conv2d_wgrad (3x3, stride 1 and 2: fp16 matrix cores, transposing LDS reads, dy rescaled by its measured
    range) against an fp64 evaluation; the bar is torch's own fp32 weight gradient on CPU and on the GPU, x1.5."""
    g = torch.Generator().manual_seed(cin + w)
    x = torch.randn(n, cin, h, w, generator=g).relu_()
    oh, ow = (h - 1) // stride + 1, (w - 1) // stride + 1
    dy = torch.randn(n, cout, oh, ow, generator=g) * dyscale
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), stride=stride, padding=1)
    f32 = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, stride=stride, padding=1)
    g32 = torch.nn.grad.conv2d_weight(x.to(DEV), (cout, cin, 3, 3), dy.to(DEV), stride=stride, padding=1).cpu()
    dw = _ops().conv2d_weight_grad(x.to(DEV), dy.to(DEV), 3, stride=stride, pad=1)
    den = ref.abs().max().item()
    err = lambda t: (t.double().cpu() - ref).abs().max().item() / den  # noqa: E731
    assert err(dw) <= BAR * max(err(f32), err(g32)) + 2e-7, (err(dw), err(f32), err(g32))


@pytest.mark.parametrize("cin,cout,k,stride,pad,h,w", [(32, 64, 3, 2, 1, 32, 57), (64, 128, 1, 2, 0, 32, 57),
                                                       (3, 64, 7, 2, 3, 64, 96)])
def test_other_weight_gradients(cin, cout, k, stride, pad, h, w):
    g = torch.Generator().manual_seed(k + h)
    x = torch.randn(2, cin, h, w, generator=g)
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    dy = torch.randn(2, cout, oh, ow, generator=g)
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, k, k), dy.double(), stride=stride, padding=pad)
    dw = _ops().conv2d_weight_grad(x.to(DEV), dy.to(DEV), k, stride=stride, pad=pad)
    assert ((dw.double().cpu() - ref).abs().max() / ref.abs().max()).item() < 5e-6


def test_mfma_probe_counts_its_flops_and_is_deterministic():
    """adx_probe_mfma_fp16 (bench.py's `roofline.sustained`): the checksum of a launch is a pure function of the operand cells (the
    loop is really executed, on the operands given), zero operands give zero, and the reported flops are workgroups x 4 waves x
    iters x 12 MFMAs x 2 x 32 x 32 x 16."""
    import ctypes as C
    from autonomous_driving_with_diffusion_model_amd import _lib as L
    g = torch.Generator().manual_seed(5)
    cells = ((torch.rand(4096 * 8, generator=g) * 0.875 + 0.125) * 0.01).half().to(DEV)
    out = [torch.empty(8 * 256, device=DEV) for _ in range(2)]
    fl = C.c_double(0.0)
    for o in out:
        L.check(L.lib().adx_probe_mfma_fp16(cells.data_ptr(), o.data_ptr(), 8, 64, C.byref(fl), L.stream_ptr(torch.device(DEV))))
    torch.cuda.synchronize()
    assert fl.value == 8 * 4 * 64 * 12 * 2 * 32 * 32 * 16
    assert torch.equal(out[0], out[1]) and out[0].abs().max().item() > 0 and torch.isfinite(out[0]).all()
    z = torch.zeros(4096 * 8, dtype=torch.float16, device=DEV)
    L.check(L.lib().adx_probe_mfma_fp16(z.data_ptr(), out[0].data_ptr(), 8, 64, None, L.stream_ptr(torch.device(DEV))))
    assert out[0].abs().max().item() == 0.0


@pytest.mark.parametrize("h,w,n,dyscale", [(64, 96, 2, 1e-4), (70, 131, 3, 1.0), (32, 48, 1, 1e-2), (33, 250, 2, 3e-7),
                                           (256, 900, 2, 1e-6), (130, 450, 5, 1e-3)])
def test_stem_weight_gradient_split_fp16(h, w, n, dyscale):
    """The stem's 7x7 stride-2 weight gradient on the fp16 matrix cores (csrc/conv2d_wgrad_stem_hs.hip: output pixels as the
    reduction axis, stride and tap shift resolved while staging, dy rescaled by its measured range) against an fp64 evaluation;
    the bar is torch's own fp32 weight gradient on CPU and on the GPU, x1.5 -- odd sizes, several segments and row chunks,
    gradients far below fp16's range.  ADX_WGRAD_EXACT=1 keeps the exact-fp32 kernel (test_other_weight_gradients runs both
    through estimate_range)."""
    g = torch.Generator().manual_seed(h + w)
    x = torch.randn(n, 3, h, w, generator=g)
    oh, ow = (h + 6 - 7) // 2 + 1, (w + 6 - 7) // 2 + 1
    dy = torch.randn(n, 64, oh, ow, generator=g) * dyscale
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, 3, 7, 7), dy.double(), stride=2, padding=3)
    f32 = torch.nn.grad.conv2d_weight(x, (64, 3, 7, 7), dy, stride=2, padding=3)
    g32 = torch.nn.grad.conv2d_weight(x.to(DEV), (64, 3, 7, 7), dy.to(DEV), stride=2, padding=3).cpu()
    dw = _ops().conv2d_weight_grad(x.to(DEV), dy.to(DEV), 7, stride=2, pad=3)
    den = ref.abs().max().item()
    err = lambda t: (t.double().cpu() - ref).abs().max().item() / den  # noqa: E731
    assert err(dw) <= BAR * max(err(f32), err(g32)) + 2e-7, (err(dw), err(f32), err(g32))
    # without a range estimate the exact-fp32 kernel answers: both paths stay alive
    dwx = _ops().conv2d_weight_grad(x.to(DEV), dy.to(DEV), 7, stride=2, pad=3, estimate_range=False)
    assert err(dwx) < 5e-6


@split_only
def test_weight_gradient_deterministic_mode_is_bit_reproducible(tmp_path):
    """ADX_WGRAD_DETERMINISTIC=1 (csrc/conv2d_wgrad_hs.hip): every (tile, split) workgroup leaves its block in a copy of dw of its
    split and one more launch adds the copies up in index order, instead of float atomics whose arrival order moves the last
    bits from run to run.  At the full batch of a training step, every 3x3 shape: three runs bit-equal in a process with the
    switch, and equal to the atomic path of this process up to the order of the sums."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "tests"))
    import wgrad_det_worker as W
    out = str(tmp_path / "det.pt")
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "wgrad_det_worker.py"), out, "64"],
                       env=dict(os.environ, ADX_WGRAD_DETERMINISTIC="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    det = torch.load(out)
    for shape in W.SHAPES:
        equal, dw_det = det[shape]
        assert equal, shape
        cin, cout, s, h, w = shape
        x, dy = W.inputs(cin, cout, s, h, w, 64, DEV)
        dw = _ops().conv2d_weight_grad(x, dy, 3, stride=s, pad=1).cpu()
        err = (dw - dw_det).abs().max().item()
        assert err <= 2e-6 * dw.abs().max().item(), (shape, err)


def test_conv3x3q_persistent_launch_is_bit_identical(tmp_path):
    """csrc/conv2d_hs16.hip (round 6): the LDS-DMA launches are persistent -- one workgroup per CU walks its XCD's spatial tiles and
    fetches the next tile's first chunk and first tap under the last chunk of the current one.  Which workgroup computes a tile, and
    whether its first operands arrived in a prologue or under the previous tile, must not change a bit: the same launches with
    ADX_HS_PERSIST=0 (one tile per workgroup, a process of its own) -- several tiles per CU at one, two and four cout tiles, with and
    without a residual, ragged tile rows, tile counts that do not divide by eight, a single tile."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import conv_cells_worker as W
    out = str(tmp_path / "one_tile_per_workgroup.pt")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "conv_cells_worker.py"), out],
                       env=dict(os.environ, ADX_HS_PERSIST="0"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    ref = torch.load(out)
    got = W.outputs()
    assert set(got) == set(ref) and len(got) == len(W.CASES)
    for k, v in got.items():
        assert torch.equal(v, ref[k]), k


@pytest.mark.parametrize("cin,cout,h,w,n,dyscale", [(128, 128, 12, 113, 9, 1.0), (256, 256, 9, 57, 16, 3e-7), (512, 512, 8, 29, 33, 1e-3),
                                                    (64, 128, 7, 50, 5, 1e-5), (128, 256, 5, 29, 3, 30.0), (64, 64, 6, 70, 4, 1.0)])
def test_conv3x3_weight_gradient_of_cell_operands(cin, cout, h, w, n, dyscale, tmp_path):
    """adx_conv2d_wgrad_cells: the weight gradient as the training executor launches it -- both operands pre-split cell tensors, dy
    under a power-of-two scale (`conv2d_wgrad_hs_kernel<NPX, 1, true, true>`).  Same bar as the fp32-operand launch: fp64 reference,
    1.5 x torch's own fp32 weight gradient, per input channel; relu'd x (half the activations exactly zero) with channel scales over
    three decades, 0.01 .. 10.  (Round 6 built a 128 x 64 block on ONE accumulator with unscaled lo halves behind this entry point:
    4-6 % slower and 1.5e-6 where the bar is 1.1e-6 on the smallest reduction here -- removed, profiles/README.md.)"""
    g = torch.Generator().manual_seed(cin + w)
    x = torch.randn(n, cin, h, w, generator=g).relu_() * torch.logspace(-2, 1, cin, base=10.0).reshape(1, cin, 1, 1)
    dy = torch.randn(n, cout, h, w, generator=g) * dyscale
    ref = torch.nn.grad.conv2d_weight(x.double(), (cout, cin, 3, 3), dy.double(), stride=1, padding=1)
    f32 = torch.nn.grad.conv2d_weight(x, (cout, cin, 3, 3), dy, stride=1, padding=1)
    g32 = torch.nn.grad.conv2d_weight(x.to(DEV), (cout, cin, 3, 3), dy.to(DEV), stride=1, padding=1).cpu()
    dw = _ops().conv2d_weight_grad_cells(x.to(DEV), dy.to(DEV))
    # per input channel (their scales span three decades: a global maximum would hide the small ones)
    den = ref.abs().amax(dim=(0, 2, 3), keepdim=True)
    err = lambda t: ((t.double().cpu() - ref).abs() / den).max().item()  # noqa: E731
    assert err(dw) <= BAR * max(err(f32), err(g32)) + 2e-7, (err(dw), err(f32), err(g32))
