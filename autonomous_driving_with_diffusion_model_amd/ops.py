"""Op-level Python entry points over the C ABI (one call = one kernel launch).  Used by the parity
tests and available to callers who want a single fused op; the model itself goes through the
native executors (adx_unet_forward / adx_resnet_forward)."""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch

from . import _lib as L


def _strides3(t: torch.Tensor):
    return t.stride(0), t.stride(1), t.stride(2)


def tconv(x0: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, *, x1: Optional[torch.Tensor] = None,
          kind: int = 0, stride: int = 1, pad: int = 0, gn_weight: Optional[torch.Tensor] = None,
          gn_bias: Optional[torch.Tensor] = None, groups: int = 0, eps: float = 1e-5,
          tbias: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
          out: Optional[torch.Tensor] = None, scratch: Optional[torch.Tensor] = None,
          tickets: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Temporal conv (+bias) [-> GroupNorm(groups) -> Mish] [+ tbias[:, :, None]] [+ res].

    scratch: optional float32 device buffer (adx_tconv_io::scratch) that allows a tiny-batch launch to split its
    reduction over more workgroups; tickets: with it, 256 ZERO int32 device words (adx_tconv_io::tickets): the split
    reduction then needs no reduce launch and leaves the words zero.

    x0/x1: [B, C, L] views with arbitrary strides (x1 is concatenated after x0 along C);
    weight: Conv1d [Cout, Cin, k] (kind 0) or ConvTranspose1d [Cin, Cout, k] (kind 1)."""
    assert x0.is_cuda and x0.dtype == torch.float32 and x0.dim() == 3
    B, c0, lin = x0.shape
    c1 = 0 if x1 is None else x1.shape[1]
    taps = weight.shape[2]
    cout = weight.shape[0] if kind == 0 else weight.shape[1]
    lout = (lin + 2 * pad - taps) // stride + 1 if kind == 0 else (lin - 1) * stride - 2 * pad + taps
    d = L.TConvDesc(kind, taps, stride, pad, c0, c1, cout, lin, lout, groups, eps)
    nbytes = L.lib().adx_tconv_packed_bytes(C.byref(d))
    if nbytes == 0:
        L.check(-1, "adx_tconv_packed_bytes")
    packed = torch.empty(nbytes // 4, dtype=torch.float32, device=x0.device)
    s = L.stream_ptr(x0.device)
    w = weight.detach().contiguous()
    L.check(L.lib().adx_tconv_pack(C.byref(d), w.data_ptr(), packed.data_ptr(), s), "adx_tconv_pack")
    y = out if out is not None else torch.empty((B, cout, lout), dtype=torch.float32, device=x0.device)
    io = L.TConvIO()
    io.x0 = x0.data_ptr()
    io.x0_sb, io.x0_sc, io.x0_sl = _strides3(x0)
    if x1 is not None:
        io.x1 = x1.data_ptr()
        io.x1_sb, io.x1_sc, io.x1_sl = _strides3(x1)
    io.packed_w = packed.data_ptr()
    keep = [w, packed]
    for name, t in (("bias", bias), ("gamma", gn_weight), ("beta", gn_bias)):
        if t is not None:
            tc = t.detach().contiguous()
            keep.append(tc)
            setattr(io, name, tc.data_ptr())
    if tbias is not None:
        assert tbias.dim() == 2 and tbias.stride(1) == 1
        io.tbias, io.tbias_stride = tbias.data_ptr(), tbias.stride(0)
    if res is not None:
        io.res = res.data_ptr()
        io.res_sb, io.res_sc, io.res_sl = _strides3(res)
    io.y = y.data_ptr()
    io.y_sb, io.y_sc, io.y_sl = _strides3(y)
    io.batch = B
    if scratch is not None:
        assert scratch.is_cuda and scratch.dtype == torch.float32 and scratch.is_contiguous()
        io.scratch, io.scratch_floats = scratch.data_ptr(), scratch.numel()
        if tickets is not None:
            assert tickets.is_cuda and tickets.dtype == torch.int32 and tickets.numel() >= 256 and tickets.is_contiguous()
            io.tickets = tickets.data_ptr()
    L.check(L.lib().adx_tconv_forward(C.byref(d), C.byref(io), s), "adx_tconv_forward")
    return y


def embed(freqs, w1, b1, w3, b3, t, img_feature, rows, cond=None, cond_mlp=None):
    """time_embed [rows, dim], mish_cond [rows, 2 dim] (see adx_embed_forward in include/adx.h)."""
    dim = w3.shape[0]
    ew = L.EmbedWeights()
    keep = []
    for name, v in (("freqs", freqs), ("w1", w1), ("b1", b1), ("w3", w3), ("b3", b3)):
        v = v.detach().contiguous()
        keep.append(v)
        setattr(ew, name, v.data_ptr())
    if cond_mlp is not None:
        for name, v in zip(("cw0", "cb0", "cw2", "cb2"), cond_mlp):
            v = v.detach().contiguous()
            keep.append(v)
            setattr(ew, name, v.data_ptr())
    te = torch.empty((rows, dim), dtype=torch.float32, device=t.device)
    mc = torch.empty((rows, 2 * dim), dtype=torch.float32, device=t.device)
    t = t.contiguous()
    img_feature = img_feature.contiguous()
    if cond is not None:
        cond = cond.contiguous()
    L.check(L.lib().adx_embed_forward(C.byref(ew), dim, t.data_ptr(), t.shape[0], L.ptr(cond), img_feature.data_ptr(),
                                      img_feature.shape[0], rows, te.data_ptr(), mc.data_ptr(),
                                      L.stream_ptr(t.device)), "adx_embed_forward")
    return te, mc


def conv2d(x: torch.Tensor, weight: torch.Tensor, *, stride: int = 1, pad: int = 0, scale=None, shift=None, res=None,
           relu: bool = False, packed: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None):
    """NCHW conv2d with the fused BatchNorm(eval)/residual/ReLU epilogue of the perception encoder."""
    assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.is_contiguous()
    n, cin, h, w = x.shape
    cout, _, k, _ = weight.shape
    d = L.Conv2dDesc(cin, cout, k, stride, pad)
    s = L.stream_ptr(x.device)
    if packed is None:
        nbytes = L.lib().adx_conv2d_packed_bytes(C.byref(d))
        if nbytes == 0:
            L.check(-1, "adx_conv2d_packed_bytes")
        packed = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
        wc = weight.detach().contiguous()
        L.check(L.lib().adx_conv2d_pack(C.byref(d), wc.data_ptr(), packed.data_ptr(), s), "adx_conv2d_pack")
    oh, ow = (h + 2 * pad - k) // stride + 1, (w + 2 * pad - k) // stride + 1
    y = out if out is not None else torch.empty((n, cout, oh, ow), dtype=torch.float32, device=x.device)
    L.check(L.lib().adx_conv2d_forward(C.byref(d), x.data_ptr(), packed.data_ptr(), L.ptr(scale), L.ptr(shift),
                                       L.ptr(res), y.data_ptr(), n, h, w, int(relu), s), "adx_conv2d_forward")
    return y, packed


def to_cells(x: torch.Tensor) -> torch.Tensor:
    """fp32 [N, C, H, W] (C % 8 == 0) -> the cell layout of csrc/conv2d_hs.hip as a uint8 tensor of the same byte size:
    per image [C / 8][hi, lo][H][W] cells of eight fp16 channels, hi = fp16(x), lo = fp16((x - hi) * 2^11)."""
    n, c, h, w = x.shape
    hi = x.to(torch.float16)
    lo = ((x - hi.float()) * 2048.0).to(torch.float16)
    cells = torch.stack([hi.view(n, c // 8, 8, h, w), lo.view(n, c // 8, 8, h, w)], dim=2)      # [N, C/8, 2, 8, H, W]
    return cells.permute(0, 1, 2, 4, 5, 3).contiguous().view(torch.uint8).reshape(-1)


def from_cells(cells: torch.Tensor, shape) -> torch.Tensor:
    """Inverse of to_cells up to the split's 2^-23: hi + lo / 2^11 as fp32 [N, C, H, W]."""
    n, c, h, w = shape
    v = cells.view(torch.float16).view(n, c // 8, 2, h, w, 8).float()
    return (v[:, :, 0] + v[:, :, 1] / 2048.0).permute(0, 1, 4, 2, 3).reshape(n, c, h, w).contiguous()


def conv2d_cells(x: torch.Tensor, packed: torch.Tensor, cin: int, cout: int, n: int, h: int, w: int, *, x_cells: bool,
                 scale=None, shift=None, res=None, res_cells: bool = False, relu: bool = False,
                 out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """3x3 stride-1 pad-1 conv with a cell-layout output (and optionally cell-layout input / residual); returns the
    output cells (uint8, fp32 byte size).  Only for launches adx_conv2d_cells_supported accepts."""
    d = L.Conv2dDesc(cin, cout, 3, 1, 1)
    if not L.lib().adx_conv2d_cells_supported(C.byref(d), n, h, w):
        raise ValueError(f"conv2d_cells: {cin}->{cout} at {n}x{h}x{w} is not a plain launch of the pipelined 3x3 kernel")
    y = out if out is not None else torch.empty(n * cout * h * w * 4, dtype=torch.uint8, device=x.device)
    fmt = (1 if x_cells else 0) | 2 | (4 if (res is not None and res_cells) else 0)
    L.check(L.lib().adx_conv2d_forward_cells(C.byref(d), x.data_ptr(), packed.data_ptr(), L.ptr(scale), L.ptr(shift), L.ptr(res),
                                             y.data_ptr(), n, h, w, int(relu), fmt, L.stream_ptr(x.device)),
            "adx_conv2d_forward_cells")
    return y


def conv2d_weight_grad(x: torch.Tensor, dy: torch.Tensor, k: int, *, stride: int = 1, pad: int = 0,
                       estimate_range: bool = True) -> torch.Tensor:
    """d(loss)/d(weight) of conv2d(x, weight, stride, pad) given dy = d(loss)/d(output): [cout, cin, k, k]."""
    assert x.is_cuda and dy.is_cuda and x.dtype == dy.dtype == torch.float32 and x.is_contiguous() and dy.is_contiguous()
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    d = L.Conv2dDesc(cin, cout, k, stride, pad)
    dw = torch.empty((cout, cin, k, k), dtype=torch.float32, device=x.device)
    # the scratch carries the range partials AND, under ADX_WGRAD_DETERMINISTIC=1, the per-split copies of dW: the library
    # says how much that is (no second parse of the environment here), the range estimate is asked for explicitly
    scratch = torch.empty(L.lib().adx_conv2d_wgrad_scratch_bytes(), dtype=torch.uint8, device=x.device)
    L.check(L.lib().adx_conv2d_wgrad_ex(C.byref(d), x.data_ptr(), dy.data_ptr(), dw.data_ptr(), n, h, w, scratch.data_ptr(),
                                        int(bool(estimate_range)), L.stream_ptr(x.device)), "adx_conv2d_wgrad_ex")
    return dw


def conv2d_weight_grad_cells(x: torch.Tensor, dy: torch.Tensor) -> torch.Tensor:
    """d(loss)/d(weight) of a 3x3 stride-1 pad-1 convolution with both operands handed over as CELL tensors, the way the training
    executor holds them (csrc/resnet_train.hip): x [n, cin, h, w] and dy [n, cout, h, w] fp32 are re-laid here (`to_cells`), dy
    under the power of two that moves its maximum into [2^14, 2^15).  Returns dw [cout, cin, 3, 3]."""
    assert x.is_cuda and dy.is_cuda and x.dtype == dy.dtype == torch.float32
    n, cin, h, w = x.shape
    cout = dy.shape[1]
    amax = float(dy.abs().max())
    s = 2.0 ** (14 - math.floor(math.log2(amax))) if amax > 0 else 1.0
    scale = torch.tensor([s, 1.0 / s], dtype=torch.float32, device=x.device)
    d = L.Conv2dDesc(cin, cout, 3, 1, 1)
    dw = torch.empty((cout, cin, 3, 3), dtype=torch.float32, device=x.device)
    scratch = torch.empty(L.lib().adx_conv2d_wgrad_scratch_bytes(), dtype=torch.uint8, device=x.device)
    xc, dc = to_cells(x.contiguous()), to_cells((dy * s).contiguous())
    L.check(L.lib().adx_conv2d_wgrad_cells(C.byref(d), xc.data_ptr(), dc.data_ptr(), scale.data_ptr(), dw.data_ptr(), n, h, w,
                                           scratch.data_ptr(), L.stream_ptr(x.device)), "adx_conv2d_wgrad_cells")
    return dw


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


def image_transform(frames_u8: torch.Tensor, mean=IMAGENET_MEAN, std=IMAGENET_STD) -> torch.Tensor:
    """The agents' `T.Compose([T.ToTensor(), T.Normalize(mean, std)])` (interact.py:73-78) for uint8 camera frames
    [H, W, 3] or [N, H, W, 3] already on the GPU -> fp32 [N, 3, H, W]."""
    if frames_u8.dim() == 3:
        frames_u8 = frames_u8[None]
    if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.shape[-1] != 3:
        raise L.AdxError("image_transform expects a uint8 [N, H, W, 3] tensor on the GPU")
    f = frames_u8.contiguous()
    n, h, w, _ = f.shape
    out = torch.empty((n, 3, h, w), dtype=torch.float32, device=f.device)
    m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
    L.check(L.lib().adx_image_normalize(f.data_ptr(), out.data_ptr(), n, h, w, m, s, L.stream_ptr(f.device)),
            "adx_image_normalize")
    return out
