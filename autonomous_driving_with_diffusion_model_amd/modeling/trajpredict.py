"""`TrajPredict` state head used by classifier guidance (reference: modeling/helpers.py:22-59),
executed by libadx.so (csrc/trajpred.hip).

Parameter holder with the reference's keys (input_proj, encoder_traj.layers.N.*, encoder_traj.norm,
output_proj).  `forward(x, time_embed)` is an autograd node whose backward returns the gradient
w.r.t. `x` (the action), which is what `GuidanceLoss` differentiates (control/guidance.py:47-50).
In eval mode (guidance) there is no dropout and only d(action) is produced; in train mode the node applies the
encoder layers' dropout (regenerable hash masks) and also returns every parameter gradient and d(time_embed).
"""
from __future__ import annotations

import ctypes as C
import math

import torch
import torch.nn as nn

from .. import _lib as L
from .holders import populate
from .spec import traj_predict_entries


class _TrajPredictTrainFn(torch.autograd.Function):
    """Training node: gradients w.r.t. every state_pred parameter, the action and time_embed."""

    @staticmethod
    def forward(ctx, action, time_embed, module, *params):
        action_c = L.require_gpu_f32(action.detach(), "action")
        te = L.require_gpu_f32(time_embed.detach(), "time_embed")
        B, T, _ = action_c.shape
        out = torch.empty((B, T, module.out_dim), dtype=torch.float32, device=action_c.device)
        h, packed = module._ensure_packed(action_c.device, B, T)
        ctx.drop_p, ctx.seed = float(module.dropout_p), module._next_seed()
        L.check(L.lib().adx_trajpred_forward_train(h, packed.data_ptr(), action_c.data_ptr(), action_c.stride(0),
                                                   action_c.stride(1), te.data_ptr(), out.data_ptr(), B, T, ctx.drop_p,
                                                   ctx.seed, L.stream_ptr(action_c.device)), "adx_trajpred_forward_train")
        ctx.module, ctx.params = module, params
        ctx.needs_action = action.requires_grad
        ctx.save_for_backward(action_c, te)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        action_c, te = ctx.saved_tensors
        module, params = ctx.module, ctx.params
        B, T, _ = action_c.shape
        g = L.require_gpu_f32(grad_out, "grad_out")
        h, packed = module._ensure_packed(g.device, B, T)
        image = torch.empty(packed.numel() // 4, dtype=torch.float32, device=g.device)
        ga = torch.empty((B, T, 3), dtype=torch.float32, device=g.device) if ctx.needs_action else None
        dte = torch.empty((B, module.hidden_dim), dtype=torch.float32, device=g.device)
        L.check(L.lib().adx_trajpred_backward_params(h, packed.data_ptr(), action_c.data_ptr(), action_c.stride(0),
                                                     action_c.stride(1), te.data_ptr(), g.data_ptr(), L.ptr(ga),
                                                     image.data_ptr(), dte.data_ptr(), B, T, ctx.drop_p, ctx.seed,
                                                     L.stream_ptr(g.device)),
                "adx_trajpred_backward_params")
        offs = module._param_offsets()
        grads = [L.grad_buffer(p).copy_(image[o:o + p.numel()].view_as(p)) for o, p in zip(offs, params)]
        return (ga, dte, None, *grads)


class _TrajPredictFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, action, time_embed, module):
        action_c = L.require_gpu_f32(action.detach(), "action")
        te = L.require_gpu_f32(time_embed.detach(), "time_embed")
        B, T, _ = action_c.shape
        out = torch.empty((B, T, module.out_dim), dtype=torch.float32, device=action_c.device)
        h, packed = module._ensure_packed(action_c.device, B, T)
        L.check(L.lib().adx_trajpred_forward(h, packed.data_ptr(), action_c.data_ptr(), action_c.stride(0),
                                             action_c.stride(1), te.data_ptr(), out.data_ptr(), B, T,
                                             L.stream_ptr(action_c.device)), "adx_trajpred_forward")
        ctx.module = module
        ctx.save_for_backward(action_c, te)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        action_c, te = ctx.saved_tensors
        module = ctx.module
        B, T, _ = action_c.shape
        g = L.require_gpu_f32(grad_out, "grad_out")
        ga = torch.empty((B, T, 3), dtype=torch.float32, device=g.device)
        h, packed = module._ensure_packed(g.device, B, T)
        L.check(L.lib().adx_trajpred_backward(h, packed.data_ptr(), action_c.data_ptr(), action_c.stride(0),
                                              action_c.stride(1), te.data_ptr(), g.data_ptr(), ga.data_ptr(), B, T,
                                              L.stream_ptr(g.device)), "adx_trajpred_backward")
        return ga, None, None


class TrajPredict(nn.Module):
    def __init__(self, in_dim: int = 3, out_dim: int = 3, pred_len: int = 16, hidden_dim: int = 256,
                 num_heads: int = 4, num_layers: int = 3):
        super().__init__()
        if (in_dim, hidden_dim, num_heads, num_layers) != (3, 64, 4, 2):
            raise NotImplementedError("TrajPredict kernels are built for in_dim 3, hidden 64, 4 heads, 2 layers "
                                      "(the only configuration TemporalMapUnet instantiates, temporal.py:187-189)")
        self.in_dim, self.out_dim, self.pred_len = in_dim, out_dim, pred_len
        self.hidden_dim, self.num_heads, self.num_layers = hidden_dim, num_heads, num_layers
        self._entries = traj_predict_entries("", in_dim, out_dim, hidden_dim, num_layers)
        populate(self, self._entries, init_prefix="state_pred.")
        self._handle = None
        self._packed = None
        self._ws = None
        self._pack_key = None
        self._freqs = None
        # nn.TransformerEncoderLayer's default (modeling/helpers.py:35-41 does not override it); train mode only
        self.dropout_p = 0.1
        self._calls = 0

    def _next_seed(self) -> int:
        """64-bit key of this call's dropout masks: torch's global seed (so torch.manual_seed reproduces a run) mixed
        with a per-module call counter; no device round trip."""
        self._calls += 1
        x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._calls * 0xD1B54A32D192ED03) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        return (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF

    def _native(self):
        if self._handle is None:
            h = L.vp()
            L.check(L.lib().adx_trajpred_create(self.out_dim, C.byref(h)), "adx_trajpred_create")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().adx_trajpred_destroy(self._handle)
        except Exception:
            pass

    def invalidate(self):
        self._pack_key = None
        self._param_list = None

    def _params(self):
        # looked up once (see TemporalMapUnet._unet_params); invalidate() / _apply / load_state_dict drop the list
        if getattr(self, "_param_list", None) is None:
            named = dict(self.named_parameters())
            self._param_list = [named[e.key] for e in self._entries]
        return self._param_list

    def _apply(self, fn, *a, **k):
        self._param_list = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._param_list = None
        return super().load_state_dict(*a, **k)

    def _ensure_packed(self, device, batch=0, T=0):
        """The native handle and the packed weight image; for sequences of 32..63 rows also lends the handle the scratch
        its 64-row kernels need for `batch` samples (`_ws`: GraphedSampler watches its address like the other workspaces)."""
        ps = self._params()
        key = (ps[0].data_ptr(), sum(L.write_stamp(p) for p in ps))
        h = self._native()
        if key != self._pack_key or self._packed is None or self._packed.device != device:
            ts = [L.require_gpu_f32(p.detach(), "state_pred parameter") for p in ps]
            n = L.lib().adx_trajpred_num_params(h)
            assert n == len(ts), (n, len(ts))
            self._packed = torch.empty(L.lib().adx_trajpred_packed_bytes(h), dtype=torch.uint8, device=device)
            half = self.hidden_dim // 2
            self._freqs = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1))).to(device=device,
                                                                                           dtype=torch.float32)
            L.check(L.lib().adx_trajpred_pack(h, L.ptr_array(ts), n, self._freqs.data_ptr(), self._packed.data_ptr(),
                                              L.stream_ptr(device)), "adx_trajpred_pack")
            self._pack_key = key
        need = L.lib().adx_trajpred_scratch_bytes(h, batch, T)
        if need and (self._ws is None or self._ws.device != device or self._ws.numel() < need):
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
            L.check(L.lib().adx_trajpred_set_scratch(h, self._ws.data_ptr(), self._ws.numel()), "adx_trajpred_set_scratch")
        return h, self._packed

    def _param_offsets(self):
        if getattr(self, "_offs", None) is None:
            h = self._native()
            n = L.lib().adx_trajpred_num_params(h)
            arr = (L.i64 * n)()
            L.check(L.lib().adx_trajpred_param_offsets(h, arr, n), "adx_trajpred_param_offsets")
            self._offs = list(arr)
        return self._offs

    def forward(self, x: torch.Tensor, time_embed: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3 or x.shape[2] != self.in_dim or x.shape[1] > 63:
            raise ValueError(f"x must be [B, T <= 63, {self.in_dim}], got {tuple(x.shape)}")
        if self.training:
            # train mode applies nn.TransformerEncoderLayer's dropout (self.dropout_p, 0.1 like the reference) with
            # regenerable hash masks; they follow the same distribution as torch's but not its Philox stream
            named = dict(self.named_parameters())
            return _TrajPredictTrainFn.apply(x, time_embed, self, *[named[e.key] for e in self._entries])
        return _TrajPredictFn.apply(x, time_embed, self)

    def guided_output(self, action: torch.Tensor, time_embed: torch.Tensor, target: torch.Tensor, model_std: float,
                      scale: float) -> torch.Tensor:
        """Fused interact.py:153-160 + GuidanceLoss(STEP=1) + TargetGuidance for every sample:
        returns the guided, clipped model output [B, H, out_dim + 3]."""
        a = L.require_gpu_f32(action.detach(), "action")
        te = L.require_gpu_f32(time_embed.detach(), "time_embed")
        B, H, _ = a.shape
        tg = L.require_gpu_f32(target.reshape(-1, 2).expand(B, 2) if target.numel() == 2 else target, "target")
        out = torch.empty((B, H, self.out_dim + 3), dtype=torch.float32, device=a.device)
        h, packed = self._ensure_packed(a.device, B, H - 1)
        L.check(L.lib().adx_guided_output(h, packed.data_ptr(), a.data_ptr(), te.data_ptr(), tg.data_ptr(),
                                          float(model_std), float(scale), out.data_ptr(), None, B, H - 1,
                                          L.stream_ptr(a.device)), "adx_guided_output")
        return out
