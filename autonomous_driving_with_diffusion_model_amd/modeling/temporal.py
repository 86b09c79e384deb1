"""`TemporalMapUnet` / `build_model` with the reference's call surface
(modeling/temporal.py:58-258), executed by libadx.so on an MI355X.

Same constructor arguments, `forward(x, img, time, cond=None, return_action_and_time_only=False)`,
attributes `.perception`, `.state_pred`, `.magic_num`, state_dict keys and parameter order as the
reference.  The nn.Module tree only HOLDS parameters (modeling/holders.py); one forward is
  perception(img)  -> adx_resnet_forward   (memoised per image tensor in eval mode, see below)
  everything else  -> adx_unet_forward     (one native call, ~45 fused kernel launches)

Perception memoisation: the reference re-runs the ResNet-34 on the same image at every denoising
step (modeling/temporal.py:203) although in eval mode the result cannot change.  In eval mode
under no_grad the feature is cached against the *identity* of the image tensor object (weak
reference + version counter + weight fingerprint), which makes the agents' sampling loops
(interact.py:131-164) pay for one perception pass per scene without touching their code.  Set
`model.cache_perception = False` for the reference-faithful per-step behaviour.  Inference tensors carry no version
counter, so an in-place refill cannot be seen: they are memoised only with `model.cache_perception = "identity"`.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import weakref
from typing import Optional, Sequence

import torch
import torch.nn as nn

from .. import _lib as L
from ..misc.constant import GuidanceType
from .holders import populate
from .perception import PerceptionResNet34
from .spec import unet_entries
from .trajpredict import TrajPredict


class _UnetTrainFn(torch.autograd.Function):
    """Temporal stack with gradients: adx_unet_forward_train keeps a tape in a per-call workspace,
    adx_unet_backward turns d(out) into one gradient per parameter plus d(img_feature)."""

    @staticmethod
    def forward(ctx, x, feat, time, cond, module, *params):
        h = module._native()
        rows = x.shape[0]
        module._ensure_packed(x.device)
        nbytes = L.lib().adx_unet_train_workspace_bytes(h, rows)
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        tape = L.NativeTape(L.lib().adx_unet_tape_create, L.lib().adx_unet_tape_destroy, "adx_unet_tape_create")
        classifier = module.use_cond == GuidanceType.CLASSIFIER_GUIDANCE
        out = torch.empty((rows, module.horizon, 3 if classifier else module.transition_dim), dtype=torch.float32,
                          device=x.device)
        te = torch.empty((rows, module.dim), dtype=torch.float32, device=x.device)
        feat_c = L.require_gpu_f32(feat.detach(), "img_feature")
        io = L.UnetIO()
        io.x, io.img_feature, io.feat_rows = x.data_ptr(), feat_c.data_ptr(), feat_c.shape[0]
        io.t, io.t_rows, io.cond, io.rows = time.data_ptr(), time.shape[0], L.ptr(cond), rows
        io.out, io.time_embed = out.data_ptr(), te.data_ptr()
        try:
            L.check(L.lib().adx_unet_forward_train(h, module._packed.data_ptr(), ws.data_ptr(), nbytes, C.byref(io),
                                                   tape.handle, L.stream_ptr(x.device)), "adx_unet_forward_train")
        except Exception:
            tape.release()
            raise
        ctx.module, ctx.tape, ctx.ws, ctx.nbytes = module, tape, ws, nbytes
        ctx.keep = (x, feat_c, time, cond)          # the tape holds raw pointers into these
        ctx.params = params
        ctx.pack_key = module._pack_key
        return out, te

    @staticmethod
    def backward(ctx, grad_out, grad_te):
        module, params = ctx.module, ctx.params
        if not ctx.tape.alive:
            raise RuntimeError("the temporal stack's tape was consumed by an earlier backward (retain_graph / a second "
                               "backward through the same forward is not supported: run the forward again)")
        if module._pack_key != ctx.pack_key:
            raise RuntimeError("parameters changed between forward and backward")
        h = module._native()
        g = L.require_gpu_f32(grad_out, "grad_out")
        rows = g.shape[0]
        gte = None if grad_te is None else L.require_gpu_f32(grad_te, "grad_time_embed")
        grads = [L.grad_buffer(p) for p in params]     # born in their communication buckets under DataParallel
        d_feat = torch.empty((rows, module.dim), dtype=torch.float32, device=g.device)
        pa = L.ptr_array([p.detach() for p in params])
        ga = L.ptr_array(grads)
        try:
            L.check(L.lib().adx_unet_backward(h, module._packed.data_ptr(), ctx.ws.data_ptr(), ctx.nbytes,
                                              ctx.tape.handle, g.data_ptr(), L.ptr(gte), d_feat.data_ptr(), pa, ga,
                                              len(grads), L.stream_ptr(g.device)), "adx_unet_backward")
        finally:
            ctx.tape.release()
            ctx.ws = None            # the tape's activations are dead with it
        return (None, d_feat, None, None, None, *grads)


class TemporalMapUnet(nn.Module):
    def __init__(self, horizon, transition_dim=2, attention=False, dim=128, dim_mults=(1, 2, 4, 8),
                 diffuser_building_block="concat", use_cond=GuidanceType.NO_GUIDANCE):
        super().__init__()
        if diffuser_building_block != "concat":
            raise NotImplementedError  # modeling/temporal.py:71-74
        if attention:
            # MODEL.USE_ATTN defaults to False and the reference's up path is broken with it
            # (temporal.py:168 builds LinearAttention(dim_out) for a dim_in tensor)
            raise NotImplementedError("USE_ATTN=True is not supported (it raises in the reference's up path too)")
        self.horizon, self.transition_dim, self.dim = int(horizon), int(transition_dim), int(dim)
        self.dim_mults = tuple(int(m) for m in dim_mults)
        self.use_cond = use_cond
        dims = [transition_dim, *[dim * m for m in self.dim_mults]]
        if int(os.environ.get("LOCAL_RANK", "-1")) <= 0:
            print(f"[ models/temporal ] Channel dimensions: {list(zip(dims[:-1], dims[1:]))}")

        entries = unet_entries(use_cond.name, self.transition_dim, self.dim, self.dim_mults)
        # registration order: perception, [cond_mlp], time_mlp, downs, ups, mid_block1, mid_block2, heads
        self.perception = PerceptionResNet34(self.dim)
        rest = [e for e in entries if not e.key.startswith("perception.")]
        unet_side = [e for e in rest if not e.key.startswith("state_pred.")]
        populate(self, unet_side)
        if use_cond == GuidanceType.CLASSIFIER_GUIDANCE:
            self.state_pred = TrajPredict(in_dim=3, out_dim=self.transition_dim - 3, pred_len=self.horizon - 1,
                                          hidden_dim=64, num_layers=2)
        self._unet_keys = [e.key for e in unet_side]
        self.magic_num = 23.315
        self._unet_param_list = None
        self.cache_perception = True
        self._handle = None
        self._packed = None
        self._pack_key = None
        self._ws = None
        self._ws_rows = 0
        self._freqs = None
        self._feat_cache = None  # (weakref(img), L.write_stamp(img), weights_key, feature)

    # -- native object management --------------------------------------------------------------
    def _native(self):
        if self._handle is None:
            cfg = L.UnetConfig()
            cfg.horizon, cfg.transition_dim, cfg.dim = self.horizon, self.transition_dim, self.dim
            cfg.n_mults = len(self.dim_mults)
            for i, m in enumerate(self.dim_mults):
                cfg.dim_mults[i] = m
            cfg.guidance = self.use_cond.value
            h = L.vp()
            L.check(L.lib().adx_unet_create(C.byref(cfg), C.byref(h)), "adx_unet_create")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().adx_unet_destroy(self._handle)
        except Exception:
            pass

    def _unet_params(self):
        # looked up once: walking the module tree on every forward was a visible part of an eagerly launched B = 1 step;
        # refresh_weights() -- called by everything that could replace a Parameter object -- drops the list
        if self._unet_param_list is None:
            named = dict(self.named_parameters())
            self._unet_param_list = [named[k] for k in self._unet_keys]
        return self._unet_param_list

    def _weights_key(self):
        ps = self._unet_params()
        return (ps[0].data_ptr(), sum(L.write_stamp(p) for p in ps))

    def refresh_weights(self):
        """Force a re-pack of the HIP weight images (needed only after out-of-band `.data` writes)."""
        self._pack_key = None
        self._unet_param_list = None
        self.perception.invalidate()
        self._feat_cache = None
        if hasattr(self, "state_pred"):
            self.state_pred.invalidate()

    def train(self, mode: bool = True):
        if mode != self.training:
            # EMAModel.copy_to / restore write through `.data` around evaluate() (train.py:307-318),
            # which version counters do not see; a mode flip always brackets them
            self.refresh_weights()
        return super().train(mode)

    def _apply(self, fn, *a, **k):
        self.refresh_weights()
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self.refresh_weights()
        return super().load_state_dict(*a, **k)

    def _ensure_packed(self, device):
        key = self._weights_key()
        if key == self._pack_key:
            return
        h = self._native()
        ps = [L.require_gpu_f32(p.detach(), "parameter") for p in self._unet_params()]
        n = L.lib().adx_unet_num_params(h)
        assert n == len(ps), (n, len(ps))
        nbytes = L.lib().adx_unet_packed_bytes(h)
        if self._packed is None or self._packed.numel() != nbytes or self._packed.device != device:
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=device)
        if self._freqs is None or self._freqs.device != device:
            # helpers.py:66-69, evaluated with torch on the host exactly as the reference does on CPU
            half = self.dim // 2
            scale = math.log(10000) / (half - 1)
            self._freqs = torch.exp(torch.arange(half) * -scale).to(device=device, dtype=torch.float32)
        L.check(L.lib().adx_unet_pack(h, L.ptr_array(ps), n, self._freqs.data_ptr(), self._packed.data_ptr(),
                                      L.stream_ptr(device)), "adx_unet_pack")
        self._pack_key = key

    # -- perception with per-image memoisation -------------------------------------------------
    def image_feature(self, img: torch.Tensor) -> torch.Tensor:
        # eval mode: the forward is a plain native call in every grad mode (no_grad, inference_mode -- train.py:53 -- or grad
        # enabled, as interact.py:147-155 calls it), its result depends on (image, weights) only
        use_cache = self.cache_perception and not self.training
        if use_cache and img.is_inference() and self.cache_perception != "identity":
            # an inference tensor (everything made under torch.inference_mode(), train.py:53) has no version counter: a caller
            # that refills one preallocated frame buffer in place would be handed the previous frame's feature.  Such tensors
            # are memoised only on request (`cache_perception = "identity"`: the caller vouches that a tensor object's
            # content never changes); by default the encoder runs, as in the reference
            use_cache = False
        if use_cache and self._feat_cache is not None:
            ref, stamp, wkey, feat = self._feat_cache
            if ref() is img and stamp == L.write_stamp(img) and wkey == self.perception.weights_key():
                return feat
        feat = self.perception(img)
        self._feat_cache = (weakref.ref(img), L.write_stamp(img), self.perception.weights_key(), feat) if use_cache else None
        return feat

    # -- training path -----------------------------------------------------------------------------
    def unet_forward_train(self, x, img_feature, time, cond=None):
        """TemporalMapUnet.forward minus the perception pass, differentiable w.r.t. every temporal-stack
        parameter and w.r.t. `img_feature` (train.py:242)."""
        x = L.require_gpu_f32(x, "x")
        rows = x.shape[0]
        time = L.require_gpu_f32(time.reshape(-1), "time", torch.int64)
        if time.shape[0] != rows or img_feature.shape[0] != rows:
            raise RuntimeError("training expects time [B] and img_feature [B, dim] for x [B, H, D]")
        cond_t = None
        if self.use_cond == GuidanceType.FREE_GUIDANCE and cond is not None:
            cond_t = L.require_gpu_f32(cond, "cond")
        out, te = _UnetTrainFn.apply(x, img_feature, time, cond_t, self, *self._unet_params())
        if self.use_cond != GuidanceType.CLASSIFIER_GUIDANCE:
            return out
        # temporal.py:233-242: the state head sees the DETACHED action but the live time_embed
        action = out
        state = self.state_pred(action.detach()[:, :-1], te)
        state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
        return torch.cat([state, action], dim=-1)

    # -- sampling loops: everything that does not depend on the trajectory, once per loop -------------
    def _workspace(self, rows: int, device):
        nbytes = L.lib().adx_unet_workspace_bytes(self._native(), rows)
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != device:
            # no initialisation needed: the ticket words of the split reductions (adx_tconv_io::tickets) at its front are
            # cleared by every adx_unet_forward itself
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        return self._ws

    @torch.no_grad()
    def time_conditioning(self, img, timesteps, cond=None, rows: Optional[int] = None) -> "TimeConditioning":
        """The time MLP, the condition MLP and the 16 per-block time Linears (modeling/temporal.py:206-216,
        modeling/helpers.py:121-123) for EVERY timestep of a sampling loop in one pass.

        Inside the reference's loop (interact.py:128-166) they are recomputed each step from (t, target, image
        feature), none of which changes during the loop: at one scene per tick that is 2 of ~50 dependent launches per
        step.  `timesteps` int64 [n]; `cond` None or [rows, 2]; `rows` = rows of one step (default: cond's, else the
        image batch).  Pass the result to `forward(..., time_cond=(tc, i))`; row-wise it equals what forward computes
        itself (same kernels, each row independent of the batch it is in)."""
        if self.training:
            raise RuntimeError("time_conditioning is an inference-loop helper (eval mode)")
        feat = self.image_feature(img)
        dev = feat.device
        free = self.use_cond == GuidanceType.FREE_GUIDANCE
        cond = L.require_gpu_f32(cond, "cond") if (cond is not None and free) else None
        if rows is None:
            rows = cond.shape[0] if cond is not None else feat.shape[0]
        if rows % feat.shape[0] != 0 or (cond is not None and tuple(cond.shape) != (rows, 2)):
            raise ValueError(f"rows {rows} vs image batch {feat.shape[0]} / cond {None if cond is None else tuple(cond.shape)}")
        ts = L.require_gpu_f32(timesteps.reshape(-1), "timesteps", torch.int64)
        n = ts.shape[0]
        total = n * rows
        # row (step s, row r) of the table: t[s], cond[r], feature[r % image batch]
        t_full = ts.repeat_interleave(rows).contiguous()
        feat_full = feat.repeat(total // feat.shape[0], 1).contiguous()
        cond_full = None if cond is None else cond.repeat(n, 1).contiguous()
        self._ensure_packed(dev)
        h = self._native()
        width = L.lib().adx_unet_time_bias_width(h)
        tb = torch.empty((n, rows, width), dtype=torch.float32, device=dev)
        te = torch.empty((n, rows, self.dim), dtype=torch.float32, device=dev)
        io = L.UnetIO()
        io.img_feature, io.feat_rows = feat_full.data_ptr(), total
        io.t, io.t_rows, io.cond, io.rows = t_full.data_ptr(), total, L.ptr(cond_full), total
        # a scratch of its own (3 * dim floats per table row), not the forward's activation workspace grown to n * rows rows
        ws = torch.empty(L.lib().adx_unet_time_conditioning_workspace_bytes(h, total), dtype=torch.uint8, device=dev)
        L.check(L.lib().adx_unet_time_conditioning(h, self._packed.data_ptr(), ws.data_ptr(),
                                                   C.byref(io), te.data_ptr(), tb.data_ptr(), L.stream_ptr(dev)),
                "adx_unet_time_conditioning")
        return TimeConditioning(tb, te, rows, self._weights_key())

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, x, img, time, cond=None, return_action_and_time_only=False, *, time_cond=None):
        """x [B, T, D]; img [B or 1, 3, H, W]; time int64 [B or 1]; cond None or [B, 2].

        time_cond = (TimeConditioning, step index): use the loop's precomputed table instead of (img, time, cond),
        which are then not read; x may be [1, T, D] for a table of more rows (every row reads the one trajectory: the
        classifier-free pair of interact.py:131 without the torch.cat)."""
        if self.training:
            feat = self.perception(img)          # train-mode perception: batch-statistics BatchNorm, autograd node
            return self.unet_forward_train(x, feat, time, cond)
        x = L.require_gpu_f32(x, "x")
        if x.dim() != 3 or x.shape[1] != self.horizon or x.shape[2] != self.transition_dim:
            raise ValueError(f"x must be [B, {self.horizon}, {self.transition_dim}], got {tuple(x.shape)}")
        if time_cond is not None:
            return self._forward_precomputed(x, time_cond, return_action_and_time_only)
        rows = x.shape[0]
        feat = self.image_feature(img)
        time = L.require_gpu_f32(time.reshape(-1), "time", torch.int64)
        free = self.use_cond == GuidanceType.FREE_GUIDANCE
        if cond is not None and free:
            cond = L.require_gpu_f32(cond, "cond")
            if tuple(cond.shape) != (rows, 2):
                raise ValueError(f"cond must be [{rows}, 2], got {tuple(cond.shape)}")
        else:
            cond = None
        if not free and (time.shape[0] != rows or feat.shape[0] != rows):
            raise RuntimeError(f"Sizes of tensors must match: time {time.shape[0]}, img {feat.shape[0]}, x {rows} "
                               "(torch.cat at modeling/temporal.py:213)")
        self._ensure_packed(x.device)
        h = self._native()
        self._workspace(rows, x.device)
        classifier = self.use_cond == GuidanceType.CLASSIFIER_GUIDANCE
        out_ch = 3 if classifier else self.transition_dim
        out = torch.empty((rows, self.horizon, out_ch), dtype=torch.float32, device=x.device)
        te = torch.empty((rows, self.dim), dtype=torch.float32, device=x.device) if classifier else None
        io = L.UnetIO()
        io.x, io.img_feature, io.feat_rows = x.data_ptr(), feat.data_ptr(), feat.shape[0]
        io.t, io.t_rows, io.cond, io.rows = time.data_ptr(), time.shape[0], L.ptr(cond), rows
        io.out, io.time_embed = out.data_ptr(), L.ptr(te)
        L.check(L.lib().adx_unet_forward(h, self._packed.data_ptr(), self._ws.data_ptr(), C.byref(io),
                                         L.stream_ptr(x.device)), "adx_unet_forward")
        return self._finish(out, te, return_action_and_time_only)

    def _finish(self, out, te, return_action_and_time_only):
        if self.use_cond != GuidanceType.CLASSIFIER_GUIDANCE:
            return out
        action = out
        if return_action_and_time_only:
            return action, te
        # temporal.py:238-242: state from the detached action, dummy zero first row, concat
        state = self.state_pred(action.detach()[:, :-1], te)
        state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
        return torch.cat([state, action], dim=-1)

    def _forward_precomputed(self, x, time_cond, return_action_and_time_only):
        tc, i = time_cond
        rows = tc.rows
        if x.shape[0] != rows and x.shape[0] != 1:
            raise ValueError(f"x has {x.shape[0]} rows; the conditioning table was made for {rows} (or pass one row)")
        if not 0 <= i < tc.time_bias.shape[0]:
            raise IndexError(f"step {i} outside the conditioning table of {tc.time_bias.shape[0]} steps")
        if i == 0:        # once per loop: fingerprinting 196 parameters is a visible part of an eagerly launched step
            if tc.weights_key != self._weights_key():
                raise RuntimeError("the model's weights changed after time_conditioning() was computed")
            self._ensure_packed(x.device)
        classifier = self.use_cond == GuidanceType.CLASSIFIER_GUIDANCE
        out = torch.empty((rows, self.horizon, 3 if classifier else self.transition_dim), dtype=torch.float32, device=x.device)
        io = L.UnetIO()
        io.x, io.x_rows, io.rows, io.out = x.data_ptr(), x.shape[0], rows, out.data_ptr()
        io.time_bias = tc.time_bias.data_ptr() + i * tc.step_bytes
        L.check(L.lib().adx_unet_forward(self._native(), self._packed.data_ptr(), self._workspace(rows, x.device).data_ptr(),
                                         C.byref(io), L.stream_ptr(x.device)), "adx_unet_forward")
        return self._finish(out, tc.time_embed[i] if classifier else None, return_action_and_time_only)


class TimeConditioning:
    """Per-loop table of `TemporalMapUnet.time_conditioning`: time_bias [n_steps, rows, sum of block widths],
    time_embed [n_steps, rows, dim]."""

    def __init__(self, time_bias, time_embed, rows, weights_key):
        self.time_bias, self.time_embed, self.rows, self.weights_key = time_bias, time_embed, rows, weights_key
        self.step_bytes = time_bias.stride(0) * time_bias.element_size()     # one step's [rows, width] slice


def build_model(cfg) -> TemporalMapUnet:
    """modeling/temporal.py:248-258."""
    return TemporalMapUnet(
        horizon=cfg.MODEL.HORIZON,
        transition_dim=cfg.MODEL.TRANSITION_DIM,
        attention=cfg.MODEL.USE_ATTN,
        dim=cfg.MODEL.DIM,
        dim_mults=cfg.MODEL.DIM_MULTS,
        diffuser_building_block=cfg.MODEL.DIFFUSER_BUILDING_BLOCK,
        use_cond=GuidanceType[cfg.TRAIN.USE_COND],
    )
