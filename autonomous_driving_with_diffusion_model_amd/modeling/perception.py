"""ResNet-34 camera encoder (reference: modeling/resnet.py:163-296,325-333 with the fc replaced
by Linear(512, dim) at modeling/temporal.py:83-84), executed by libadx.so.

Eval mode: BatchNorm uses running statistics, applied as scale/shift in the conv epilogue.
Train mode: batch statistics, running buffers updated in place (momentum 0.1, num_batches_tracked
incremented), differentiable w.r.t. every parameter through `_PerceptionTrainFn`.
"""
from __future__ import annotations

import ctypes as C
import os

import torch
import torch.nn as nn

from .. import _lib as L
from .holders import populate
from .spec import resnet34_entries


class _PerceptionTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, module, *params):
        h = module._native()
        B, _, H, W = img.shape
        ts = [L.require_gpu_f32(t.detach(), "perception tensor") for t in module._tensors()]
        nbytes = L.lib().adx_resnet_train_workspace_bytes(h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"image {H}x{W} too small for ResNet-34")
        ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)
        pk = L.lib().adx_resnet_packed_bytes(h)
        packed = torch.empty(pk, dtype=torch.uint8, device=img.device)
        tape = L.NativeTape(L.lib().adx_resnet_tape_create, L.lib().adx_resnet_tape_destroy, "adx_resnet_tape_create")
        out = torch.empty((B, module.out_dim), dtype=torch.float32, device=img.device)
        try:
            L.check(L.lib().adx_resnet_forward_train(h, L.ptr_array(ts), len(ts), packed.data_ptr(), ws.data_ptr(), nbytes,
                                                     img.data_ptr(), B, H, W, out.data_ptr(), tape.handle, 1,
                                                     L.stream_ptr(img.device)), "adx_resnet_forward_train")
        except Exception:
            tape.release()
            raise
        # num_batches_tracked of the 36 BatchNorm layers: one multi-tensor add, not 36 launches
        counters = [b for b in module.buffers() if b.dtype == torch.int64]
        if counters:
            torch._foreach_add_(counters, 1)
        module.invalidate()                 # running statistics moved: eval-mode image is stale
        ctx.module, ctx.tape, ctx.ws, ctx.nbytes, ctx.ts, ctx.img = module, tape, ws, nbytes, ts, img
        return out

    @staticmethod
    def backward(ctx, grad_out):
        module = ctx.module
        if not ctx.tape.alive:
            raise RuntimeError("the perception tape was consumed by an earlier backward (retain_graph / a second backward "
                               "through the same forward is not supported: run the forward again)")
        g = L.require_gpu_f32(grad_out, "grad_out")
        named = dict(module.named_parameters())
        entries = [e for e in module._entries if e.dtype == "f32"]
        grads, slots = {}, []
        for e in entries:
            if e.is_buffer:
                slots.append(None)
            else:
                grads[e.key] = L.grad_buffer(named[e.key])      # born in its communication bucket under DataParallel
                slots.append(grads[e.key])
        garr = (L.vp * len(slots))()
        for i, t in enumerate(slots):
            garr[i] = None if t is None else t.data_ptr()
        # Completion events, one per group of layers in the order the backward produces their gradients (fc, the BasicBlocks
        # from layer4 down, the stem): a parameter carries the event of its group until its gradient has been consumed
        # (`_adx_grad_event`), so that parallel.GradientAverager can start a bucket's reduction when ITS gradients exist instead
        # of when this whole call -- 85 of the model's 149 MB of gradients -- has run (train.py:176-178, 251: what DDP's reducer
        # does per parameter while accelerator.backward(loss) runs).
        events = module._backward_events(g.device)
        earr, n_ev = None, 0
        if events is not None:
            n_ev = len(events)
            earr = (L.vp * n_ev)(*[ev.cuda_event for ev in events])
        try:
            L.check(L.lib().adx_resnet_backward_events(module._native(), L.ptr_array(ctx.ts), garr, len(slots), ctx.ws.data_ptr(),
                                                       ctx.nbytes, ctx.tape.handle, g.data_ptr(), earr, n_ev, L.stream_ptr(g.device)),
                    "adx_resnet_backward_events")
            if events is not None:
                for i, e in enumerate(entries):
                    grp = module._tensor_groups[i]
                    if not e.is_buffer and grp >= 0:
                        # the group's event covers this gradient only where the native call wrote it INTO the bucket view; a
                        # buffer of its own (`.grad` already present: accumulation over several backwards; the view lent to
                        # another node of the graph) reaches the bucket through AccumulateGrad's add, which is queued on the
                        # compute stream after this whole call -- tagged (grp, None): "no event covers this one"
                        p = named[e.key]
                        view = getattr(p, "_adx_grad_view", None)
                        in_view = view is not None and slots[i].data_ptr() == view.data_ptr()
                        p._adx_grad_event = (grp, events[grp] if in_view else None)
        finally:
            ctx.tape.release()
            ctx.ws = None            # 27 GB of taped activations at B = 64: free them with the tape
        return (None, None, *[grads[k] for k, _ in module.named_parameters()])


class PerceptionResNet34(nn.Module):
    def __init__(self, out_dim: int):
        super().__init__()
        self.out_dim = out_dim
        self._entries = resnet34_entries("", out_dim)
        self._tensor_list = None
        populate(self, self._entries)
        self._handle = None
        self._packed = None
        self._pack_key = None
        self._ws = None
        # eval passes of >= 16 images run on a stream of their own and may run ahead of the caller's queued work when the image
        # is the one of the previous pass (_on_pass_stream).  `.run_ahead`: "frozen" (default) -- only inside a `frozen_image`
        # context, where the caller vouches that nothing writes the image; "version" -- also outside, on identity + autograd's
        # version counter (the caller vouches that the image is only ever written through torch ops on ITS tensor object: a
        # write through `img.data`, DLPack, or a raw data_ptr() moves no counter); False / ADX_PERCEPTION_AHEAD=0 -- never
        env = os.environ.get("ADX_PERCEPTION_AHEAD", "frozen")
        self.run_ahead = False if env == "0" else ("version" if env == "version" else "frozen")
        self._pass_stream = None
        self._pass_seen = None
        self._pack_gen = 0            # moves whenever _ensure_packed re-lays the weight images (their key can repeat)
        self._frozen = None           # (weakref(img), token) while a frozen_image context is open

    # -- native object management ------------------------------------------------------------
    def _native(self):
        if self._handle is None:
            h = L.vp()
            L.check(L.lib().adx_resnet_create(self.out_dim, C.byref(h)), "adx_resnet_create")
            self._handle = h
        return self._handle

    def __del__(self):
        try:
            if self._handle is not None:
                L.lib().adx_resnet_destroy(self._handle)
        except Exception:
            pass

    def _backward_events(self, device):
        """One timing-enabled event per gradient group of adx_resnet_backward_events (created once per device; each is recorded
        once here so that its native handle exists -- torch creates it lazily), or None where torch does not expose the handle."""
        cached = getattr(self, "_bwd_events", None)
        if cached is not None and cached[0] == device:
            return cached[1]
        h = self._native()
        n = L.lib().adx_resnet_backward_groups(h)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(n)]
        st = torch.cuda.current_stream(device)
        for ev in evs:
            ev.record(st)
        if not all(getattr(ev, "cuda_event", 0) for ev in evs):
            evs = None
        n_f32 = sum(1 for e in self._entries if e.dtype == "f32")
        self._tensor_groups = [L.lib().adx_resnet_tensor_group(h, i) for i in range(n_f32)]
        self._bwd_events = (device, evs)
        return evs

    def _tensors(self):
        # the tensor OBJECTS are looked up once (walking the module tree twice per call was a visible part of an eagerly
        # launched B = 1 step); nn.Module keeps them across .to() / load_state_dict() / optimizer steps, and everything that
        # could replace them (_apply, load_state_dict, invalidate) drops the list
        if self._tensor_list is None:
            sd = dict(self.named_parameters())
            sd.update(dict(self.named_buffers()))
            self._tensor_list = [sd[e.key] for e in self._entries if e.dtype == "f32"]
        return self._tensor_list

    def _apply(self, fn, *a, **k):
        self._tensor_list = None
        return super()._apply(fn, *a, **k)

    def load_state_dict(self, *a, **k):
        self._tensor_list = None
        return super().load_state_dict(*a, **k)

    def weights_key(self):
        ts = self._tensors()
        return (ts[0].data_ptr(), sum(L.write_stamp(t) for t in ts), self.training)

    def invalidate(self):
        self._pack_key = None
        self._tensor_list = None

    def _ensure_packed(self):
        key = self.weights_key()
        if key == self._pack_key:
            return
        h = self._native()
        ts = [L.require_gpu_f32(t.detach(), "perception tensor") for t in self._tensors()]
        n = L.lib().adx_resnet_num_tensors(h)
        assert n == len(ts), (n, len(ts))
        nbytes = L.lib().adx_resnet_packed_bytes(h)
        if self._packed is None or self._packed.numel() != nbytes or self._packed.device != ts[0].device:
            self._packed = torch.empty(nbytes, dtype=torch.uint8, device=ts[0].device)
        arr = L.ptr_array(ts)
        L.check(L.lib().adx_resnet_pack(h, arr, n, self._packed.data_ptr(), L.stream_ptr(ts[0].device)),
                "adx_resnet_pack")
        self._pack_key = key
        # the re-lay was queued on the caller's stream: the next pass must join it, even when the new key EQUALS the one it
        # remembers (native running-statistics writes move no version counter: train-mode forward, no optimizer step, eval)
        self._pack_gen += 1
        self._pass_seen = None

    def forward_frames(self, frames_u8: torch.Tensor, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)) -> torch.Tensor:
        """Eval-mode forward straight from uint8 camera frames [N, H, W, 3] (or [H, W, 3]): the agents'
        `T.Compose([T.ToTensor(), T.Normalize(mean, std)])` (interact.py:73-78, e2e_driving/diffusion_agent.py:96-101)
        runs inside the stem kernel's staging load, so the normalised fp32 image is never written or re-read.
        Bit-identical to `self(ops.image_transform(frames_u8, mean, std))`."""
        if self.training:
            raise RuntimeError("forward_frames is the inference front-end (eval mode)")
        if frames_u8.dim() == 3:
            frames_u8 = frames_u8[None]
        if not frames_u8.is_cuda or frames_u8.dtype != torch.uint8 or frames_u8.dim() != 4 or frames_u8.shape[-1] != 3:
            raise L.AdxError("forward_frames expects a uint8 [N, H, W, 3] tensor on the GPU")
        f = frames_u8.contiguous()
        self._ensure_packed()
        h = self._native()
        B, H, W, _ = f.shape
        nbytes = L.lib().adx_resnet_workspace_bytes(h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"image {H}x{W} too small for ResNet-34")
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != f.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=f.device)
        m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)

        def run():
            out = torch.empty((B, self.out_dim), dtype=torch.float32, device=f.device)
            L.check(L.lib().adx_resnet_forward_u8(h, self._packed.data_ptr(), self._ws.data_ptr(), f.data_ptr(), m, s, B, H, W,
                                                  out.data_ptr(), L.stream_ptr(f.device)), "adx_resnet_forward_u8")
            return out
        return self._on_pass_stream(f, run)

    def forward(self, img: torch.Tensor) -> torch.Tensor:
        img = L.require_gpu_f32(img, "img")
        if img.dim() != 4 or img.shape[1] != 3:
            raise ValueError(f"img must be [B, 3, H, W], got {tuple(img.shape)}")
        if self.training:
            return _PerceptionTrainFn.apply(img, self, *[p for _, p in self.named_parameters()])
        self._ensure_packed()
        h = self._native()
        B, _, H, W = img.shape
        nbytes = L.lib().adx_resnet_workspace_bytes(h, B, H, W)
        if nbytes == 0:
            raise ValueError(f"image {H}x{W} too small for ResNet-34")
        if self._ws is None or self._ws.numel() < nbytes or self._ws.device != img.device:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=img.device)

        def run():
            out = torch.empty((B, self.out_dim), dtype=torch.float32, device=img.device)
            L.check(L.lib().adx_resnet_forward(h, self._packed.data_ptr(), self._ws.data_ptr(), img.data_ptr(), B, H, W,
                                               out.data_ptr(), L.stream_ptr(img.device)), "adx_resnet_forward")
            return out
        return self._on_pass_stream(img, run)

    # -- the eval pass on a stream of its own --------------------------------------------------------------------------
    def frozen_image(self, img: torch.Tensor):
        """Context manager: the caller vouches that NOTHING writes `img` (by any route: torch ops, `.data`, raw pointers,
        other streams) while the context is open -- what a sampling loop knows about the image of its tick (the reference's
        loops pass the same, untouched tensor to every denoising step: interact.py:133-155, train.py:80-88).  Inside, eval
        passes on that tensor object may run ahead of the caller's queued work (_on_pass_stream); the first pass of a context
        always joins the caller's stream (the image's producer), and every pass is joined by it afterwards, so writes queued
        after the context has closed are ordered behind the last pass."""
        import contextlib
        import weakref

        @contextlib.contextmanager
        def ctx():
            prev = self._frozen
            self._frozen = (weakref.ref(img), object())
            try:
                yield self
            finally:
                self._frozen = prev
        return ctx()

    def _on_pass_stream(self, img: torch.Tensor, run):
        """Large eval passes run on a stream of their own, which the caller's stream joins when the pass is done.  What that
        buys: a pass on an image that is KNOWN to be the one of the previous pass, unwritten -- the reference's sampling loop
        runs the encoder in every denoising step on the image of the tick (modeling/temporal.py:203) -- needs nothing of what
        the caller has queued since (that step's temporal stack and scheduler step), so it runs AHEAD of it: the previous
        step's ~30 small latency-bound launches execute beside this step's encoder instead of in front of it.

        "Known" is the caller's statement, not a guess: inside `frozen_image(img)` (this package's loops, bench.py), or with
        `run_ahead = "version"` on identity + version counter.  Everything else -- another tensor object, a moved version
        counter, re-laid weights (pack generation), another workspace, an inference tensor outside a frozen context, a stream
        capture, fewer than 16 images, a pass that bypassed this stream in between -- joins the caller's stream first, as any
        launch would.  A write the caller did not declare (through `img.data`, a raw pointer) is therefore only ever missed
        where the caller said there is none (tests/test_gpu_model.py::test_perception_pass_sees_undeclared_image_writes)."""
        dev = img.device
        cur = torch.cuda.current_stream(dev)
        mode = "version" if self.run_ahead is True else self.run_ahead
        if not mode or img.shape[0] < 16 or torch.cuda.is_current_stream_capturing():
            self._pass_seen = None       # this pass uses the workspace on the caller's stream: the next one must join it
            return run()
        if self._pass_stream is None or self._pass_stream.device != dev:
            self._pass_stream = torch.cuda.Stream(device=dev)
            self._pass_seen = None
        ps = self._pass_stream
        seen = self._pass_seen
        token = self._frozen[1] if (self._frozen is not None and self._frozen[0]() is img) else None
        stamp = None if img.is_inference() else img._version
        same = (seen is not None and seen[0]() is img and seen[2] == self._pack_gen and seen[3] == self._ws.data_ptr()
                and ((token is not None and seen[4] is token)
                     or (mode == "version" and stamp is not None and seen[1] == stamp)))
        if not same:
            ps.wait_stream(cur)          # the image's producer, the weight images' re-lay, whoever used the workspace before
            self._ws.record_stream(ps)   # allocated on the caller's stream, used on this one
            self._packed.record_stream(ps)
        with torch.cuda.stream(ps):
            out = run()
        cur.wait_stream(ps)
        out.record_stream(cur)           # allocated on the pass stream, consumed on the caller's
        img.record_stream(ps)
        import weakref
        self._pass_seen = (weakref.ref(img), stamp, self._pack_gen, self._ws.data_ptr(), token)
        return out
