"""Data-parallel pieces of the training loop (reference: accelerate -> DistributedDataParallel, train.py:
115-117,176-178,251).

One process per GPU; `torch.distributed` backend "nccl" is RCCL on ROCm.  The model's gradients arrive
through ordinary autograd, so `torch.nn.parallel.DistributedDataParallel(model)` works as in the
reference.  `GradientAverager` is the explicit equivalent used by this package's own trainer: gradients
are flattened into a few large buckets (fewer, larger collectives suit xGMI's per-link bandwidth) in
REVERSE registration order, i.e. temporal-stack buckets first — with `attach()` their all-reduce is launched
from autograd hooks as soon as the temporal backward has produced them and overlaps the perception backward
(two thirds of the step); `average()` is the plain after-backward form.  Sampling needs no collective: scenes are sharded with `shard_range`.
"""
from __future__ import annotations

from typing import Iterable, List, Optional, Sequence, Tuple

import os

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced shard [lo, hi) of n_items scenes for `rank` (first n % world ranks get one more)."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError(f"bad rank {rank} / world_size {world_size}")
    q, r = divmod(n_items, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def make_buckets(sizes: Sequence[int], bucket_elems: int) -> List[List[int]]:
    """Greedy buckets over indices in REVERSE order (last-registered parameters first = gradients that
    become ready first)."""
    buckets, cur, cur_n = [], [], 0
    for i in reversed(range(len(sizes))):
        if cur and cur_n + sizes[i] > bucket_elems:
            buckets.append(cur)
            cur, cur_n = [], 0
        cur.append(i)
        cur_n += sizes[i]
    if cur:
        buckets.append(cur)
    return buckets


class GradientAverager:
    """Bucketed gradient mean over the ranks, reduced IN PLACE.

    Every bucket is one flat tensor and every parameter owns a view into it (`param._adx_grad_view`).  This package's
    backward nodes (temporal stack, perception, TrajPredict) write a parameter's gradient straight into that view when
    the parameter has no gradient yet (`_lib.grad_buffer`), autograd adopts the view as `.grad` without copying, and the
    collective then runs on the bucket itself: no copy-in, no copy-out (2 x 149 MB of HBM traffic per step before).  A
    gradient that lives elsewhere (a foreign module's, or one accumulated over several backwards) is copied in and out
    as before -- the result is the same, only slower.

    primitive  "all_reduce" (one collective per bucket) or "reduce_scatter" (reduce_scatter_tensor into this rank's
               slice of the bucket + all_gather_into_tensor back, both in place: the explicit two-phase form SURVEY 8e
               prefers on a fully connected xGMI node; buckets are padded to a multiple of the world size).
    mean       True: `.grad` holds the mean after `synchronize()` (ReduceOp.AVG where the backend has it -- probed once --,
               else SUM and one scaling pass).
               False: `.grad` holds the SUM and `grad_scale` = 1 / world is left to the consumer --
               `FusedAdamWEMA(grad_scale=...)` folds it into its single pass over the gradients."""

    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, group=None,
                 primitive: str = "all_reduce", mean: bool = True, force: Optional[bool] = None):
        if primitive not in ("all_reduce", "reduce_scatter"):
            raise ValueError(f"primitive must be 'all_reduce' or 'reduce_scatter', got {primitive!r}")
        self.params = [p for p in params if p.requires_grad]
        self.group, self.primitive, self.mean = group, primitive, mean
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        # force (or ADX_FORCE_COLLECTIVES=1): issue every collective on a ONE-rank group too.  One-rank collectives are
        # identities, so the gradients do not change -- but the real RCCL kernels then run on the side stream behind the
        # per-group events while backward is still running: the whole transport path on the one GPU a box has
        # (tests/test_gpu_parallel.py::test_one_rank_rccl_behind_the_events).
        if force is None:
            force = os.environ.get("ADX_FORCE_COLLECTIVES", "0") == "1"
        if force and not dist.is_initialized():
            raise RuntimeError("GradientAverager(force=True) needs an initialised process group (a one-rank group will do)")
        self.active = self.world > 1 or bool(force)
        self.grad_scale = 1.0 if mean else 1.0 / self.world
        self.buckets = make_buckets([p.numel() for p in self.params], int(bucket_mb * 1024 * 1024 / 4))
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._views: List[list] = [[] for _ in self.buckets]
        self._hooks: list = []
        self._pending: list = []
        self._ready: List[int] = []
        self._seen_tagged: List[bool] = []       # per bucket: a gradient with a completion event has been accumulated ...
        self._untagged_late: List[bool] = []     # ... and one without an event after it (then no event covers the bucket)
        self.copied_in = 0            # gradients of the last reduction that did not live in their bucket (diagnostic)
        self.trace = False            # True: device events around every bucket's collective (see overlap_report)
        self._trace_events: list = []
        self._comm_stream = None
        self._avg_ok = True
        if self.active:
            for bi in range(len(self.buckets)):
                self._ensure_bucket(bi)
            if mean and self.params:
                self._avg_ok = self._probe_avg(self.params[0].device)

    def _probe_avg(self, device) -> bool:
        """ReduceOp.AVG exists in RCCL / NCCL >= 2.10 (every ROCm build torch ships with) and not in gloo: decided from the backend's
        name, without a trial collective (a collective that raises can leave an RCCL communicator in an error state, and a
        transient failure would silently turn every step into SUM + one more pass).  Only an unknown backend is probed, and the
        fallback is logged."""
        backend = str(dist.get_backend(self.group)).lower()
        if "nccl" in backend:
            return True
        if "gloo" in backend:
            return False
        try:
            t = torch.ones(1, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group)
            ok = bool(abs(float(t) - 1.0) < 1e-6)
        except (RuntimeError, ValueError, NotImplementedError):
            ok = False
        if not ok:
            import warnings
            warnings.warn(f"GradientAverager: backend {backend!r} has no ReduceOp.AVG; buckets carry the SUM and are scaled after the wait")
        return ok

    def _ensure_bucket(self, bi: int) -> torch.Tensor:
        idxs = self.buckets[bi]
        p0 = self.params[idxs[0]]
        flat = self._flat[bi]
        if flat is not None and flat.device == p0.device:
            return flat
        n = sum(self.params[i].numel() for i in idxs)
        padded = (n + self.world - 1) // self.world * self.world       # reduce_scatter needs equal slices
        flat = self._flat[bi] = torch.zeros(padded, dtype=p0.dtype, device=p0.device)
        views, off = [], 0
        for i in idxs:
            p = self.params[i]
            v = flat[off:off + p.numel()].view(p.shape)
            p._adx_grad_view = v          # the backward nodes write here (see _lib.grad_buffer)
            p._adx_grad_leased = False
            views.append(v)
            off += p.numel()
        self._views[bi] = views
        return flat

    def attach(self) -> "GradientAverager":
        """Overlap mode: a bucket's collective is launched from autograd hooks the moment its last gradient has been
        accumulated, i.e. DURING backward (the temporal stack's gradients are complete before the perception backward
        -- two thirds of the step -- has even been queued).  Call `synchronize()` after `loss.backward()`."""
        if self._hooks:
            return self
        self._where = {}
        for bi, idxs in enumerate(self.buckets):
            for i in idxs:
                self._where[i] = bi
        self._ready = [0] * len(self.buckets)
        self._seen_tagged = [False] * len(self.buckets)
        self._untagged_late = [False] * len(self.buckets)
        self._pending = []
        for i, p in enumerate(self.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        return self

    def _make_hook(self, i: int):
        def hook(_param):
            _param._adx_grad_leased = False        # accumulated: the bucket view may be lent again once .grad is cleared
            if not self.active:
                return
            bi = self._where[i]
            tag = getattr(_param, "_adx_grad_event", None)
            if tag is not None and tag[1] is not None:
                self._seen_tagged[bi] = True
            elif tag is not None or self._seen_tagged[bi]:
                # a gradient of the event-recording call that was NOT written into its bucket view by that call (`.grad` already
                # existed, the view was lent to another node: AccumulateGrad's add runs on the compute stream after the whole
                # call), or an untagged gradient behind a tagged one: no event covers the bucket
                self._untagged_late[bi] = True
            self._ready[bi] += 1
            if self._ready[bi] == len(self.buckets[bi]):
                self._ready[bi] = 0
                with torch.no_grad():
                    self._pending.append(self._launch(bi))
        return hook

    def detach(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        for p in self.params:
            p._adx_grad_leased = False

    @torch.no_grad()
    def synchronize(self) -> None:
        """Overlap mode: wait for the collectives the hooks launched (and write back what had to be copied in)."""
        if not self.active:
            return
        if any(self._ready):
            raise RuntimeError("a bucket is incomplete: every rank must produce every gradient in every backward")
        self.finish(self._pending)
        self._pending = []

    @torch.no_grad()
    def _launch(self, bi: int):
        idxs = self.buckets[bi]
        flat = self._ensure_bucket(bi)
        copied = []
        for i, v in zip(idxs, self._views[bi]):
            self.params[i]._adx_grad_leased = False      # (the no-hook form: average() after backward)
            g = self.params[i].grad
            if g is None:
                raise RuntimeError("a parameter has no gradient (every rank must produce every gradient)")
            if g.data_ptr() != v.data_ptr() or g.device != v.device:      # not born in the bucket: bring it in
                v.copy_(g)
                copied.append(i)
        op = dist.ReduceOp.AVG if (self.mean and self._avg_ok) else dist.ReduceOp.SUM

        def issue():
            if self.primitive == "reduce_scatter":
                shard = flat.view(self.world, -1)[self.rank]
                w1 = dist.reduce_scatter_tensor(shard, flat, op=op, group=self.group, async_op=True)
                # The all-gather must not start before the reduce-scatter has written this rank's slice.  That order is
                # EXPLICIT, not left to the backend's issue order: on a device backend `Work.wait()` makes the issuing (side)
                # stream depend on the reduce-scatter and the all-gather's stream picks that dependency up when it is issued;
                # on gloo it blocks the host until the worker thread is done.
                w1.wait()
                return [w1, dist.all_gather_into_tensor(flat, shard, group=self.group, async_op=True)]
            return [dist.all_reduce(flat, op=op, group=self.group, async_op=True)]

        if not flat.is_cuda:
            return (issue(), bi, copied)
        # Issued from a side stream that first joins the compute stream (the bucket's gradients were written there), so the
        # compute stream -- still running the rest of backward -- is never held up by a collective; it joins in finish().
        cur = torch.cuda.current_stream(flat.device)
        if self._comm_stream is None or self._comm_stream.device != flat.device:
            self._comm_stream = torch.cuda.Stream(device=flat.device)
        ready = done = None
        # When every gradient of the bucket carries the completion event of the launch that wrote it (the perception backward
        # is ONE native call whose layer groups finish one after the other: modeling/perception.py), the side stream waits for
        # the LAST of those events -- the bucket is reduced while the compute stream is still differentiating the layers below.
        # Otherwise it joins everything queued on the compute stream so far (the temporal stack's per-layer nodes).
        tags = [getattr(self.params[i], "_adx_grad_event", None) for i in idxs]
        tagged = [t for t in tags if t is not None and t[1] is not None]
        for i in idxs:
            self.params[i]._adx_grad_event = None            # an event belongs to the backward that recorded it
        # gradients of the bucket without an event are covered by the last event as long as they were accumulated BEFORE the
        # first tagged one (their kernels precede it on the compute stream); _make_hook keeps that book per bucket.  A
        # gradient that had to be copied into the bucket just now (`copied`: the copy was queued on the compute stream a
        # moment ago) or one the recording call did not write into its view (tag without an event) is covered by no event.
        late = self._untagged_late[bi] if bi < len(self._untagged_late) else True
        late = late or bool(copied) or any(t is not None and t[1] is None for t in tags)
        if bi < len(self._untagged_late):
            self._untagged_late[bi] = self._seen_tagged[bi] = False
        last = max(tagged, key=lambda t: t[0]) if tagged and not late else None
        if last is not None:
            self._comm_stream.wait_event(last[1])
            ready = last[1] if self.trace else None
        else:
            if self.trace:
                ready = torch.cuda.Event(enable_timing=True)
                ready.record(cur)
            self._comm_stream.wait_stream(cur)
        with torch.cuda.stream(self._comm_stream):
            works = issue()
            if self.trace:
                works[-1].wait()
                done = torch.cuda.Event(enable_timing=True)
                done.record(self._comm_stream)
                self._trace_events.append((bi, flat.numel() * flat.element_size(), ready, done))
        return (works, bi, copied)

    def overlap_report(self, backward_end: "torch.cuda.Event") -> list:
        """With `trace = True` during one backward: per bucket, when its last gradient was ready and when its collective had
        finished, in ms RELATIVE TO THE END OF BACKWARD on the device (`backward_end` = an event recorded on the compute stream
        right after `loss.backward()` returned; negative = before).  A collective that finishes before 0 was fully hidden
        behind backward.  Call after `synchronize()` and a device sync."""
        out = []
        for bi, nbytes, ready, done in self._trace_events:
            out.append({"bucket": bi, "mb": round(nbytes / 1e6, 1), "ready_ms": round(-ready.elapsed_time(backward_end), 3),
                        "done_ms": round(-done.elapsed_time(backward_end), 3)})
        self._trace_events = []
        return out

    @torch.no_grad()
    def average(self, async_op: bool = False):
        """Reduce every bucket after backward (no hooks).  Returns the list of pending reductions when async_op."""
        if not self.active:
            return []
        works = [self._launch(bi) for bi in range(len(self.buckets))]
        if async_op:
            return works
        self.finish(works)
        return []

    @torch.no_grad()
    def finish(self, works) -> None:
        self.copied_in = 0
        for handles, bi, copied in works:
            for h in handles:
                h.wait()
            if self.mean and not self._avg_ok:
                self._flat[bi].mul_(1.0 / self.world)
            self.copied_in += len(copied)
            where = dict(zip(self.buckets[bi], self._views[bi]))
            for i in copied:
                self.params[i].grad.copy_(where[i])


class BufferBroadcaster:
    """DDP's `broadcast_buffers=True` (the reference's setting: accelerate builds `DistributedDataParallel(model)` with
    torch's defaults, train.py:176-178): before EVERY training forward rank 0's buffers overwrite everybody's, i.e. the
    BatchNorm running statistics of ranks > 0 are discarded each step and rank 0's -- computed from rank 0's batches
    only, never averaged -- are what a checkpoint holds.  108 buffers, 68 KB for the ResNet-34: they travel as ONE flat
    fp32 broadcast plus one int64 broadcast (num_batches_tracked) instead of 108 collectives."""

    def __init__(self, module: torch.nn.Module, src: int = 0, group=None):
        self.module, self.src, self.group = module, src, group
        bufs = list(module.buffers())
        self._float = [b for b in bufs if b.is_floating_point()]
        self._int = [b for b in bufs if not b.is_floating_point()]
        self._flat_f: Optional[torch.Tensor] = None
        self._flat_i: Optional[torch.Tensor] = None

    @torch.no_grad()
    def sync(self) -> int:
        """Returns the number of collectives issued (0 on a single rank)."""
        if not dist.is_initialized() or dist.get_world_size(self.group) == 1:
            return 0
        n = 0
        for bufs, attr, dtype in ((self._float, "_flat_f", torch.float32), (self._int, "_flat_i", torch.int64)):
            if not bufs:
                continue
            total = sum(b.numel() for b in bufs)
            flat = getattr(self, attr)
            if flat is None or flat.numel() != total or flat.device != bufs[0].device:
                flat = torch.empty(total, dtype=dtype, device=bufs[0].device)
                setattr(self, attr, flat)
            if dist.get_rank(self.group) == self.src:
                torch.cat([b.reshape(-1).to(dtype) for b in bufs], out=flat)
            dist.broadcast(flat, src=self.src, group=self.group)
            n += 1
            if dist.get_rank(self.group) != self.src:
                off = 0
                for b in bufs:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()
        # No refresh_weights() here: the copy_ above moves the buffers' version counters, which is what the eval-mode weight
        # images are keyed on (PerceptionResNet34.weights_key), so they are rebuilt at the next eval forward -- and ranks
        # > 0 do no host work per step that rank 0 does not (every collective waits for the slowest rank).
        return n


class DataParallel(torch.nn.Module):
    """This package's counterpart of `DistributedDataParallel(model)` as accelerate builds it (train.py:176-178):
    construction broadcasts rank 0's parameters and buffers, every train-mode forward starts with the buffer broadcast
    (`BufferBroadcaster`), gradients are averaged bucket by bucket from autograd hooks while backward is still running
    (`GradientAverager.attach`).  After `loss.backward()` call `.synchronize()` (DDP does that inside backward; here it
    is one explicit call so that the collective's wait sits where the trainer wants it), then the optimizer step.
    `.module` is the wrapped model, like DDP's attribute that `accelerator.unwrap_model` reads."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64.0, broadcast_buffers: bool = True, group=None,
                 primitive: str = "all_reduce", optimizer=None, force: Optional[bool] = None):
        """optimizer: a `FusedAdamWEMA` over the same parameters.  The buckets then carry the SUM and the 1 / world goes
        into the optimizer's single pass over the gradients (`optimizer.grad_scale`); `.grad` between `synchronize()` and
        `optimizer.step()` is the sum, not the mean.  Without it `.grad` holds the mean, as under DDP."""
        super().__init__()
        self.module = module
        broadcast_parameters(module, src=0, group=group)
        self.buffers_sync = BufferBroadcaster(module, src=0, group=group) if broadcast_buffers else None
        self.averager = GradientAverager(module.parameters(), bucket_mb=bucket_mb, group=group, primitive=primitive,
                                         mean=optimizer is None, force=force).attach()
        if optimizer is not None:
            optimizer.grad_scale = self.averager.grad_scale

    def forward(self, *args, **kwargs):
        if self.buffers_sync is not None and self.module.training and torch.is_grad_enabled():
            self.buffers_sync.sync()
        return self.module(*args, **kwargs)

    def synchronize(self) -> None:
        self.averager.synchronize()


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """What DDP's constructor does (train.py:176-178): rank `src`'s parameters and buffers win."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            dist.broadcast(t, src=src, group=group)
    if hasattr(module, "refresh_weights"):
        module.refresh_weights()
