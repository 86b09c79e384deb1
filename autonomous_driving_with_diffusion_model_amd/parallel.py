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
    def __init__(self, params: Iterable[torch.nn.Parameter], bucket_mb: float = 64.0, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = make_buckets([p.numel() for p in self.params], int(bucket_mb * 1024 * 1024 / 4))
        self._flat: List[Optional[torch.Tensor]] = [None] * len(self.buckets)
        self._hooks: list = []
        self._pending: list = []
        self._ready: List[int] = []

    def attach(self) -> "GradientAverager":
        """Overlap mode: a bucket's all-reduce is launched from autograd hooks the moment its last gradient has been
        accumulated, i.e. DURING backward (the temporal stack's gradients are complete before the perception backward
        -- two thirds of the step -- has even been queued).  Call `synchronize()` after `loss.backward()`."""
        if self._hooks:
            return self
        self._where = {}
        for bi, idxs in enumerate(self.buckets):
            for i in idxs:
                self._where[i] = bi
        self._ready = [0] * len(self.buckets)
        self._pending = []
        for i, p in enumerate(self.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        return self

    def _make_hook(self, i: int):
        def hook(_param):
            if self.world == 1:
                return
            bi = self._where[i]
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

    @torch.no_grad()
    def synchronize(self) -> None:
        """Overlap mode: wait for the collectives the hooks launched and write the means back."""
        if self.world == 1:
            return
        if any(self._ready):
            raise RuntimeError("a bucket is incomplete: every rank must produce every gradient in every backward")
        self.finish(self._pending)
        self._pending = []

    @torch.no_grad()
    def _launch(self, bi: int):
        idxs = self.buckets[bi]
        grads = [self.params[i].grad for i in idxs]
        if any(g is None for g in grads):
            raise RuntimeError("a parameter has no gradient (every rank must produce every gradient)")
        n = sum(g.numel() for g in grads)
        flat = self._flat[bi]
        if flat is None or flat.numel() != n or flat.device != grads[0].device:
            flat = self._flat[bi] = torch.empty(n, dtype=grads[0].dtype, device=grads[0].device)
        off = 0
        for g in grads:
            flat[off:off + g.numel()].copy_(g.reshape(-1))
            off += g.numel()
        return (dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True), bi)

    @torch.no_grad()
    def average(self, async_op: bool = False):
        """All-reduce (mean) every gradient after backward.  Returns the list of work handles when async_op."""
        if self.world == 1:
            return []
        works = [self._launch(bi) for bi in range(len(self.buckets))]
        if async_op:
            return works
        self.finish(works)
        return []

    @torch.no_grad()
    def finish(self, works) -> None:
        for work, bi in works:
            work.wait()
            flat = self._flat[bi]
            flat.div_(self.world)
            off = 0
            for i in self.buckets[bi]:
                g = self.params[i].grad
                g.copy_(flat[off:off + g.numel()].view_as(g))
                off += g.numel()


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
        if n and dist.get_rank(self.group) != self.src and hasattr(self.module, "refresh_weights"):
            self.module.refresh_weights()       # eval-mode weight images fold the running statistics
        return n


class DataParallel(torch.nn.Module):
    """This package's counterpart of `DistributedDataParallel(model)` as accelerate builds it (train.py:176-178):
    construction broadcasts rank 0's parameters and buffers, every train-mode forward starts with the buffer broadcast
    (`BufferBroadcaster`), gradients are averaged bucket by bucket from autograd hooks while backward is still running
    (`GradientAverager.attach`).  After `loss.backward()` call `.synchronize()` (DDP does that inside backward; here it
    is one explicit call so that the collective's wait sits where the trainer wants it), then the optimizer step.
    `.module` is the wrapped model, like DDP's attribute that `accelerator.unwrap_model` reads."""

    def __init__(self, module: torch.nn.Module, bucket_mb: float = 64.0, broadcast_buffers: bool = True, group=None):
        super().__init__()
        self.module = module
        broadcast_parameters(module, src=0, group=group)
        self.buffers_sync = BufferBroadcaster(module, src=0, group=group) if broadcast_buffers else None
        self.averager = GradientAverager(module.parameters(), bucket_mb=bucket_mb, group=group).attach()

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
