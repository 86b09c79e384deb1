"""Fused AdamW + EMA + gradient sanitising for the training loop (reference: train.py:146-153,170-174,
252-261: torch.optim.AdamW(lr, betas=(0.95, 0.999), eps=1e-7) + diffusers EMAModel + nan_to_num on
every gradient + get_constant_schedule_with_warmup).  One HIP launch per step over all parameters.

The reference's own `torch.optim.AdamW` / `EMAModel` also work on this package's model (gradients
arrive through ordinary autograd); this class is the single-pass alternative.
"""
from __future__ import annotations

from typing import Iterable, List, Optional

import numpy as np
import torch

from . import _lib as L


def ema_decay(optimization_step: int, update_after_step: int = 0, inv_gamma: float = 1.0, power: float = 2 / 3,
              min_decay: float = 0.0, max_decay: float = 0.9999, use_ema_warmup: bool = True) -> float:
    """diffusers.training_utils.EMAModel.get_decay."""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    cur = 1 - (1 + step / inv_gamma) ** -power if use_ema_warmup else (1 + step) / (10 + step)
    return max(min(cur, max_decay), min_decay)


class FusedAdamWEMA:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-4, betas=(0.95, 0.999), eps: float = 1e-7,
                 weight_decay: float = 0.01, warmup_steps: int = 0, use_ema: bool = True, ema_max_decay: float = 0.9999,
                 ema_inv_gamma: float = 1.0, ema_power: float = 0.75, ema_update_after_step: int = 5000,
                 sanitize_grads: bool = True, lr_ticks_per_step: int = 1):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.AdxError("FusedAdamWEMA needs parameters on the GPU (no CPU path)")
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.warmup_steps, self.lr_ticks_per_step = warmup_steps, lr_ticks_per_step
        self.use_ema, self.sanitize = use_ema, sanitize_grads
        self.ema_kw = dict(update_after_step=ema_update_after_step, inv_gamma=ema_inv_gamma, power=ema_power,
                           max_decay=ema_max_decay, use_ema_warmup=True)
        self.step_count = 0
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.shadow_params = [p.detach().clone() for p in self.params] if use_ema else [None] * len(self.params)
        chunk = L.lib().adx_optim_chunk()
        bt, bc = [], []
        for i, p in enumerate(self.params):
            nb = (p.numel() + chunk - 1) // chunk
            bt += [i] * nb
            bc += list(range(nb))
        self._block_tensor = torch.tensor(bt, dtype=torch.int32, device=dev)
        self._block_chunk = torch.tensor(bc, dtype=torch.int32, device=dev)
        self._n_blocks = len(bt)
        self._table_host = torch.empty((len(self.params), 6), dtype=torch.int64).pin_memory()
        self._table_dev = torch.empty((len(self.params), 6), dtype=torch.int64, device=dev)

    def current_lr(self) -> float:
        """get_constant_schedule_with_warmup; accelerate ticks the schedule `num_processes` times per
        optimizer step, hence lr_ticks_per_step."""
        k = self.step_count * self.lr_ticks_per_step
        return self.lr * (min(1.0, k / max(1.0, self.warmup_steps)) if self.warmup_steps > 0 else 1.0)

    @torch.no_grad()
    def step(self) -> None:
        for p in self.params:
            if p.grad is None:
                raise RuntimeError("a parameter has no gradient (DDP find_unused_parameters=False contract)")
        lr = self.current_lr()
        self.step_count += 1
        # pointer table through numpy (one vectorised write per column; per-element tensor indexing costs ~2 us each)
        tn = self._table_host.numpy()
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in self.params]
        tn[:, 0] = [p.data_ptr() for p in self.params]
        tn[:, 1] = [g.data_ptr() for g in grads]
        tn[:, 2] = [t.data_ptr() for t in self.exp_avg]
        tn[:, 3] = [t.data_ptr() for t in self.exp_avg_sq]
        tn[:, 4] = [t.data_ptr() for t in self.shadow_params] if self.use_ema else 0
        tn[:, 5] = [p.numel() for p in self.params]
        th = self._table_host
        self._table_dev.copy_(th, non_blocking=True)
        decay = ema_decay(self.step_count, **self.ema_kw) if self.use_ema else 0.0
        L.check(L.lib().adx_adamw_ema_step(self._table_dev.data_ptr(), self._block_tensor.data_ptr(),
                                           self._block_chunk.data_ptr(), self._n_blocks, lr, self.betas[0], self.betas[1],
                                           self.eps, self.weight_decay, self.step_count, decay, int(self.use_ema),
                                           int(self.sanitize), L.stream_ptr(self.params[0].device)), "adx_adamw_ema_step")
        # the kernel wrote through raw pointers: move the version counters so the model re-packs its weights
        bump = getattr(torch.autograd.graph, "increment_version", None)
        for p in self.params:
            if bump is not None:
                bump(p)
            else:
                p.add_(0)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for p in self.params:
            if set_to_none:
                p.grad = None
            elif p.grad is not None:
                p.grad.zero_()

    def ema_state_dict(self) -> dict:
        """Same key the reference checkpoints carry (train.py:288-294, interact.py:104)."""
        return {"shadow_params": [s.detach().clone() for s in self.shadow_params], "optimization_step": self.step_count}
