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
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.warmup_steps, self.lr_ticks_per_step = warmup_steps, lr_ticks_per_step
        self.use_ema, self.sanitize = use_ema, sanitize_grads
        self.ema_kw = dict(update_after_step=ema_update_after_step, inv_gamma=ema_inv_gamma, power=ema_power,
                           max_decay=ema_max_decay, use_ema_warmup=True)
        self.step_count = 0
        self.grad_scale = 1.0           # parallel.DataParallel(optimizer=self) sets 1 / world: the buckets carry the sum
        self.exp_avg = [torch.zeros_like(p) for p in self.params]
        self.exp_avg_sq = [torch.zeros_like(p) for p in self.params]
        self.shadow_params = [p.detach().clone() for p in self.params] if use_ema else [None] * len(self.params)
        self._native_ready = False      # the launch tables are built on the first step(): state round trips work on any device

    _RING = 3

    def _prepare_native(self) -> None:
        dev = self.params[0].device
        if dev.type != "cuda":
            raise L.AdxError("FusedAdamWEMA.step needs parameters on the GPU (no CPU path)")
        chunk = L.lib().adx_optim_chunk()
        bt, bc = [], []
        for i, p in enumerate(self.params):
            nb = (p.numel() + chunk - 1) // chunk
            bt += [i] * nb
            bc += list(range(nb))
        self._block_tensor = torch.tensor(bt, dtype=torch.int32, device=dev)
        self._block_chunk = torch.tensor(bc, dtype=torch.int32, device=dev)
        self._n_blocks = len(bt)
        # Pointer table [n_params, 6] = (param, grad, exp_avg, exp_avg_sq, shadow, numel).  The host copy lives in a ring
        # of pinned buffers: an asynchronous H2D copy reads its pinned source when the stream gets there, which can be
        # more than one optimizer step after the host queued it, so a slot is rewritten only after the event recorded
        # behind its last copy has completed.  When no pointer moved since the previous step (gradients kept allocated,
        # zero_grad(set_to_none=False)) the device table is still valid and nothing is copied.
        n = len(self.params)
        self._table_ring = [torch.empty((n, 6), dtype=torch.int64).pin_memory() for _ in range(self._RING)]
        self._ring_events = [None] * self._RING
        self._ring_pos = 0
        self._table_dev = torch.empty((n, 6), dtype=torch.int64, device=dev)
        self._table_key = None
        self._native_ready = True

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
        if not self._native_ready:
            self._prepare_native()
        lr = self.current_lr()
        self.step_count += 1
        grads = [p.grad if p.grad.is_contiguous() else p.grad.contiguous() for p in self.params]
        cols = ([p.data_ptr() for p in self.params], [g.data_ptr() for g in grads],
                [t.data_ptr() for t in self.exp_avg], [t.data_ptr() for t in self.exp_avg_sq],
                [t.data_ptr() for t in self.shadow_params] if self.use_ema else [0] * len(self.params))
        key = tuple(map(tuple, cols))
        if key != self._table_key:
            slot = self._ring_pos
            self._ring_pos = (slot + 1) % self._RING
            if self._ring_events[slot] is not None:
                self._ring_events[slot].synchronize()      # the copy that last read this pinned slot has finished
            tn = self._table_ring[slot].numpy()            # one vectorised write per column (per-element indexing: ~2 us each)
            for c, col in enumerate(cols):
                tn[:, c] = col
            tn[:, 5] = [p.numel() for p in self.params]
            self._table_dev.copy_(self._table_ring[slot], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.params[0].device))
            self._ring_events[slot] = ev
            self._table_key = key
        self._keep_grads = grads                           # contiguous copies (if any) must outlive the launch
        decay = ema_decay(self.step_count, **self.ema_kw) if self.use_ema else 0.0
        L.check(L.lib().adx_adamw_ema_step_scaled(self._table_dev.data_ptr(), self._block_tensor.data_ptr(),
                                                  self._block_chunk.data_ptr(), self._n_blocks, lr, self.betas[0],
                                                  self.betas[1], self.eps, self.weight_decay, self.step_count, decay,
                                                  int(self.use_ema), int(self.sanitize), float(self.grad_scale),
                                                  L.stream_ptr(self.params[0].device)), "adx_adamw_ema_step")
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

    # -- checkpoint state, in the layouts the reference's objects read and write (train.py:197-201,288-294) ---------
    def ema_state_dict(self) -> dict:
        """diffusers 0.28 `EMAModel.state_dict()` layout (the keys `EMAModel.load_state_dict` reads; interact.py:104 uses
        only `shadow_params`, positionally)."""
        kw = self.ema_kw
        return {"decay": kw["max_decay"], "min_decay": 0.0, "optimization_step": self.step_count,
                "update_after_step": kw["update_after_step"], "use_ema_warmup": kw["use_ema_warmup"],
                "inv_gamma": kw["inv_gamma"], "power": kw["power"],
                "shadow_params": [s.detach().clone() for s in self.shadow_params] if self.use_ema else []}

    def load_ema_state_dict(self, sd: dict) -> None:
        shadow = sd.get("shadow_params") or []
        if self.use_ema:
            if len(shadow) != len(self.params):
                raise ValueError(f"ema_state_dict has {len(shadow)} shadow_params, the model has {len(self.params)} parameters")
            with torch.no_grad():
                for dst, src in zip(self.shadow_params, shadow):
                    if tuple(dst.shape) != tuple(src.shape):
                        raise ValueError(f"shadow parameter shape {tuple(src.shape)} != {tuple(dst.shape)}")
                    dst.copy_(src)
        for k_src, k_dst in (("decay", "max_decay"), ("update_after_step", "update_after_step"),
                             ("use_ema_warmup", "use_ema_warmup"), ("inv_gamma", "inv_gamma"), ("power", "power")):
            if sd.get(k_src) is not None:
                self.ema_kw[k_dst] = sd[k_src]

    def state_dict(self) -> dict:
        """`torch.optim.AdamW.state_dict()` layout: `AdamW(model.parameters(), ...).load_state_dict()` accepts it."""
        n = len(self.params)
        state = {i: {"step": torch.tensor(float(self.step_count)), "exp_avg": self.exp_avg[i].detach().clone(),
                     "exp_avg_sq": self.exp_avg_sq[i].detach().clone()} for i in range(n)} if self.step_count > 0 else {}
        group = {"lr": self.current_lr(), "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "decoupled_weight_decay": True, "initial_lr": self.lr, "params": list(range(n))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: dict) -> None:
        """Restore the moments and the step count from a `torch.optim.AdamW` state dict (the reference's checkpoints,
        train.py:200) or from `state_dict()` above."""
        groups = sd.get("param_groups") or []
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.params):
            raise ValueError(f"optimizer state covers {len(ids)} parameters, this optimizer has {len(self.params)}")
        g0 = groups[0]
        self.lr = float(g0.get("initial_lr", g0["lr"]))
        self.betas, self.eps = tuple(g0["betas"]), float(g0["eps"])
        self.weight_decay = float(g0["weight_decay"])
        state, steps = sd.get("state", {}), set()
        with torch.no_grad():
            for pos, pid in enumerate(ids):
                st = state.get(pid)
                if st is None:
                    self.exp_avg[pos].zero_()
                    self.exp_avg_sq[pos].zero_()
                    continue
                if tuple(st["exp_avg"].shape) != tuple(self.params[pos].shape):
                    raise ValueError(f"exp_avg of parameter {pid} has shape {tuple(st['exp_avg'].shape)}")
                self.exp_avg[pos].copy_(st["exp_avg"])
                self.exp_avg_sq[pos].copy_(st["exp_avg_sq"])
                steps.add(int(float(st["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter step counts differ: {sorted(steps)}")
        self.step_count = steps.pop() if steps else 0

    def lr_scheduler_state_dict(self) -> dict:
        """`LambdaLR.state_dict()` layout of get_constant_schedule_with_warmup (train.py:171,201): `lr_lambdas` holds
        None for a plain function, which `LambdaLR.load_state_dict` pops and skips."""
        ticks = self.step_count * self.lr_ticks_per_step
        return {"base_lrs": [self.lr], "last_epoch": ticks, "_step_count": ticks + 1, "_is_initial": False,
                "_get_lr_called_within_step": False, "_last_lr": [self.current_lr()], "lr_lambdas": [None]}

    def load_lr_scheduler_state_dict(self, sd: dict) -> None:
        ticks = int(sd["last_epoch"])
        if ticks % self.lr_ticks_per_step:
            raise ValueError(f"last_epoch {ticks} is not a multiple of lr_ticks_per_step {self.lr_ticks_per_step}")
        if ticks // self.lr_ticks_per_step != self.step_count:
            raise ValueError(f"lr schedule is at optimizer step {ticks // self.lr_ticks_per_step}, the moments at {self.step_count}")
