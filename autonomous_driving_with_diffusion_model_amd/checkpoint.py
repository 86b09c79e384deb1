"""Checkpoint interop with the reference.

The reference saves `{"state_dict", "optimizer", "lr_scheduler", "iter", "ema_state_dict"}` with `torch.save`
(train.py:283-299).  Two readers exist:
  * the agents (interact.py:102-106, e2e_driving/diffusion_agent.py): `load_state_dict(ckpt["state_dict"])` by key, then
    the parameters are overwritten POSITIONALLY with `ckpt["ema_state_dict"]["shadow_params"]` (misc/load_param.py:4-8);
  * train.py's resume (train.py:191-203): `EMAModel.load_state_dict`, `AdamW.load_state_dict`,
    `LambdaLR.load_state_dict` on the three other entries.
Key names and parameter order of this package's model equal the reference's (tests/golden/state_spec.json), so the
reference's `.pth` files load here unchanged.  Files written here satisfy both readers: `optimizer` is a
`torch.optim.AdamW` state dict, `lr_scheduler` a `LambdaLR` state dict (`lr_lambdas: [None]`), `ema_state_dict` has the
diffusers-0.28 `EMAModel` keys; an entry that cannot be written in its reader's layout is refused rather than stored as
a placeholder.  `resume_training` is the inverse for `FusedAdamWEMA`.
"""
from __future__ import annotations

from typing import Optional

import torch

from .misc.load_param import copy_parameters


def _lambda_lr_state(iteration: int, base_lr: float, last_lr: float, ticks_per_step: int = 1) -> dict:
    ticks = int(iteration) * ticks_per_step
    return {"base_lrs": [base_lr], "last_epoch": ticks, "_step_count": ticks + 1, "_is_initial": False,
            "_get_lr_called_within_step": False, "_last_lr": [last_lr], "lr_lambdas": [None]}


def save_checkpoint(path: str, model: torch.nn.Module, optimizer, iteration: int, shadow_params=None,
                    lr_scheduler=None) -> None:
    """Write the reference's 5-key dict.

    `optimizer`: a `FusedAdamWEMA` (moments, LR schedule and EMA all come from it) or a torch optimizer.  With a torch
    optimizer pass the `lr_scheduler` object too (its `state_dict()` is stored, as train.py:291 does); without one the
    schedule entry is derived from the optimizer's param group (constant LR)."""
    if optimizer is None:
        raise ValueError("save_checkpoint needs the optimizer: train.py's resume calls AdamW.load_state_dict on the "
                         "entry, which rejects a placeholder (use torch.save({'state_dict': ...}) for weights only)")
    ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "optimizer": None,
            "lr_scheduler": None, "iter": int(iteration), "ema_state_dict": None}     # train.py:288-294 key order
    ema = None
    if hasattr(optimizer, "lr_scheduler_state_dict"):           # FusedAdamWEMA
        ckpt["optimizer"] = _to_cpu(optimizer.state_dict())
        ckpt["lr_scheduler"] = optimizer.lr_scheduler_state_dict()
        if optimizer.use_ema:
            ema = optimizer.ema_state_dict()
    else:
        ckpt["optimizer"] = _to_cpu(optimizer.state_dict())
        if lr_scheduler is not None:
            ckpt["lr_scheduler"] = lr_scheduler.state_dict()
        else:
            g = optimizer.param_groups[0]
            ckpt["lr_scheduler"] = _lambda_lr_state(iteration, g.get("initial_lr", g["lr"]), g["lr"])
    if shadow_params is not None or ema is None:
        src = shadow_params if shadow_params is not None else [p.detach() for p in model.parameters()]
        base = ema or {"decay": 0.9999, "min_decay": 0.0, "update_after_step": 0, "use_ema_warmup": False,
                       "inv_gamma": 1.0, "power": 2 / 3}
        ema = dict(base, optimization_step=int(iteration), shadow_params=[s.detach().clone() for s in src])
    ema["shadow_params"] = [s.detach().cpu() for s in ema["shadow_params"]]
    ckpt["ema_state_dict"] = ema
    torch.save(ckpt, path)


def _to_cpu(obj):
    if torch.is_tensor(obj):
        return obj.detach().cpu()
    if isinstance(obj, dict):
        return {k: _to_cpu(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_to_cpu(v) for v in obj)
    return obj


def load_checkpoint(path: str, model: torch.nn.Module, use_ema: bool = True, map_location="cpu") -> dict:
    """interact.py:102-106: load_state_dict by key, then the positional EMA copy.  The file holds tensors, lists, dicts
    and Python scalars only, so it is read with `weights_only=True`."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ckpt["state_dict"])
    if use_ema and "ema_state_dict" in ckpt and ckpt["ema_state_dict"].get("shadow_params"):
        copy_parameters(ckpt["ema_state_dict"]["shadow_params"], model.parameters())
    if hasattr(model, "refresh_weights"):
        model.refresh_weights()
    return ckpt


def resume_training(path: str, model: torch.nn.Module, optimizer, map_location="cpu") -> int:
    """train.py:191-203 for a `FusedAdamWEMA`: weights by key (NOT the EMA copy: training continues from the raw
    weights), moments + step count, LR-schedule position, EMA shadow parameters.  Returns the next iteration."""
    ckpt = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(ckpt["state_dict"])
    if hasattr(model, "refresh_weights"):
        model.refresh_weights()
    optimizer.load_state_dict(ckpt["optimizer"])
    optimizer.load_lr_scheduler_state_dict(ckpt["lr_scheduler"])
    if optimizer.use_ema:
        optimizer.load_ema_state_dict(ckpt["ema_state_dict"])
    return int(ckpt["iter"]) + 1
