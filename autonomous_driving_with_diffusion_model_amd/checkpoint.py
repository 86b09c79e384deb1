"""Checkpoint interop with the reference.

The reference saves `{"state_dict", "optimizer", "lr_scheduler", "iter", "ema_state_dict"}` with
`torch.save` (train.py:283-299) and the agents load `state_dict` by key and then overwrite the
parameters POSITIONALLY with `ema_state_dict["shadow_params"]` (interact.py:102-106,
misc/load_param.py:4-8).  Key names and parameter order of this package's model equal the
reference's, so its `.pth` files load unchanged and files written here load in the reference.
"""
from __future__ import annotations

from typing import Optional

import torch

from .misc.load_param import copy_parameters


def save_checkpoint(path: str, model: torch.nn.Module, optimizer=None, iteration: int = 0,
                    shadow_params=None, lr_scheduler_state: Optional[dict] = None) -> None:
    """Write the reference's 5-key dict.  `optimizer` may be a torch optimizer (state_dict() is stored) or a
    FusedAdamWEMA (its moments are stored in torch.optim.AdamW's state layout)."""
    ckpt = {"state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()}, "iter": int(iteration),
            "lr_scheduler": lr_scheduler_state or {"last_epoch": int(iteration)}}
    if optimizer is not None and hasattr(optimizer, "exp_avg"):
        n = len(optimizer.params)
        ckpt["optimizer"] = {
            "state": {i: {"step": torch.tensor(float(optimizer.step_count)), "exp_avg": optimizer.exp_avg[i].cpu(),
                          "exp_avg_sq": optimizer.exp_avg_sq[i].cpu()} for i in range(n)},
            "param_groups": [{"lr": optimizer.lr, "betas": tuple(optimizer.betas), "eps": optimizer.eps,
                              "weight_decay": optimizer.weight_decay, "params": list(range(n))}]}
        if shadow_params is None and optimizer.use_ema:
            shadow_params = optimizer.shadow_params
    elif optimizer is not None:
        ckpt["optimizer"] = optimizer.state_dict()
    else:
        ckpt["optimizer"] = {}
    if shadow_params is None:
        shadow_params = [p.detach() for p in model.parameters()]
    ckpt["ema_state_dict"] = {"shadow_params": [s.detach().cpu().clone() for s in shadow_params],
                              "optimization_step": int(iteration)}
    torch.save(ckpt, path)


def load_checkpoint(path: str, model: torch.nn.Module, use_ema: bool = True, map_location="cpu") -> dict:
    """interact.py:102-106: load_state_dict by key, then the positional EMA copy."""
    ckpt = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(ckpt["state_dict"])
    if use_ema and "ema_state_dict" in ckpt and ckpt["ema_state_dict"].get("shadow_params"):
        copy_parameters(ckpt["ema_state_dict"]["shadow_params"], model.parameters())
    if hasattr(model, "refresh_weights"):
        model.refresh_weights()
    return ckpt
