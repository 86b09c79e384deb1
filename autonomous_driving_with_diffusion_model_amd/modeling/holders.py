"""Parameter holders: an nn.Module tree whose state_dict keys, parameter registration order and
default initialisation equal the reference's, built from the table in modeling/spec.py.

The holders carry no compute: every forward goes through libadx.so.  They exist so that
`.to()`, `.parameters()` (== EMA `shadow_params` order, misc/load_param.py:4-8), `.state_dict()`,
`.load_state_dict()`, DDP wrapping and optimizers see exactly what they see on the reference.
"""
from __future__ import annotations

import math
from typing import Iterable

import torch
import torch.nn as nn

from .spec import Entry


class Node(nn.Module):
    """Plain container; children and parameters are attached by `populate`."""


def _init_tensor(e: Entry, all_keys) -> torch.Tensor:
    """Default initialisation of the reference for the layer that owns `e` (PyTorch defaults:
    kaiming_uniform(a=sqrt 5) weights with U(+-1/sqrt(fan_in)) biases; ResNet: kaiming_normal
    fan_out / BN 1,0 (modeling/resnet.py:212-217); TrajPredict: xavier_uniform on every tensor of
    dim > 1 (modeling/helpers.py:47-50))."""
    shape, key = e.shape, e.key
    leaf = key.rsplit(".", 1)[-1]
    if e.dtype == "i64":
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_mean":
        return torch.zeros(shape)
    if leaf == "running_var":
        return torch.ones(shape)
    is_norm = (key[: -len(leaf)] + "running_mean") in all_keys or ".block.2." in key or ".norm" in key
    if is_norm:
        return torch.ones(shape) if leaf == "weight" else torch.zeros(shape)
    if key.startswith("state_pred."):
        if len(shape) > 1:
            t = torch.empty(shape)
            nn.init.xavier_uniform_(t)
            return t
        if leaf == "in_proj_bias" or key.endswith("out_proj.bias"):
            return torch.zeros(shape)
    if key.startswith("perception.") and len(shape) == 4:
        t = torch.empty(shape)
        nn.init.kaiming_normal_(t, mode="fan_out", nonlinearity="relu")
        return t
    if len(shape) > 1:
        t = torch.empty(shape)
        nn.init.kaiming_uniform_(t, a=math.sqrt(5))
        return t
    # bias: U(+-1/sqrt(fan_in)) of the sibling weight
    wkey = key[: -len(leaf)] + "weight"
    wshape = all_keys.get(wkey)
    if wshape is None or len(wshape) < 2:
        return torch.zeros(shape)
    if ".3.conv." in key and key.startswith("ups."):
        fan_in = wshape[1] * wshape[2]  # ConvTranspose1d: weight.size(1) * k
    else:
        fan_in = int(torch.tensor(wshape[1:]).prod())
    bound = 1.0 / math.sqrt(fan_in) if fan_in > 0 else 0.0
    return torch.empty(shape).uniform_(-bound, bound)


def populate(root: nn.Module, entries: Iterable[Entry], prefix: str = "", init_prefix: str = "") -> None:
    """Attach every entry below `root`, creating intermediate containers in first-seen order.
    `init_prefix` is prepended to the keys only to select the initialisation rule."""
    entries = list(entries)
    all_keys = {init_prefix + e.key: e.shape for e in entries}
    for e in entries:
        assert e.key.startswith(prefix)
        parts = e.key[len(prefix):].split(".")
        mod = root
        for p in parts[:-1]:
            if p not in mod._modules:
                mod.add_module(p, Node())
            mod = mod._modules[p]
        value = _init_tensor(Entry(init_prefix + e.key, e.shape, e.is_buffer, e.dtype), all_keys)
        if e.is_buffer:
            mod.register_buffer(parts[-1], value)
        else:
            mod.register_parameter(parts[-1], nn.Parameter(value))
