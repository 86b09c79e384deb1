"""Parameter table of the denoiser, in the reference's registration order.

The checkpoint format of the reference is (a) a `state_dict` keyed by module path and
(b) a *positional* EMA list `shadow_params` that is zipped against `model.parameters()`
(interact.py:102-106, misc/load_param.py:4-8), so both the key names and the order in
which parameters are registered are part of the drop-in boundary.

This module derives that table from the architecture hyper-parameters alone
(modeling/temporal.py:59-195, modeling/resnet.py:163-296, modeling/helpers.py:22-59);
`modeling.temporal` builds its parameter holders from it and the tests compare it with a
golden dump of the reference's own `state_dict()` keys/shapes.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Sequence, Tuple

RESNET34_LAYERS = (3, 4, 6, 3)
GROUPS = 8  # GroupNorm groups in Conv1dBlock (modeling/helpers.py:100)


@dataclass(frozen=True)
class Entry:
    key: str
    shape: Tuple[int, ...]
    is_buffer: bool = False
    dtype: str = "f32"


def _conv1d(p, cin, cout, k):
    return [Entry(p + "weight", (cout, cin, k)), Entry(p + "bias", (cout,))]


def _linear(p, cin, cout):
    return [Entry(p + "weight", (cout, cin)), Entry(p + "bias", (cout,))]


def _norm(p, c):
    return [Entry(p + "weight", (c,)), Entry(p + "bias", (c,))]


def _bn(p, c):
    return _norm(p, c) + [Entry(p + "running_mean", (c,), True), Entry(p + "running_var", (c,), True),
                          Entry(p + "num_batches_tracked", (), True, "i64")]


def _conv1d_block(p, cin, cout, k=5):
    return _conv1d(p + "block.0.", cin, cout, k) + _norm(p + "block.2.", cout)


def _res_block(p, cin, cout, embed):
    e = _conv1d_block(p + "blocks.0.", cin, cout) + _conv1d_block(p + "blocks.1.", cout, cout)
    e += _linear(p + "time_mlp.1.", embed, cout)
    if cin != cout:
        e += _conv1d(p + "residual_conv.", cin, cout, 1)
    return e


def resnet34_entries(p: str, out_dim: int) -> List[Entry]:
    e = [Entry(p + "conv1.weight", (64, 3, 7, 7))] + _bn(p + "bn1.", 64)
    inplanes = 64
    for li, (planes, n) in enumerate(zip((64, 128, 256, 512), RESNET34_LAYERS), start=1):
        for bi in range(n):
            q = f"{p}layer{li}.{bi}."
            e.append(Entry(q + "conv1.weight", (planes, inplanes, 3, 3)))
            e += _bn(q + "bn1.", planes)
            e.append(Entry(q + "conv2.weight", (planes, planes, 3, 3)))
            e += _bn(q + "bn2.", planes)
            if bi == 0 and (li > 1 or inplanes != planes):
                e.append(Entry(q + "downsample.0.weight", (planes, inplanes, 1, 1)))
                e += _bn(q + "downsample.1.", planes)
            inplanes = planes
    return e + _linear(p + "fc.", 512, out_dim)


def traj_predict_entries(p: str, in_dim: int, out_dim: int, hidden: int, layers: int) -> List[Entry]:
    e = _linear(p + "input_proj.", in_dim, hidden)
    for i in range(layers):
        q = f"{p}encoder_traj.layers.{i}."
        e += [Entry(q + "self_attn.in_proj_weight", (3 * hidden, hidden)),
              Entry(q + "self_attn.in_proj_bias", (3 * hidden,))]
        e += _linear(q + "self_attn.out_proj.", hidden, hidden)
        e += _linear(q + "linear1.", hidden, 4 * hidden) + _linear(q + "linear2.", 4 * hidden, hidden)
        e += _norm(q + "norm1.", hidden) + _norm(q + "norm2.", hidden)
    return e + _norm(p + "encoder_traj.norm.", hidden) + _linear(p + "output_proj.", hidden, out_dim)


def level_channels(transition_dim: int, dim: int, dim_mults: Sequence[int]) -> List[Tuple[int, int]]:
    dims = [transition_dim] + [dim * m for m in dim_mults]
    return list(zip(dims[:-1], dims[1:]))


def unet_entries(use_cond: str = "NO_GUIDANCE", transition_dim: int = 7, dim: int = 64,
                 dim_mults: Sequence[int] = (1, 2, 4, 8)) -> List[Entry]:
    """Everything registered by TemporalMapUnet.__init__, perception included."""
    in_out = level_channels(transition_dim, dim, dim_mults)
    embed = 2 * dim
    e = resnet34_entries("perception.", dim)
    if use_cond == "FREE_GUIDANCE":
        e += _linear("cond_mlp.0.", 2, dim) + _linear("cond_mlp.2.", dim, dim)
    e += _linear("time_mlp.1.", dim, 4 * dim) + _linear("time_mlp.3.", 4 * dim, dim)
    n = len(in_out)
    for i, (ci, co) in enumerate(in_out):
        e += _res_block(f"downs.{i}.0.", ci, co, embed) + _res_block(f"downs.{i}.1.", co, co, embed)
        if i < n - 1:
            e += _conv1d(f"downs.{i}.3.conv.", co, co, 3)
    final_up = None
    for i, (ci, co) in enumerate(reversed(in_out[1:])):
        e += _res_block(f"ups.{i}.0.", 2 * co, ci, embed) + _res_block(f"ups.{i}.1.", ci, ci, embed)
        # ConvTranspose1d weight layout is [Cin, Cout, k]; the reference upsamples at all 3 levels
        e += [Entry(f"ups.{i}.3.conv.weight", (ci, ci, 4)), Entry(f"ups.{i}.3.conv.bias", (ci,))]
        final_up = ci
    mid = in_out[-1][1]
    e += _res_block("mid_block1.", mid, mid, embed) + _res_block("mid_block2.", mid, mid, embed)
    if use_cond == "CLASSIFIER_GUIDANCE":
        e += _conv1d_block("act_conv.0.", final_up, final_up) + _conv1d("act_conv.1.", final_up, 3, 1)
        e += traj_predict_entries("state_pred.", 3, transition_dim - 3, 64, 2)
    else:
        e += _conv1d_block("final_conv.0.", final_up, final_up) + _conv1d("final_conv.1.", final_up, transition_dim, 1)
    return e


def residual_block_prefixes(n_levels: int = 4) -> List[str]:
    """The 16 residual blocks in execution order (modeling/temporal.py:215-231)."""
    p = []
    for i in range(n_levels):
        p += [f"downs.{i}.0.", f"downs.{i}.1."]
    p += ["mid_block1.", "mid_block2."]
    for i in range(n_levels - 1):
        p += [f"ups.{i}.0.", f"ups.{i}.1."]
    return p
