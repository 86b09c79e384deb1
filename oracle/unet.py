"""Oracle (test infrastructure): TemporalMapUnet forward as plain torch-CPU functional ops.

Every function takes the reference's `state_dict` (key -> tensor) so that no nn.Module
of the product or of the reference is involved.  Citations are to /root/reference.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from .resnet import resnet34_forward

SD = Dict[str, torch.Tensor]

NO_GUIDANCE, FREE_GUIDANCE, CLASSIFIER_GUIDANCE = "NO_GUIDANCE", "FREE_GUIDANCE", "CLASSIFIER_GUIDANCE"


def sinusoidal_pos_emb(x: torch.Tensor, dim: int) -> torch.Tensor:
    """modeling/helpers.py:62-74.  `x` may be int64 (diffusion time) or float (positions)."""
    half = dim // 2
    scale = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half, device=x.device) * -scale)
    arg = x[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def conv1d_block(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """Conv1d(k, pad k//2) -> GroupNorm(8) -> Mish.  modeling/helpers.py:95-112."""
    w = sd[p + "block.0.weight"]
    y = F.conv1d(x, w, sd[p + "block.0.bias"], padding=w.shape[-1] // 2)
    y = F.group_norm(y, 8, sd[p + "block.2.weight"], sd[p + "block.2.bias"], eps=1e-5)
    return F.mish(y)


def residual_block(sd: SD, p: str, x: torch.Tensor, cond: torch.Tensor) -> torch.Tensor:
    """ResidualTemporalMapBlockConcat.forward, modeling/temporal.py:46-55 (additive time bias)."""
    tb = F.linear(F.mish(cond), sd[p + "time_mlp.1.weight"], sd[p + "time_mlp.1.bias"])
    out = conv1d_block(sd, p + "blocks.0.", x) + tb[:, :, None]
    out = conv1d_block(sd, p + "blocks.1.", out)
    if (p + "residual_conv.weight") in sd:
        res = F.conv1d(x, sd[p + "residual_conv.weight"], sd[p + "residual_conv.bias"])
    else:
        res = x
    return out + res


def downsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """modeling/helpers.py:77-83: Conv1d(C, C, 3, stride 2, pad 1)."""
    return F.conv1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def upsample(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """modeling/helpers.py:86-92: ConvTranspose1d(C, C, 4, stride 2, pad 1)."""
    return F.conv_transpose1d(x, sd[p + "conv.weight"], sd[p + "conv.bias"], stride=2, padding=1)


def time_mlp(sd: SD, t: torch.Tensor, dim: int) -> torch.Tensor:
    """modeling/temporal.py:93-98."""
    e = sinusoidal_pos_emb(t, dim).to(sd["time_mlp.1.weight"].dtype)   # no-op in fp32; lets tests run the oracle in fp64
    e = F.linear(e, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    return F.linear(F.mish(e), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])


def cond_mlp(sd: SD, cond: torch.Tensor) -> torch.Tensor:
    """modeling/temporal.py:88-92."""
    cond = cond.to(sd["cond_mlp.0.weight"].dtype)       # no-op in fp32 (the zeros of cond=None are fp32: lets tests run in fp64)
    h = F.mish(F.linear(cond, sd["cond_mlp.0.weight"], sd["cond_mlp.0.bias"]))
    return F.linear(h, sd["cond_mlp.2.weight"], sd["cond_mlp.2.bias"])


def _mha(x: torch.Tensor, in_w, in_b, out_w, out_b, heads: int, att_mask=None) -> torch.Tensor:
    B, T, E = x.shape
    dh = E // heads
    q, k, v = F.linear(x, in_w, in_b).chunk(3, dim=-1)
    q = q.reshape(B, T, heads, dh).transpose(1, 2)
    k = k.reshape(B, T, heads, dh).transpose(1, 2)
    v = v.reshape(B, T, heads, dh).transpose(1, 2)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    if att_mask is not None:          # train-mode attention dropout: multiplier tensor [B, heads, T, T]
        att = att * att_mask
    o = (att @ v).transpose(1, 2).reshape(B, T, E)
    return F.linear(o, out_w, out_b)


def traj_predict(sd: SD, p: str, action: torch.Tensor, time_embed: torch.Tensor, heads: int = 4,
                 masks=None) -> torch.Tensor:
    """TrajPredict.forward, modeling/helpers.py:22-59; eval mode unless `masks` is given.

    masks: {(layer, site): multiplier tensor (0 or 1/(1-p))} with the four dropout sites of
    nn.TransformerEncoderLayer in train mode -- 0: attention probabilities [B, heads, T, T], 1: dropout1 on the
    attention branch [B, T, E], 2: dropout on the feed-forward activation [B, T, 4E], 3: dropout2 [B, T, E].


    post-norm nn.TransformerEncoderLayer (SiLU feed-forward, no dropout in eval),
    final LayerNorm, Linear.  action [B, T, 3], time_embed [B, hidden] -> [B, T, out].
    """
    hidden = sd[p + "input_proj.weight"].shape[0]
    T = action.shape[1]
    pos = sinusoidal_pos_emb(torch.arange(T, device=action.device).float(), hidden).to(action.dtype)
    x = F.linear(action, sd[p + "input_proj.weight"], sd[p + "input_proj.bias"]) + pos[None] + time_embed[:, None, :]
    li = 0
    while (p + f"encoder_traj.layers.{li}.linear1.weight") in sd:
        q = p + f"encoder_traj.layers.{li}."
        m = (lambda site: None if masks is None else masks[(li, site)])  # noqa: E731
        sa = _mha(x, sd[q + "self_attn.in_proj_weight"], sd[q + "self_attn.in_proj_bias"],
                  sd[q + "self_attn.out_proj.weight"], sd[q + "self_attn.out_proj.bias"], heads, att_mask=m(0))
        if masks is not None:
            sa = sa * m(1)
        x = F.layer_norm(x + sa, (hidden,), sd[q + "norm1.weight"], sd[q + "norm1.bias"], 1e-5)
        act = F.silu(F.linear(x, sd[q + "linear1.weight"], sd[q + "linear1.bias"]))
        if masks is not None:
            act = act * m(2)
        ff = F.linear(act, sd[q + "linear2.weight"], sd[q + "linear2.bias"])
        if masks is not None:
            ff = ff * m(3)
        x = F.layer_norm(x + ff, (hidden,), sd[q + "norm2.weight"], sd[q + "norm2.bias"], 1e-5)
        li += 1
    x = F.layer_norm(x, (hidden,), sd[p + "encoder_traj.norm.weight"], sd[p + "encoder_traj.norm.bias"], 1e-5)
    return F.linear(x, sd[p + "output_proj.weight"], sd[p + "output_proj.bias"])


def state_from_action(sd: SD, action: torch.Tensor, time_embed: torch.Tensor, detach: bool = False) -> torch.Tensor:
    """temporal.py:238-241 (detach=True) / interact.py:158-160 (detach=False):
    state_pred on action[:, :-1], zero row prepended, cat([state, action])."""
    src = action.detach() if detach else action
    state = traj_predict(sd, "state_pred.", src[:, :-1], time_embed)
    state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
    return torch.cat([state, action], dim=-1)


def unet_forward(sd: SD, x: torch.Tensor, img: Optional[torch.Tensor], time: torch.Tensor,
                 cond: Optional[torch.Tensor] = None, *, use_cond: str = NO_GUIDANCE,
                 dim: int = 64, dim_mults: Sequence[int] = (1, 2, 4, 8),
                 return_action_and_time_only: bool = False,
                 img_feature: Optional[torch.Tensor] = None):
    """TemporalMapUnet.forward, modeling/temporal.py:197-245.

    `img_feature` short-circuits the perception pass (hoisted mode: identical in eval).
    """
    if img_feature is None:
        img_feature = resnet34_forward(sd, "perception.", img)
    x = x.transpose(1, 2)  # b h t -> b t h
    te = time_mlp(sd, time, dim)
    if use_cond == FREE_GUIDANCE:
        if cond is None:
            cond = torch.zeros((x.shape[0], 2), device=x.device)
        if te.shape[0] != cond.shape[0]:
            te = te.repeat(cond.shape[0] // te.shape[0], 1)
        if img_feature.shape[0] != cond.shape[0]:
            img_feature = img_feature.repeat(cond.shape[0] // img_feature.shape[0], 1)
        te = te + cond_mlp(sd, cond)
    ci = torch.cat([te, img_feature], dim=-1)

    n_res = len(dim_mults)
    skips = []
    for i in range(n_res):
        x = residual_block(sd, f"downs.{i}.0.", x, ci)
        x = residual_block(sd, f"downs.{i}.1.", x, ci)
        skips.append(x)
        if i < n_res - 1:
            x = downsample(sd, f"downs.{i}.3.", x)
    x = residual_block(sd, "mid_block1.", x, ci)
    x = residual_block(sd, "mid_block2.", x, ci)
    for i in range(n_res - 1):
        x = torch.cat((x, skips.pop()), dim=1)
        x = residual_block(sd, f"ups.{i}.0.", x, ci)
        x = residual_block(sd, f"ups.{i}.1.", x, ci)
        x = upsample(sd, f"ups.{i}.3.", x)  # the reference upsamples on all three (temporal.py:154)

    if use_cond == CLASSIFIER_GUIDANCE:
        a = conv1d_block(sd, "act_conv.0.", x)
        a = F.conv1d(a, sd["act_conv.1.weight"], sd["act_conv.1.bias"]).transpose(1, 2)
        if return_action_and_time_only:
            return a, te
        return state_from_action(sd, a, te, detach=True)
    y = conv1d_block(sd, "final_conv.0.", x)
    y = F.conv1d(y, sd["final_conv.1.weight"], sd["final_conv.1.bias"])
    return y.transpose(1, 2)
