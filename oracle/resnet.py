"""Oracle (test infrastructure): ResNet-34 perception forward, plain torch-CPU functional ops.

Restates modeling/resnet.py:56-102 (BasicBlock), :163-296 (ResNet) with layers [3,4,6,3]
(:325-333) and the replaced `fc = Linear(512, dim)` (modeling/temporal.py:83-84).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
LAYERS = (3, 4, 6, 3)


def batch_norm(sd: SD, p: str, x: torch.Tensor, training: bool = False) -> torch.Tensor:
    """nn.BatchNorm2d: eval = running stats; train = batch stats (running stats are not updated here)."""
    if training:
        return F.batch_norm(x, None, None, sd[p + "weight"], sd[p + "bias"], True, 0.1, 1e-5)
    return F.batch_norm(x, sd[p + "running_mean"], sd[p + "running_var"], sd[p + "weight"], sd[p + "bias"],
                        False, 0.1, 1e-5)


def basic_block(sd: SD, p: str, x: torch.Tensor, stride: int, training: bool = False) -> torch.Tensor:
    """modeling/resnet.py:87-102."""
    out = F.conv2d(x, sd[p + "conv1.weight"], None, stride=stride, padding=1)
    out = F.relu(batch_norm(sd, p + "bn1.", out, training))
    out = F.conv2d(out, sd[p + "conv2.weight"], None, stride=1, padding=1)
    out = batch_norm(sd, p + "bn2.", out, training)
    if (p + "downsample.0.weight") in sd:
        idt = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=stride)
        idt = batch_norm(sd, p + "downsample.1.", idt, training)
    else:
        idt = x
    return F.relu(out + idt)


def resnet34_features(sd: SD, p: str, img: torch.Tensor, training: bool = False) -> torch.Tensor:
    """Stem + 4 stages; returns the layer4 map [B, 512, h, w]."""
    x = F.conv2d(img, sd[p + "conv1.weight"], None, stride=2, padding=3)
    x = F.relu(batch_norm(sd, p + "bn1.", x, training))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    for li, n in enumerate(LAYERS, start=1):
        for bi in range(n):
            stride = 2 if (li > 1 and bi == 0) else 1
            x = basic_block(sd, f"{p}layer{li}.{bi}.", x, stride, training)
    return x


def resnet34_forward(sd: SD, p: str, img: torch.Tensor, training: bool = False) -> torch.Tensor:
    """modeling/resnet.py:277-293: features -> global avgpool -> fc -> [B, dim]."""
    x = resnet34_features(sd, p, img, training)
    x = torch.flatten(F.adaptive_avg_pool2d(x, (1, 1)), 1)
    return F.linear(x, sd[p + "fc.weight"], sd[p + "fc.bias"])
