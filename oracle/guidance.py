"""Oracle (test infrastructure): classifier-guidance gradient step.

G1 GuidanceLoss.forward   control/guidance.py:35-59
G2 TargetGuidance.forward control/guidance_loss.py:10-22

The reference only works for B = 1 and one target point (the Python `if`
on a tensor).  The batched rule used by the build is "vmap of the B = 1 reference":
`guidance_update_batched` applies the B = 1 function to every sample independently.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def target_guidance_loss(x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """G2, x [1, H, D], target [2] (or [1, 2])."""
    while target.dim() < x.dim():
        target = target.unsqueeze(0)
    w = F.softmin(torch.norm(target, dim=-1), dim=-1)
    dist = torch.sum((x[..., :2].unsqueeze(1) - target.unsqueeze(2)) ** 2, dim=-1)
    target_to_agent = torch.norm(target - x[:, 0, :2], dim=-1)
    final_to_agent = torch.norm(x[:, -1, :2] - x[:, 0, :2], dim=-1)
    if bool(final_to_agent < target_to_agent):
        choose = 0
    else:
        choose = dist.argmin(dim=-1)
    return (dist[:, :, choose] * w).mean(dim=-1).sum()


def guidance_update(x_guidance: torch.Tensor, action: torch.Tensor, target: torch.Tensor,
                    grad_scale: Optional[torch.Tensor], scale: float, steps: int = 1) -> torch.Tensor:
    """G1 for B = 1.  `x_guidance` = cat(state_pred(action), action) must be in `action`'s graph."""
    for _ in range(steps):
        with torch.enable_grad():
            if not x_guidance.requires_grad:
                x_guidance.requires_grad_()
            loss = target_guidance_loss(x_guidance, target)
            g_x, g_a = torch.autograd.grad([loss], [x_guidance, action])
            grad = torch.cat([g_x[..., :-3], g_a], dim=-1)
        if grad_scale is not None:
            grad = grad * grad_scale
        x_guidance = x_guidance.detach().clone()
        x_guidance[..., :-3] = x_guidance[..., :-3] - scale / 15 * grad[..., :-3]
        x_guidance[..., -3:] = x_guidance[..., -3:] - scale * grad[..., -3:]
    return x_guidance.clip(-1, 1)
