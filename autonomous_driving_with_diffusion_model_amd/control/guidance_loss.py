"""TargetGuidance (reference: control/guidance_loss.py:10-22).

The reference's `if final_to_agent < target_to_agent:` is a Python branch on a tensor, so it is only
defined for B = 1 and a single target point.  This implementation applies the
same rule PER SAMPLE ("vmap of the B = 1 reference"): identical for B = 1, defined for B > 1.
`target` may be [2] (one goal for every sample, as interact.py passes it) or [B, 2].
"""
import torch
import torch.nn as nn


class TargetGuidance(nn.Module):
    def forward(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        B = x.shape[0]
        tgt = target.reshape(-1, 2)
        if tgt.shape[0] == 1:
            tgt = tgt.expand(B, 2)
        if tgt.shape[0] != B:
            raise ValueError(f"target must be [2] or [{B}, 2], got {tuple(target.shape)}")
        # softmin over a single target point is 1 (guidance_loss.py:14)
        dist = torch.sum((x[..., :2] - tgt[:, None, :]) ** 2, dim=-1)               # [B, H]
        target_to_agent = torch.norm(tgt - x[:, 0, :2], dim=-1)                      # [B]
        final_to_agent = torch.norm(x[:, -1, :2] - x[:, 0, :2], dim=-1)              # [B]
        choose = torch.where(final_to_agent < target_to_agent, torch.zeros_like(dist[:, 0], dtype=torch.long),
                             dist.argmin(dim=-1))                                    # dummy point 0 (:17)
        return dist.gather(1, choose[:, None]).sum()
