"""GuidanceLoss (reference: control/guidance.py:22-59): gradient step on the model output that
pulls the predicted trajectory towards the goal, called from the schedulers' step()."""
import importlib
from typing import Union

import torch
import torch.nn as nn


def convert(loss_config):
    it = iter(loss_config)
    return dict(zip(it, it))


class GuidanceLoss(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        module = importlib.import_module(__package__ + ".guidance_loss")
        self.loss_list = nn.ModuleList([])
        for loss_cls, loss_config in cfg.GUIDANCE.LOSS_LIST:
            self.loss_list.append(getattr(module, loss_cls)(**convert(loss_config)))
        self.guidance_step = cfg.GUIDANCE.STEP
        self.scale = cfg.GUIDANCE.CLASSIFIER_SCALE

    def compute_loss(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        total = 0
        for loss in self.loss_list:
            total = total + loss(x, target)
        return total

    def forward(self, x_guidance: torch.Tensor, action: torch.Tensor, target: torch.Tensor,
                grad_scale: Union[float, torch.Tensor] = None) -> torch.Tensor:
        for _ in range(self.guidance_step):
            with torch.enable_grad():
                if not x_guidance.requires_grad:
                    x_guidance.requires_grad_()
                loss = self.compute_loss(x_guidance, target)
                state_grad, action_grad = torch.autograd.grad([loss], [x_guidance, action])
                grad = torch.cat([state_grad[..., :-3], action_grad], dim=-1)
            if grad_scale is not None:
                grad = grad * (grad_scale.to(grad.device) if torch.is_tensor(grad_scale) else grad_scale)
            x_guidance = x_guidance.detach().clone()
            x_guidance[..., :-3] = x_guidance[..., :-3] - self.scale / 15 * grad[..., :-3]
            x_guidance[..., -3:] = x_guidance[..., -3:] - self.scale * grad[..., -3:]
        return x_guidance.clip(-1, 1)
