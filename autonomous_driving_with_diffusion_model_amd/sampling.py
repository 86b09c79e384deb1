"""The callers' sampling loops, restated over the drop-in model/scheduler objects.

`generate_traj` mirrors `interact.Agent.generate_traj` (interact.py:115-168) ==
`DiffusionAgent.generate_traj` (e2e_driving/diffusion_agent.py:179-232); `evaluate_sample` mirrors
the loop of `train.evaluate` (train.py:62-90).  They exist so that parity tests and the benchmark can
drive the exact call sequence the reference's callers issue: model.forward + scheduler.step per
timestep, `trajs[:, 0, :3] = 0` before the loop and after every step, final clamp and xy scaling.

Differences from the callers, both optional and numerically identical:
  * `target` may be [B, 2] (one goal per scene) instead of one [2] goal repeated over the batch;
  * `fuse=True` folds the classifier-free combine and the `[:, 0, :3] = 0` write into the scheduler's
    step kernel instead of issuing them as separate torch ops.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from .misc.constant import GuidanceType


def _targets(target: Optional[torch.Tensor], batch: int) -> Optional[torch.Tensor]:
    if target is None:
        return None
    t = target.reshape(-1, 2)
    if t.shape[0] == 1:
        t = t.repeat(batch, 1)   # interact.py:120
    if t.shape[0] != batch:
        raise ValueError(f"target must be [2] or [{batch}, 2], got {tuple(target.shape)}")
    return t.contiguous()


def generate_traj(model, scheduler, cfg, image: torch.Tensor, target: Optional[torch.Tensor] = None,
                  init_trajs: Optional[torch.Tensor] = None, *, fuse: bool = True, scale_xy: bool = True,
                  step_noise: Optional[Callable[[int, tuple], torch.Tensor]] = None) -> torch.Tensor:
    use = GuidanceType[cfg.GUIDANCE.USE_COND]
    model.eval()
    device = image.device
    if init_trajs is None:
        init_trajs = torch.randn((image.shape[0], cfg.MODEL.HORIZON, cfg.MODEL.TRANSITION_DIM), device=device)
    trajs = init_trajs.clone().detach()
    B = trajs.shape[0]
    tgt = _targets(target, B)
    cond = None
    if tgt is not None and use == GuidanceType.FREE_GUIDANCE:
        cond = torch.cat([tgt, torch.zeros_like(tgt)], dim=0)   # interact.py:121-127
    trajs[:, 0, :3] = 0.0
    scheduler.set_timesteps(cfg.EVAL.SAMPLE_STEPS, device=device)
    is_ddpm = not getattr(scheduler, "_is_ddim", False)
    action = None
    for i, t in enumerate(scheduler.timesteps):
        extra = {}
        if is_ddpm and step_noise is not None:
            extra["variance_noise"] = step_noise(i, tuple(trajs.shape)).to(device)
        if use == GuidanceType.FREE_GUIDANCE:
            with torch.no_grad():
                out = model(torch.cat([trajs, trajs], dim=0), image, t.reshape(-1), cond=cond)
            if fuse:
                trajs = scheduler.step(out, t, trajs, cfg_scale=cfg.GUIDANCE.FREE_SCALE, zero_first=True,
                                       **extra).prev_sample
                continue
            c, u = out.chunk(2, dim=0)
            model_output = u + cfg.GUIDANCE.FREE_SCALE * (c - u)
            trajs = scheduler.step(model_output, t, trajs, **extra).prev_sample
        elif use == GuidanceType.CLASSIFIER_GUIDANCE:
            with torch.no_grad():
                action, time_embed = model(trajs, image, t.reshape(-1).repeat(B), return_action_and_time_only=True)
            guided = getattr(scheduler, "use_classifier_guidance", False) and tgt is not None
            if fuse and guided and scheduler.guidance_loss.guidance_step == 1:
                # one launch: state_pred forward + TargetGuidance + its gradient through state_pred + update + clip
                model_output = model.state_pred.guided_output(action, time_embed, tgt, scheduler.guidance_std(t),
                                                              scheduler.guidance_loss.scale)
                trajs = scheduler.step(model_output, t, trajs, zero_first=True, **extra).prev_sample
                continue
            action = action.detach().requires_grad_()
            with torch.enable_grad():
                state = model.state_pred(action[:, :-1], time_embed)
                state = torch.cat([torch.zeros_like(state[:, :1]), state], dim=1)
                model_output = torch.cat([state, action], dim=-1)
            trajs = scheduler.step(model_output, t, trajs, target=tgt, action=action, **extra).prev_sample.detach()
        else:
            with torch.no_grad():
                model_output = model(trajs, image, t.reshape(-1).repeat(B))
            if fuse:
                trajs = scheduler.step(model_output, t, trajs, zero_first=True, **extra).prev_sample
                continue
            trajs = scheduler.step(model_output, t, trajs, **extra).prev_sample
        trajs[:, 0, :3] = 0.0
    trajs = trajs.to(torch.float32).clamp(-1, 1)
    if scale_xy:
        trajs[..., :2] *= model.magic_num
    return trajs


@torch.no_grad()
def evaluate_sample(model, noise_scheduler, image: torch.Tensor, init_trajs: torch.Tensor, n_steps: int,
                    step_noise: Optional[Callable[[int, tuple], torch.Tensor]] = None) -> torch.Tensor:
    """train.evaluate's loop: stock DDPM scheduler, B copies of one image, fresh/injected noise."""
    model.eval()
    B = init_trajs.shape[0]
    trajs = init_trajs.clone()
    trajs[:, 0, :3] = 0
    noise_scheduler.set_timesteps(n_steps, device=image.device)
    for i, t in enumerate(noise_scheduler.timesteps):
        out = model(trajs, image, t.reshape(-1).repeat(B))
        kw = {}
        if step_noise is not None:
            kw["variance_noise"] = step_noise(i, tuple(trajs.shape)).to(image.device)
        trajs = noise_scheduler.step(out, t, trajs, **kw).prev_sample
        trajs[:, 0, :3] = 0
    return trajs
