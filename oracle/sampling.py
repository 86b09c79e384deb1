"""Oracle (test infrastructure): the caller loops around the model + scheduler.

C1/C2 generate_traj     e2e_driving/diffusion_agent.py:179-232 == interact.py:115-168
D     train.evaluate    train.py:53-103 (stock DDPM sampler)
T1    training step     train.py:221-261 (loss only; autograd gives the gradients)
"""
from __future__ import annotations

from typing import Callable, Dict, Optional, Sequence

import torch
import torch.nn.functional as F

from . import unet as U
from .diffusers_base import DDPMScheduler
from .guidance import guidance_update
from .resnet import resnet34_forward
from .schedulers import GuidanceDDIM, GuidanceDDPM

SD = Dict[str, torch.Tensor]
MAGIC_NUM = 23.315  # modeling/temporal.py:195


def scheduler_kwargs(n_train: int = 100, prediction_type: str = "sample", beta_schedule: str = "squaredcos_cap_v2",
                     beta_start: float = 1e-4, beta_end: float = 0.02, thresholding: bool = True):
    """interact.py:81-90."""
    return dict(num_train_timesteps=n_train, prediction_type=prediction_type, beta_schedule=beta_schedule,
                beta_start=beta_start, beta_end=beta_end, thresholding=thresholding)


def generate_traj(sd: SD, image: torch.Tensor, init_trajs: torch.Tensor, target: Optional[torch.Tensor], *,
                  use_cond: str, n_steps: int, scheduler: str = "ddim", free_scale: float = 1.0,
                  classifier_scale: float = 0.1, guidance_steps: int = 1, dim: int = 64,
                  dim_mults: Sequence[int] = (1, 2, 4, 8), hoist_perception: bool = False,
                  step_noise: Optional[Callable[[int, tuple], torch.Tensor]] = None,
                  sched_kw: Optional[dict] = None, scale_xy: bool = True,
                  img_feature: Optional[torch.Tensor] = None) -> torch.Tensor:
    """The agent's sampling loop.  For CLASSIFIER guidance with B > 1 every sample is an
    independent B = 1 problem (the reference only defines B = 1).  `img_feature` = the perception
    output for `image` computed by the caller (oracle.resnet), to share one pass between tests."""
    kw = sched_kw or scheduler_kwargs()
    guided = use_cond == U.CLASSIFIER_GUIDANCE and target is not None

    def guidance_fn(model_output, action, tgt, model_std):
        return guidance_update(model_output, action, tgt, model_std, classifier_scale, guidance_steps)

    cls = GuidanceDDIM if scheduler == "ddim" else GuidanceDDPM
    sch = cls(guidance_fn=guidance_fn if guided else None, **kw)

    if use_cond == U.CLASSIFIER_GUIDANCE and init_trajs.shape[0] > 1:
        outs = []
        for b in range(init_trajs.shape[0]):
            img_b = image[b:b + 1] if image.shape[0] > 1 else image
            tgt_b = None if target is None else (target[b] if target.dim() > 1 else target)
            sn = None if step_noise is None else (lambda i, shape, b=b: step_noise(i, (init_trajs.shape[0],) + tuple(shape[1:]))[b:b + 1])
            outs.append(generate_traj(sd, img_b, init_trajs[b:b + 1], tgt_b, use_cond=use_cond, n_steps=n_steps,
                                      scheduler=scheduler, free_scale=free_scale, classifier_scale=classifier_scale,
                                      guidance_steps=guidance_steps, dim=dim, dim_mults=dim_mults,
                                      hoist_perception=hoist_perception, step_noise=sn, sched_kw=sched_kw,
                                      scale_xy=scale_xy,
                                      img_feature=None if img_feature is None else img_feature[b:b + 1]))
        return torch.cat(outs, dim=0)

    trajs = init_trajs.clone().detach()
    cond = None
    if target is not None and use_cond == U.FREE_GUIDANCE:
        tg = target if target.dim() > 1 else target.repeat(trajs.size(0), 1)
        cond = torch.cat([tg, torch.zeros_like(tg)], dim=0)
    feat = img_feature
    if feat is None and hoist_perception:
        feat = resnet34_forward(sd, "perception.", image)
    trajs[:, 0, :3] = 0.0
    sch.set_timesteps(n_steps)
    action = None
    for i, t in enumerate(sch.timesteps):
        if use_cond == U.FREE_GUIDANCE:
            inp = torch.cat([trajs, trajs], dim=0)
            with torch.no_grad():
                out = U.unet_forward(sd, inp, image, t.reshape(-1), cond, use_cond=use_cond, dim=dim,
                                     dim_mults=dim_mults, img_feature=feat)
                c, u = out.chunk(2, dim=0)
            model_output = u + free_scale * (c - u)
        elif use_cond == U.CLASSIFIER_GUIDANCE:
            with torch.no_grad():
                action, te = U.unet_forward(sd, trajs, image, t.reshape(-1), use_cond=use_cond, dim=dim,
                                            dim_mults=dim_mults, return_action_and_time_only=True, img_feature=feat)
            action = action.detach().requires_grad_()
            with torch.enable_grad():
                model_output = U.state_from_action(sd, action, te)
        else:
            with torch.no_grad():
                model_output = U.unet_forward(sd, trajs, image, t.reshape(-1), use_cond=use_cond, dim=dim,
                                              dim_mults=dim_mults, img_feature=feat)
        extra = {}
        if scheduler == "ddpm" and step_noise is not None:
            extra["variance_noise"] = step_noise(i, tuple(trajs.shape))
        trajs = sch.step(model_output, t, trajs, target=target if guided else None, action=action,
                         **extra).prev_sample.detach()
        trajs[:, 0, :3] = 0.0
    trajs = trajs.to(torch.float32).clamp(-1, 1)
    if scale_xy:
        trajs[..., :2] *= MAGIC_NUM
    return trajs


def evaluate_loop(sd: SD, image: torch.Tensor, init_trajs: torch.Tensor, *, n_steps: int, n_train: int = 100,
                  step_noise: Callable[[int, tuple], torch.Tensor], use_cond: str = U.NO_GUIDANCE, dim: int = 64,
                  dim_mults: Sequence[int] = (1, 2, 4, 8), hoist_perception: bool = False,
                  prediction_type: str = "sample", beta_schedule: str = "squaredcos_cap_v2") -> torch.Tensor:
    """train.evaluate: stock diffusers DDPMScheduler (clip_sample=True), noise injected per step.
    Returns the full [B, H, D] trajectories after the loop (the caller keeps [..., :2].clamp)."""
    sch = DDPMScheduler(num_train_timesteps=n_train, prediction_type=prediction_type, beta_schedule=beta_schedule,
                        beta_start=1e-4, beta_end=0.02)
    B = init_trajs.shape[0]
    trajs = init_trajs.clone()
    trajs[:, 0, :3] = 0
    feat = resnet34_forward(sd, "perception.", image) if hoist_perception else None
    sch.set_timesteps(n_steps)
    for i, t in enumerate(sch.timesteps):
        out = U.unet_forward(sd, trajs, image, t.reshape(-1).repeat(B), use_cond=use_cond, dim=dim,
                             dim_mults=dim_mults, img_feature=feat)
        z = step_noise(i, tuple(trajs.shape))
        trajs = sch.step(out, t, trajs, variance_noise=z).prev_sample
        trajs[:, 0, :3] = 0
    return trajs


def training_loss(sd: SD, imgs, trajs, target, t, noise, *, use_cond: str, drop_cond: bool = False,
                  prediction_type: str = "sample", dim: int = 64, dim_mults: Sequence[int] = (1, 2, 4, 8),
                  n_train: int = 100, bn_training: bool = True) -> torch.Tensor:
    """T1 forward half: add_noise -> zero [...,0,:3] -> model -> MSE.  `sd` tensors may require grad."""
    from . import resnet as R
    sch = DDPMScheduler(num_train_timesteps=n_train, prediction_type=prediction_type,
                        beta_schedule="squaredcos_cap_v2", beta_start=1e-4, beta_end=0.02)
    noisy = sch.add_noise(trajs, noise, t)
    noisy[..., 0, :3] = 0
    cond = None
    if use_cond == U.FREE_GUIDANCE and not drop_cond:
        cond = target
    feat = R.resnet34_forward(sd, "perception.", imgs, training=bn_training)
    pred = U.unet_forward(sd, noisy, None, t, cond, use_cond=use_cond, dim=dim, dim_mults=dim_mults,
                          img_feature=feat)
    if prediction_type == "epsilon":
        return F.mse_loss(pred.float(), noise.float())
    if prediction_type == "sample":
        return F.mse_loss(pred.float(), trajs.float())
    raise ValueError("Not supported prediction type.")
