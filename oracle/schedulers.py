"""Oracle (test infrastructure): the reference's four scheduler step() overrides, restated.

S1 GuidanceDDIMScheduler.step   scheduler/guidance_ddim_scheduler.py:60-173
S2 GuidanceDDPMScheduler.step   scheduler/guidance_ddpm_scheduler.py:59-178
S3 InpaintingDDIMScheduler.step scheduler/inpainting_ddim_scheduler.py:10-153
S4 InpaintingDDPMScheduler.step scheduler/inpainting_ddpm_scheduler.py:10-146
S5 _threshold_sample            scheduler/guidance_ddim_scheduler.py:23-58

Coefficients stay 0-dim CPU fp32 tensors exactly as in the reference, so every scalar
is rounded to fp32 at the same points.  The guidance callback is injected
(`guidance_fn(model_output, action, target, model_std) -> model_output`) so this file does
not depend on oracle/guidance.py.
"""
from __future__ import annotations

from typing import Callable, Optional

import numpy as np
import torch

from .diffusers_base import (DDIMScheduler, DDIMSchedulerOutput, DDPMScheduler, DDPMSchedulerOutput,
                             randn_tensor)


def threshold_sample(sample: torch.Tensor, ratio: float = 0.995, sample_max_value: float = 1.0) -> torch.Tensor:
    """S5.  With sample_max_value = 1.0 (the diffusers default, never overridden) s == 1."""
    b, c, *rest = sample.shape
    flat = sample.reshape(b, c * int(np.prod(rest)))
    s = torch.quantile(flat.abs(), ratio, dim=1)
    s = torch.clamp(s, min=1, max=sample_max_value).unsqueeze(1)
    return (torch.clamp(flat, -s, s) / s).reshape(b, c, *rest)


def _x0_eps(pt: str, model_output, sample, a_t, b_t, need_eps: bool):
    if pt == "epsilon":
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        eps = model_output
    elif pt == "sample":
        x0 = model_output
        eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5 if need_eps else None
    elif pt == "v_prediction":
        x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        eps = (a_t ** 0.5) * model_output + (b_t ** 0.5) * sample
    else:
        raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
    return x0, eps


def _clip(cfg, x0):
    if cfg.thresholding:
        return threshold_sample(x0, cfg.dynamic_thresholding_ratio, cfg.sample_max_value)
    if cfg.clip_sample:
        return x0.clamp(-cfg.clip_sample_range, cfg.clip_sample_range)
    return x0


class GuidanceDDIM(DDIMScheduler):
    def __init__(self, guidance_fn: Optional[Callable] = None, **kw):
        super().__init__(**kw)
        self.guidance_fn = guidance_fn

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict=True, target=None, action=None):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        variance = self._get_variance(timestep, prev_t)
        if self.guidance_fn is not None and target is not None:
            with torch.enable_grad():
                model_std = torch.exp(0.5 * variance)
                model_output = self.guidance_fn(model_output, action, target, model_std)
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0, eps = _x0_eps(self.config.prediction_type, model_output, sample, a_t, b_t, True)
        x0 = _clip(self.config, x0)
        std = eta * variance ** 0.5
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        prev = a_prev ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
            prev = prev + std * variance_noise
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class GuidanceDDPM(DDPMScheduler):
    def __init__(self, guidance_fn: Optional[Callable] = None, **kw):
        super().__init__(**kw)
        self.guidance_fn = guidance_fn

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, target=None, action=None,
             variance_noise=None):
        """`variance_noise` is an oracle-only hook to inject the draw the reference takes from the RNG."""
        t = timestep
        prev_t = self.previous_timestep(t)
        variance = self._get_variance(t)
        if self.guidance_fn is not None and target is not None:
            with torch.enable_grad():
                model_std = torch.exp(0.5 * variance)
                model_output = self.guidance_fn(model_output, action, target, model_std)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        x0, _ = _x0_eps(self.config.prediction_type, model_output, sample, a_t, b_t, False)
        x0 = _clip(self.config, x0)
        c0 = (a_prev ** 0.5 * cur_b) / b_t
        c1 = cur_a ** 0.5 * b_prev / b_t
        prev = c0 * x0 + c1 * sample
        noise_term = 0
        if t > 0:
            z = variance_noise if variance_noise is not None else randn_tensor(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            noise_term = (self._get_variance(t) ** 0.5) * z
        prev = prev + noise_term
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class InpaintingDDIM(DDIMScheduler):
    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, target_traj=None, target_mask=None, return_dict=True):
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating "
                             "the scheduler")
        prev_t = timestep - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0, eps = _x0_eps(self.config.prediction_type, model_output, sample, a_t, b_t, True)
        x0 = _clip(self.config, x0)
        variance = self._get_variance(timestep, prev_t)
        std = eta * variance ** 0.5
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        direction = (1 - a_prev - std ** 2) ** 0.5 * eps
        # quirk kept from the reference (:108-112, :124-128): the *scalar* DDIM variance is
        # added to every element, not sigma * z.
        unknown = (a_prev ** 0.5) * x0 + direction + variance
        if target_traj is not None and target_mask is not None:
            noise = variance_noise if variance_noise is not None else randn_tensor(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            known = (a_prev ** 0.5) * target_traj + ((1.0 - a_prev) ** 0.5) * (noise if timestep > 0 else 0)
            prev = target_mask * known + (1.0 - target_mask) * unknown
        else:
            prev = unknown
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise.")
            if variance_noise is None:
                variance_noise = randn_tensor(model_output.shape, generator=generator, device=model_output.device,
                                              dtype=model_output.dtype)
            prev = prev + std * variance_noise
        if not return_dict:
            return (prev,)
        return DDIMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class InpaintingDDPM(DDPMScheduler):
    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, target_traj=None,
             target_mask=None, return_dict=True):
        t = timestep
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        x0, _ = _x0_eps(self.config.prediction_type, model_output, sample, a_t, b_t, False)
        x0 = _clip(self.config, x0)
        c0 = (a_prev ** 0.5 * cur_b) / b_t
        c1 = cur_a ** 0.5 * b_prev / b_t
        # noise is drawn unconditionally, even at t == 0 (:100-109)
        noise = variance_noise if variance_noise is not None else randn_tensor(
            model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
        std = self._get_variance(t) ** 0.5
        variance = std * noise if t > 0 else 0
        unknown = c0 * x0 + c1 * sample + variance
        if target_traj is not None and target_mask is not None:
            known = (a_prev ** 0.5) * target_traj + ((1.0 - a_prev) ** 0.5) * (noise if t > 0 else 0)
            prev = target_mask * known + (1.0 - target_mask) * unknown
        else:
            prev = unknown
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)
