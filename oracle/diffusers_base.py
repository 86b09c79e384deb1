"""Oracle (test infrastructure): restatement of the `diffusers==0.28.0` scheduler bases.

The reference's four schedulers subclass `diffusers.DDPMScheduler` / `DDIMScheduler`
(scheduler/guidance_ddim_scheduler.py:12, guidance_ddpm_scheduler.py:12,
inpainting_ddim_scheduler.py:9, inpainting_ddpm_scheduler.py:9); diffusers is pinned to
0.28.0 in requirements.txt:2 and is neither vendored in /root/reference nor installed here.
This file restates, from the published 0.28.0 API, exactly the members the reference uses
(call sites listed in SURVEY.md §8c).  **PARITY UNPINNED**: the reference has no tests or
golden vectors for these; the SURVEY §8(c) known answers are asserted in tests.

`tests/golden/make_golden.py` also exposes these classes under the module name `diffusers`
so that the reference's own `scheduler/*.py` step() bodies can be executed unmodified on top
of them when generating golden vectors.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch


def randn_tensor(shape, generator=None, device=None, dtype=None, layout=None):
    """diffusers.utils.torch_utils.randn_tensor (single-generator case)."""
    return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)


@dataclass
class DDPMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


@dataclass
class DDIMSchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


def betas_for_alpha_bar(n: int, max_beta: float = 0.999) -> torch.Tensor:
    """squaredcos_cap_v2: python-float64 cosine ratio, then float32."""
    def alpha_bar(u):
        return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2

    betas = []
    for i in range(n):
        t1, t2 = i / n, (i + 1) / n
        betas.append(min(1 - alpha_bar(t2) / alpha_bar(t1), max_beta))
    return torch.tensor(betas, dtype=torch.float32)


def make_betas(schedule: str, n: int, beta_start: float, beta_end: float) -> torch.Tensor:
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    if schedule == "squaredcos_cap_v2":
        return betas_for_alpha_bar(n)
    raise NotImplementedError(f"{schedule} is not implemented")


class _SchedulerBase:
    def _init_common(self, num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type,
                     clip_sample, clip_sample_range, thresholding, dynamic_thresholding_ratio,
                     sample_max_value, timestep_spacing, steps_offset, **extra):
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, prediction_type=prediction_type, clip_sample=clip_sample,
            clip_sample_range=clip_sample_range, thresholding=thresholding,
            dynamic_thresholding_ratio=dynamic_thresholding_ratio, sample_max_value=sample_max_value,
            timestep_spacing=timestep_spacing, steps_offset=steps_offset, **extra)
        self.betas = make_betas(beta_schedule, num_train_timesteps, beta_start, beta_end)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than "
                f"`self.config.train_timesteps`: {n_train}")
        self.num_inference_steps = num_inference_steps
        if self.config.timestep_spacing != "leading":
            raise NotImplementedError("only the default 'leading' spacing is used by the reference")
        ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    def add_noise(self, original: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor) -> torch.Tensor:
        ac = self.alphas_cumprod.to(device=original.device, dtype=original.dtype)
        timesteps = timesteps.to(original.device)
        sa = (ac[timesteps] ** 0.5).flatten()
        while sa.dim() < original.dim():
            sa = sa.unsqueeze(-1)
        sb = ((1 - ac[timesteps]) ** 0.5).flatten()
        while sb.dim() < original.dim():
            sb = sb.unsqueeze(-1)
        return sa * original + sb * noise

    def scale_model_input(self, sample, timestep=None):
        return sample

    def __len__(self):
        return self.config.num_train_timesteps


class DDPMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                 sample_max_value=1.0, timestep_spacing="leading", steps_offset=0,
                 rescale_betas_zero_snr=False):
        self._init_common(num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type, clip_sample,
                          clip_sample_range, thresholding, dynamic_thresholding_ratio, sample_max_value,
                          timestep_spacing, steps_offset, variance_type=variance_type)
        self.one = torch.tensor(1.0)
        self.custom_timesteps = False
        self.variance_type = variance_type

    def previous_timestep(self, timestep):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_prev
        variance = (1 - a_prev) / (1 - a_t) * cur_beta
        variance = torch.clamp(variance, min=1e-20)
        if (variance_type or self.config.variance_type) != "fixed_small":
            raise NotImplementedError("the reference only uses variance_type='fixed_small'")
        return variance

    def _threshold_sample(self, sample: torch.Tensor) -> torch.Tensor:
        b, c, *rest = sample.shape
        flat = sample.reshape(b, c * int(np.prod(rest)))
        s = torch.quantile(flat.abs(), self.config.dynamic_thresholding_ratio, dim=1)
        s = torch.clamp(s, min=1, max=self.config.sample_max_value).unsqueeze(1)
        return (torch.clamp(flat, -s, s) / s).reshape(b, c, *rest)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True, variance_noise=None):
        """Stock DDPM step (used by train.py:87): fixed_small variance, leading spacing.
        `variance_noise` is an oracle-only hook replacing the RNG draw (parity runs inject noise)."""
        t = timestep
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        pt = self.config.prediction_type
        if pt == "epsilon":
            x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        elif pt == "sample":
            x0 = model_output
        elif pt == "v_prediction":
            x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output
        else:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample` or `v_prediction`")
        if self.config.thresholding:
            x0 = self._threshold_sample(x0)
        elif self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        c0 = (a_prev ** 0.5 * cur_b) / b_t
        c1 = cur_a ** 0.5 * b_prev / b_t
        prev = c0 * x0 + c1 * sample
        variance = 0
        if t > 0:
            z = variance_noise if variance_noise is not None else randn_tensor(
                model_output.shape, generator=generator, device=model_output.device, dtype=model_output.dtype)
            variance = (self._get_variance(t) ** 0.5) * z
        prev = prev + variance
        if not return_dict:
            return (prev,)
        return DDPMSchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDIMScheduler(_SchedulerBase):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        self._init_common(num_train_timesteps, beta_start, beta_end, beta_schedule, prediction_type, clip_sample,
                          clip_sample_range, thresholding, dynamic_thresholding_ratio, sample_max_value,
                          timestep_spacing, steps_offset, set_alpha_to_one=set_alpha_to_one)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def _get_variance(self, timestep, prev_timestep):
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        return (b_prev / b_t) * (1 - a_t / a_prev)

    _threshold_sample = DDPMScheduler._threshold_sample


# -- training helpers inherited from diffusers (train.py:146-153, 170-174) -------------------------

def ema_decay(optimization_step: int, update_after_step: int = 0, inv_gamma: float = 1.0, power: float = 2 / 3,
              min_decay: float = 0.0, max_decay: float = 0.9999, use_ema_warmup: bool = True) -> float:
    """diffusers.training_utils.EMAModel.get_decay."""
    step = max(0, optimization_step - update_after_step - 1)
    if step <= 0:
        return 0.0
    if use_ema_warmup:
        cur = 1 - (1 + step / inv_gamma) ** -power
    else:
        cur = (1 + step) / (10 + step)
    return max(min(cur, max_decay), min_decay)


def constant_with_warmup_lr(step: int, warmup: int) -> float:
    """diffusers.optimization.get_constant_schedule_with_warmup lambda."""
    if step < warmup:
        return float(step) / float(max(1.0, warmup))
    return 1.0
