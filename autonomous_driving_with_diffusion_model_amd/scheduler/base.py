"""Host side of the DDPM/DDIM schedulers.

The reference's scheduler classes subclass `diffusers==0.28.0` DDPMScheduler / DDIMScheduler
(scheduler/guidance_ddim_scheduler.py:12 etc.).  diffusers is not a dependency here: this module
provides the members the reference and its callers use (ctor kwargs, `.config`, `.betas`,
`.alphas_cumprod`, `.set_timesteps`, `.timesteps`, `.add_noise`, `._get_variance`,
`.previous_timestep`, `.init_noise_sigma`, `.scale_model_input`) with the same arithmetic:
fp32 tables built with torch on the host, scalar coefficients evaluated as 0-dim fp32 CPU tensors in
the reference's operation order, then handed BY VALUE to the fused HIP step kernel
(csrc/sched.hip) — the per-step device->host syncs of the reference disappear.

`timesteps` is a sequence of 0-dim int64 *device* tensors, each carrying its Python value, so that
`for t in scheduler.timesteps: model(x, img, t.reshape(-1)); scheduler.step(out, t, x)` never syncs.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from .. import _lib as L

PRED = {"epsilon": 0, "sample": 1, "v_prediction": 2}


@dataclass
class SchedulerOutput:
    prev_sample: torch.Tensor
    pred_original_sample: Optional[torch.Tensor] = None


DDPMSchedulerOutput = SchedulerOutput
DDIMSchedulerOutput = SchedulerOutput


class TimestepSequence:
    """What `scheduler.timesteps` returns: iterable/indexable like a 1-D int64 tensor."""

    def __init__(self, values: List[int], device=None):
        self._host = [int(v) for v in values]
        self._tensor = torch.tensor(self._host, dtype=torch.int64, device=device)
        self._items = list(self._tensor.unbind(0)) if len(self._host) else []
        for v, it in zip(self._host, self._items):
            it._adx_int = v

    def __iter__(self):
        return iter(self._items)

    def __len__(self):
        return len(self._host)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self._tensor[i]
        return self._items[i]

    def tolist(self):
        return list(self._host)

    @property
    def tensor(self):
        return self._tensor

    @property
    def device(self):
        return self._tensor.device

    @property
    def dtype(self):
        return self._tensor.dtype

    @property
    def shape(self):
        return self._tensor.shape

    def to(self, *a, **k):
        return self._tensor.to(*a, **k)

    def cpu(self):
        return self._tensor.cpu()

    def numpy(self):
        return np.asarray(self._host, dtype=np.int64)

    def __repr__(self):
        return f"TimestepSequence({self._host}, device={self._tensor.device})"


def timestep_to_int(t) -> int:
    v = getattr(t, "_adx_int", None)
    if v is not None:
        return v
    return int(t)  # foreign tensor: one device sync, like the reference


def betas_for_alpha_bar(n: int, max_beta: float = 0.999) -> torch.Tensor:
    def alpha_bar(u):
        return math.cos((u + 0.008) / 1.008 * math.pi / 2) ** 2

    return torch.tensor([min(1 - alpha_bar((i + 1) / n) / alpha_bar(i / n), max_beta) for i in range(n)],
                        dtype=torch.float32)


class SchedulerBase:
    _is_ddim = False

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, prediction_type="epsilon", thresholding=False,
                 dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False, **extra):
        if trained_betas is not None:
            betas = torch.tensor(trained_betas, dtype=torch.float32)
        elif beta_schedule == "linear":
            betas = torch.linspace(beta_start, beta_end, num_train_timesteps, dtype=torch.float32)
        elif beta_schedule == "scaled_linear":
            betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps, dtype=torch.float32) ** 2
        elif beta_schedule == "squaredcos_cap_v2":
            betas = betas_for_alpha_bar(num_train_timesteps)
        else:
            raise NotImplementedError(f"{beta_schedule} is not implemented for {self.__class__}")
        if rescale_betas_zero_snr:
            raise NotImplementedError("rescale_betas_zero_snr is not used by the reference")
        if timestep_spacing != "leading":
            raise NotImplementedError("only timestep_spacing='leading' (the diffusers default) is used by the reference")
        if prediction_type not in PRED:
            # diffusers defers this check to step(); the reference's step() raises the same ValueError
            self._bad_prediction_type = prediction_type
        self.config = SimpleNamespace(
            num_train_timesteps=num_train_timesteps, beta_start=beta_start, beta_end=beta_end,
            beta_schedule=beta_schedule, trained_betas=trained_betas, clip_sample=clip_sample,
            prediction_type=prediction_type, thresholding=thresholding,
            dynamic_thresholding_ratio=dynamic_thresholding_ratio, clip_sample_range=clip_sample_range,
            sample_max_value=sample_max_value, timestep_spacing=timestep_spacing, steps_offset=steps_offset,
            rescale_betas_zero_snr=rescale_betas_zero_snr, **extra)
        if thresholding and sample_max_value != 1.0:
            raise NotImplementedError("dynamic thresholding with sample_max_value != 1 needs the per-row quantile; "
                                      "the reference always runs with the default 1.0 (== clamp to [-1, 1])")
        self.betas = betas
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.custom_timesteps = False
        self._dev_tables = {}
        self.timesteps = TimestepSequence(list(range(num_train_timesteps))[::-1])

    # -- diffusers API ---------------------------------------------------------------------------
    def __len__(self):
        return self.config.num_train_timesteps

    def scale_model_input(self, sample, timestep=None):
        return sample

    def set_timesteps(self, num_inference_steps: int, device=None):
        n_train = self.config.num_train_timesteps
        if num_inference_steps > n_train:
            raise ValueError(
                f"`num_inference_steps`: {num_inference_steps} cannot be larger than `self.config.train_timesteps`:"
                f" {n_train} as the unet model trained with this scheduler can only handle"
                f" maximal {n_train} timesteps.")
        self.num_inference_steps = num_inference_steps
        ratio = n_train // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = TimestepSequence(ts.tolist(), device=device)

    def previous_timestep(self, timestep):
        n = self.num_inference_steps if self.num_inference_steps else self.config.num_train_timesteps
        return timestep - self.config.num_train_timesteps // n

    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                  zero_first: bool = False) -> torch.Tensor:
        """sqrt(abar_t) * x + sqrt(1 - abar_t) * noise with a per-sample gather (train.py:234).
        `zero_first=True` additionally applies train.py:235 (`noisy[..., 0, :3] = 0`) in the same kernel."""
        x = L.require_gpu_f32(original_samples, "original_samples")
        n = L.require_gpu_f32(noise, "noise")
        t = L.require_gpu_f32(timesteps.to(x.device).reshape(-1), "timesteps", torch.int64)
        if x.dim() != 3 or n.shape != x.shape or t.shape[0] != x.shape[0]:
            raise ValueError(f"add_noise expects x/noise [B,H,D] and t [B]; got {tuple(x.shape)}, {tuple(n.shape)}, "
                             f"{tuple(t.shape)}")
        sa, sb = self._tables(x.device)
        out = torch.empty_like(x)
        B, H, D = x.shape
        L.check(L.lib().adx_add_noise(x.data_ptr(), n.data_ptr(), t.data_ptr(), sa.data_ptr(), sb.data_ptr(),
                                      self.config.num_train_timesteps, out.data_ptr(), B, H, D, int(zero_first),
                                      L.stream_ptr(x.device)), "adx_add_noise")
        return out

    # -- helpers -----------------------------------------------------------------------------------
    def _tables(self, device):
        key = str(device)
        if key not in self._dev_tables:
            ac = self.alphas_cumprod
            self._dev_tables[key] = ((ac ** 0.5).to(device), ((1 - ac) ** 0.5).to(device))
        return self._dev_tables[key]

    def _check_step_inputs(self, model_output, sample, cfg_combine=False):
        mo = L.require_gpu_f32(model_output, "model_output")
        x = L.require_gpu_f32(sample, "sample")
        if x.dim() != 3:
            raise ValueError(f"sample must be [B, H, D], got {tuple(x.shape)}")
        want = (2 * x.shape[0],) + tuple(x.shape[1:]) if cfg_combine else tuple(x.shape)
        if tuple(mo.shape) != want:
            raise ValueError(f"model_output shape {tuple(mo.shape)} != expected {want}")
        return mo, x

    def _base_coef(self, a_t: torch.Tensor) -> L.StepCoef:
        c = L.StepCoef()
        pt = self.config.prediction_type
        if pt not in PRED:
            raise ValueError(f"prediction_type given as {pt} must be one of `epsilon`, `sample`, or `v_prediction`")
        c.prediction_type = PRED[pt]
        if self.config.thresholding:       # == clamp(-1, 1): sample_max_value is 1
            c.clip, c.clip_range = 1, 1.0
        elif self.config.clip_sample:
            c.clip, c.clip_range = 1, float(self.config.clip_sample_range)
        else:
            c.clip, c.clip_range = 0, 0.0
        b_t = 1 - a_t
        c.sqrt_alpha_t = float(a_t ** 0.5)
        c.sqrt_beta_t = float(b_t ** 0.5)
        return c

    def _launch(self, ddpm: bool, c: L.StepCoef, mo, x, noise, target, mask, want_x0=True):
        B, H, D = x.shape
        prev = torch.empty_like(x)
        x0 = torch.empty_like(x) if want_x0 else None
        fn = L.lib().adx_ddpm_step if ddpm else L.lib().adx_ddim_step
        import ctypes as C
        L.check(fn(C.byref(c), mo.data_ptr(), x.data_ptr(), L.ptr(noise), L.ptr(target), L.ptr(mask), prev.data_ptr(),
                   L.ptr(x0), B, H, D, L.stream_ptr(x.device)), "scheduler step")
        return prev, x0

    @staticmethod
    def _noise(shape, generator, device, dtype, variance_noise=None):
        if variance_noise is not None:
            return L.require_gpu_f32(variance_noise, "variance_noise")
        return torch.randn(tuple(shape), generator=generator, device=device, dtype=dtype)


class DDPMScheduler(SchedulerBase):
    """Stock `diffusers.DDPMScheduler` surface used by train.py:137-144,80-87,234."""

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, rescale_betas_zero_snr=False):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                         prediction_type, thresholding, dynamic_thresholding_ratio, clip_sample_range,
                         sample_max_value, timestep_spacing, steps_offset, rescale_betas_zero_snr,
                         variance_type=variance_type)
        if variance_type != "fixed_small":
            raise NotImplementedError("only variance_type='fixed_small' (the default the reference uses)")
        self.variance_type = variance_type

    def _get_variance(self, t, predicted_variance=None, variance_type=None):
        t = timestep_to_int(t)
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        cur_beta = 1 - a_t / a_prev
        variance = (1 - a_prev) / (1 - a_t) * cur_beta
        return torch.clamp(variance, min=1e-20)

    def _ddpm_coef(self, t: int) -> L.StepCoef:
        prev_t = self.previous_timestep(t)
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.one
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1 - cur_a
        c = self._base_coef(a_t)
        c.c_x0 = float((a_prev ** 0.5 * cur_b) / b_t)
        c.c_x = float(cur_a ** 0.5 * b_prev / b_t)
        c.c_noise = float(self._get_variance(t) ** 0.5)
        c.add_noise = int(t > 0)
        c.c_known = float(a_prev ** 0.5)
        c.c_known_noise = float((1.0 - a_prev) ** 0.5)
        c.known_noise = int(t > 0)
        return c

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, variance_noise=None):
        t = timestep_to_int(timestep)
        mo, x = self._check_step_inputs(model_output, sample)
        c = self._ddpm_coef(t)
        z = self._noise(x.shape, generator, x.device, x.dtype, variance_noise) if t > 0 else None
        prev, x0 = self._launch(True, c, mo, x, z, None, None)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class DDIMScheduler(SchedulerBase):
    _is_ddim = True

    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False):
        super().__init__(num_train_timesteps, beta_start, beta_end, beta_schedule, trained_betas, clip_sample,
                         prediction_type, thresholding, dynamic_thresholding_ratio, clip_sample_range,
                         sample_max_value, timestep_spacing, steps_offset, rescale_betas_zero_snr,
                         set_alpha_to_one=set_alpha_to_one)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]

    def _get_variance(self, timestep, prev_timestep):
        timestep, prev_timestep = timestep_to_int(timestep), timestep_to_int(prev_timestep)
        a_t = self.alphas_cumprod[timestep]
        a_prev = self.alphas_cumprod[prev_timestep] if prev_timestep >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        b_prev = 1 - a_prev
        return (b_prev / b_t) * (1 - a_t / a_prev)

    def _ddim_coef(self, t: int, eta: float, use_clipped_model_output: bool) -> L.StepCoef:
        if self.num_inference_steps is None:
            raise ValueError("Number of inference steps is 'None', you need to run 'set_timesteps' after creating the "
                             "scheduler")
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_prev = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        variance = self._get_variance(t, prev_t)
        std = eta * variance ** 0.5
        c = self._base_coef(a_t)
        c.c_x0 = float(a_prev ** 0.5)
        c.c_dir = float((1 - a_prev - std ** 2) ** 0.5)
        c.c_noise = float(std)
        c.add_noise = int(eta > 0)
        c.use_clipped_model_output = int(bool(use_clipped_model_output))
        c.c_const = float(variance)
        c.c_known = float(a_prev ** 0.5)
        c.c_known_noise = float((1.0 - a_prev) ** 0.5)
        c.known_noise = int(t > 0)
        return c
