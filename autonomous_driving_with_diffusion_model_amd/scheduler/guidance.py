"""GuidanceDDIMScheduler / GuidanceDDPMScheduler (reference: scheduler/guidance_ddim_scheduler.py,
scheduler/guidance_ddpm_scheduler.py): same constructor (`cfg=` + diffusers kwargs) and
`step(model_output, t, sample, ..., target=None, action=None)`; the elementwise math of one step is
one HIP kernel launch.  Deviation, documented: `GuidanceDDPMScheduler(thresholding=True)` raises
NameError in the reference (`np` is not imported, guidance_ddpm_scheduler.py:41); here it performs
the thresholding the code intended, which for sample_max_value = 1 is clamp(-1, 1)."""
from __future__ import annotations

import torch

from ..misc.constant import GuidanceType
from .base import DDIMScheduler, DDPMScheduler, SchedulerOutput, timestep_to_int


def _wants_classifier_guidance(cfg) -> bool:
    return cfg.GUIDANCE.USE_COND == GuidanceType.CLASSIFIER_GUIDANCE.name and cfg.GUIDANCE.LOSS_LIST is not None


class GuidanceDDIMScheduler(DDIMScheduler):
    def __init__(self, cfg, **kwargs):
        super().__init__(**kwargs)
        self.use_classifier_guidance = _wants_classifier_guidance(cfg)
        if self.use_classifier_guidance:
            from ..control import GuidanceLoss
            self.guidance_loss = GuidanceLoss(cfg)

    def guidance_std(self, timestep) -> float:
        """model_std = exp(0.5 * variance) handed to GuidanceLoss (guidance_ddim_scheduler.py:87-91)."""
        t = timestep_to_int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        return float(torch.exp(0.5 * self._get_variance(t, prev_t)))

    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, return_dict: bool = True, target=None, action=None,
             cfg_scale=None, zero_first: bool = False):
        """`cfg_scale` / `zero_first` are optional fusions (classifier-free combine of a [2B] model
        output, interact.py:142-144, and `prev[:, 0, :3] = 0`, interact.py:164); leaving them at their
        defaults gives exactly the reference signature and behaviour."""
        t = timestep_to_int(timestep)
        c = self._ddim_coef(t, eta, use_clipped_model_output)
        if self.use_classifier_guidance and target is not None:
            prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
            with torch.enable_grad():
                model_std = torch.exp(0.5 * self._get_variance(t, prev_t))
                model_output = self.guidance_loss(model_output, action, target, model_std)
        mo, x = self._check_step_inputs(model_output, sample, cfg_scale is not None)
        if cfg_scale is not None:
            c.cfg_combine, c.free_scale = 1, float(cfg_scale)
        c.zero_first = int(zero_first)
        z = None
        if eta > 0:
            if variance_noise is not None and generator is not None:
                raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either "
                                 "`generator` or `variance_noise` stays `None`.")
            z = self._noise(x.shape, generator, x.device, x.dtype, variance_noise)
        prev, x0 = self._launch(False, c, mo, x, z, None, None)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class GuidanceDDPMScheduler(DDPMScheduler):
    def __init__(self, cfg, **kwargs):
        super().__init__(**kwargs)
        self.use_classifier_guidance = _wants_classifier_guidance(cfg)
        if self.use_classifier_guidance:
            from ..control import GuidanceLoss
            self.guidance_loss = GuidanceLoss(cfg)

    def guidance_std(self, timestep) -> float:
        """model_std = exp(0.5 * variance) handed to GuidanceLoss (guidance_ddpm_scheduler.py:94-99)."""
        return float(torch.exp(0.5 * self._get_variance(timestep_to_int(timestep))))

    def step(self, model_output, timestep, sample, generator=None, return_dict: bool = True, target=None,
             action=None, variance_noise=None, cfg_scale=None, zero_first: bool = False):
        t = timestep_to_int(timestep)
        c = self._ddpm_coef(t)
        if self.use_classifier_guidance and target is not None:
            with torch.enable_grad():
                model_std = torch.exp(0.5 * self._get_variance(t))
                model_output = self.guidance_loss(model_output, action, target, model_std)
        mo, x = self._check_step_inputs(model_output, sample, cfg_scale is not None)
        if cfg_scale is not None:
            c.cfg_combine, c.free_scale = 1, float(cfg_scale)
        c.zero_first = int(zero_first)
        z = self._noise(x.shape, generator, x.device, x.dtype, variance_noise) if t > 0 else None
        prev, x0 = self._launch(True, c, mo, x, z, None, None)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)
