"""InpaintingDDIMScheduler / InpaintingDDPMScheduler (reference: scheduler/inpainting_ddim_scheduler.py,
scheduler/inpainting_ddpm_scheduler.py): RePaint-style blend of a known trajectory into every step.
Exported by the reference but not instantiated by any caller; kept to the same signature, including
the reference's quirk of adding the *scalar* DDIM variance to every element (:108-112,124-128)."""
from __future__ import annotations

from .base import DDIMScheduler, DDPMScheduler, SchedulerOutput, timestep_to_int
from .. import _lib as L


def _known(target_traj, target_mask, x):
    if target_traj is None or target_mask is None:
        return None, None
    tt = L.require_gpu_f32(target_traj.expand_as(x), "target_traj")
    tm = L.require_gpu_f32(target_mask.expand_as(x), "target_mask")
    return tt, tm


class InpaintingDDIMScheduler(DDIMScheduler):
    def step(self, model_output, timestep, sample, eta: float = 0.0, use_clipped_model_output: bool = False,
             generator=None, variance_noise=None, target_traj=None, target_mask=None, return_dict: bool = True):
        t = timestep_to_int(timestep)
        c = self._ddim_coef(t, eta, use_clipped_model_output)
        c.inpaint = 1
        mo, x = self._check_step_inputs(model_output, sample)
        tt, tm = _known(target_traj, target_mask, x)
        if eta > 0 and variance_noise is not None and generator is not None:
            raise ValueError("Cannot pass both generator and variance_noise. Please make sure that either "
                             "`generator` or `variance_noise` stays `None`.")
        z = None
        if tt is not None or eta > 0:
            # the reference draws the RePaint noise first and, when eta > 0 and no variance_noise was
            # given, a second independent tensor for the eta term; with variance_noise both are that tensor
            z = self._noise(x.shape, generator, x.device, x.dtype, variance_noise)
        prev, x0 = self._launch(False, c, mo, x, z, tt, tm)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)


class InpaintingDDPMScheduler(DDPMScheduler):
    def step(self, model_output, timestep, sample, generator=None, variance_noise=None, target_traj=None,
             target_mask=None, return_dict: bool = True):
        t = timestep_to_int(timestep)
        c = self._ddpm_coef(t)
        c.inpaint = 1
        mo, x = self._check_step_inputs(model_output, sample)
        tt, tm = _known(target_traj, target_mask, x)
        z = self._noise(x.shape, generator, x.device, x.dtype, variance_noise)  # drawn even at t == 0 (:100-109)
        prev, x0 = self._launch(True, c, mo, x, z, tt, tm)
        if not return_dict:
            return (prev,)
        return SchedulerOutput(prev_sample=prev, pred_original_sample=x0)
