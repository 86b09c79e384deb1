"""MI355X-native diffusion trajectory denoiser (drop-in for the hot path of
Justin900429/autonomous_driving_with_diffusion_model).

Public surface mirrors the reference's packages:
    modeling.build_model(cfg)                         (reference: modeling/__init__.py)
    scheduler.{GuidanceDDIM,GuidanceDDPM,InpaintingDDIM,InpaintingDDPM}Scheduler, DDPMScheduler
    control.GuidanceLoss
    misc.constant.GuidanceType, misc.load_param.copy_parameters
All compute runs in libadx.so (hand-written HIP for gfx950); there is no CPU fallback.
"""
from . import _lib  # noqa: F401

__all__ = ["modeling", "scheduler", "control", "misc", "config", "sampling"]
