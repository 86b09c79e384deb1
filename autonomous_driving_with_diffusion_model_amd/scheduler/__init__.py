from .base import DDIMScheduler, DDPMScheduler, SchedulerOutput, TimestepSequence
from .guidance import GuidanceDDIMScheduler, GuidanceDDPMScheduler
from .inpainting import InpaintingDDIMScheduler, InpaintingDDPMScheduler

__all__ = [
    "GuidanceDDIMScheduler",
    "GuidanceDDPMScheduler",
    "InpaintingDDIMScheduler",
    "InpaintingDDPMScheduler",
    "DDPMScheduler",
    "DDIMScheduler",
]
