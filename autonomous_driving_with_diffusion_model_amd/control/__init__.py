from .guidance import GuidanceLoss
from .guidance_loss import TargetGuidance

__all__ = ["GuidanceLoss", "TargetGuidance"]
