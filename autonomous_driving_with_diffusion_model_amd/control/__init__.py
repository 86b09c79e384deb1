from .controller import Controller, post_process_control
from .guidance import GuidanceLoss
from .guidance_loss import TargetGuidance
from .pid import PIDController

__all__ = ["GuidanceLoss", "TargetGuidance", "Controller", "PIDController", "post_process_control"]
