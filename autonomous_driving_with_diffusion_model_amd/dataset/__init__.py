from .augment import GpuAugmentor, augment_factors, sample_plan
from .carla_dataset import TrajDataset, get_loader, read_waypoint_file

__all__ = ["TrajDataset", "get_loader", "read_waypoint_file", "GpuAugmentor", "augment_factors", "sample_plan"]
