from .carla_dataset import TrajDataset, get_loader, read_waypoint_file

__all__ = ["TrajDataset", "get_loader", "read_waypoint_file"]
