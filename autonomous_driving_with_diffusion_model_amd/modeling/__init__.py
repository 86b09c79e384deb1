from .temporal import TemporalMapUnet, build_model

__all__ = ["build_model", "TemporalMapUnet"]
