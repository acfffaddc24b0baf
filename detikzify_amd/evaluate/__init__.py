from .imagesim import ImageSim

__all__ = ["ImageSim"]
