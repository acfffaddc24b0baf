from .montecarlo import MonteCarlo
from .node import Node

__all__ = ["MonteCarlo", "Node"]
