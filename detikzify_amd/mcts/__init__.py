from .search import MonteCarlo, Node

__all__ = ["MonteCarlo", "Node"]
