from .generate import DetikzifyGenerator, TikzGenerator
from .pipeline import DetikzifyPipeline
from .tikz import SyntheticTikzDocument, TikzDocument
from .tree import DynMinMaxNorm, NodeState, WideNode

__all__ = ["DetikzifyGenerator", "DetikzifyPipeline", "DynMinMaxNorm", "NodeState", "TikzGenerator",
           "WideNode", "TikzDocument", "SyntheticTikzDocument"]
