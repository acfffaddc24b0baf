from .generate import (DetikzifyGenerator, DetikzifyPipeline, DynMinMaxNorm, NodeState, TikzGenerator,
                       WideNode)
from .tikz import SyntheticTikzDocument, TikzDocument

__all__ = ["DetikzifyGenerator", "DetikzifyPipeline", "DynMinMaxNorm", "NodeState", "TikzGenerator",
           "WideNode", "TikzDocument", "SyntheticTikzDocument"]
