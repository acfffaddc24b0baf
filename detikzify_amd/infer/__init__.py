from .compile_pool import CompilePool, pooled_document_class
from .generate import DetikzifyGenerator, TikzGenerator
from .pipeline import DetikzifyPipeline
from .tikz import SyntheticTikzDocument, TikzDocument
from .tree import DynMinMaxNorm, NodeState, WideNode

__all__ = ["CompilePool", "pooled_document_class", "DetikzifyGenerator", "DetikzifyPipeline", "DynMinMaxNorm", "NodeState", "TikzGenerator",
           "WideNode", "TikzDocument", "SyntheticTikzDocument"]
