from .functools import cache_cast
from .generation import ExplicitAbort, StreamerList, TextIteratorStreamer, TokenStreamer, unwrap_processor
from .image import expand, load, remove_alpha, trim
from .subprocess import check_output

__all__ = ["cache_cast", "ExplicitAbort", "StreamerList", "TextIteratorStreamer", "TokenStreamer",
           "unwrap_processor", "expand", "load", "remove_alpha", "trim", "check_output"]
