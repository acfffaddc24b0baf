"""cache_cast — memoise a bound method whose arguments are unhashable (token tensors, images).
Same contract as reference detikzify/util/functools.py:7-23: the user-supplied cast turns the
arguments into a hashable key; used on DetikzifyGenerator.decode / .score (infer/generate.py:191-192)."""
from __future__ import annotations

from functools import wraps
from typing import Any, Callable, Dict


def cache_cast(cast_func: Callable[..., Any]):
    def decorator(func):
        memo: Dict[Any, Any] = {}

        @wraps(func)
        def wrapped(*args, **kwargs):
            key = cast_func(*args, **kwargs)
            if key not in memo:
                memo[key] = func(*args, **kwargs)
            return memo[key]

        wrapped.cache = memo  # type: ignore[attr-defined]
        return wrapped

    return decorator
