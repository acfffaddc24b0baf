"""
Search-tree value types of the MCTS glue (row a·M; behaviour of reference detikzify/infer/generate.py:35-142):

  NodeState / WideNode   a tree node = token prefix + #lines; every real node owns a "widen" child whose expansion
                         re-rolls from the same prefix (:35-82)
  DynMinMaxNorm          scores are min-max normalised LAZILY against all scores seen so far (:85-142)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Set, Union

import torch

from ..mcts import Node

Numeric = Union[int, float]


@dataclass(frozen=True)
class NodeState:
    token_ids: torch.Tensor
    num_lines: int = 0

    def __eq__(self, other: Any) -> bool:
        try:
            return self.token_ids.equal(other.token_ids)
        except (AttributeError, TypeError):
            return False

    def __hash__(self):
        return hash(tuple(self.token_ids.tolist()))


class WideNode(Node):
    state: NodeState

    def __init__(self, *args, exploration: float = 0.6, is_widen_node: bool = False, **kwargs):
        super().__init__(NodeState(*args, **kwargs))
        self.discovery_factor = exploration
        self.is_widen_node = is_widen_node
        self.update_policy_value(1.0)
        if not is_widen_node:  # the sibling that widens the tree at this prefix
            self.add_child(WideNode(*args, exploration=exploration, is_widen_node=True, **kwargs))

    def add_child(self, child: "WideNode"):
        # only real children make a node "expanded" (selectable for descent)
        self.expanded = self.expanded or not child.is_widen_node
        super().add_child(child)

    @property
    def depth(self) -> int:
        d, cur = 0, self
        while cur.parent is not None:
            d, cur = d + 1, cur.parent
        return d

    @property
    def token_ids(self) -> torch.Tensor:
        return self.state.token_ids

    @property
    def num_lines(self) -> int:
        return self.state.num_lines


class DynMinMaxNorm:
    """normalize(score) returns a lazy value whose `.score` is (s-min)/(max-min) over ALL scores
    registered so far (re-evaluated at read time), summable with further scores / plain numbers."""

    def __init__(self, default_value: Numeric = 0):
        self.scores: Set[Numeric] = set()
        self.default_value = default_value

    def normalize(self, score: Numeric) -> "DynMinMaxNorm.MinMaxScore":
        self.scores.add(score)
        return self.MinMaxScore(score, all_scores=self.scores, default_value=self.default_value)

    __call__ = normalize

    class MinMaxScore:
        def __init__(self, *scores: Numeric, all_scores: Set[Numeric], default_value: Numeric,
                     no_minmax_scores: Optional[List[Numeric]] = None):
            self.scores = list(scores)
            self.all_scores = all_scores
            self.default_value = default_value
            self.no_minmax_scores = list(no_minmax_scores or [])

        @property
        def score(self) -> Numeric:
            lo, hi = min(self.all_scores), max(self.all_scores)
            try:
                value = sum((s - lo) / (hi - lo) for s in self.scores)
            except ZeroDivisionError:
                value = self.default_value
            return value + sum(self.no_minmax_scores)

        def __add__(self, other: Any) -> "DynMinMaxNorm.MinMaxScore":
            merged = type(self)(*self.scores, all_scores=self.all_scores, default_value=self.default_value,
                                no_minmax_scores=self.no_minmax_scores)
            if hasattr(other, "scores") and hasattr(other, "no_minmax_scores"):
                merged.scores.extend(other.scores)
                merged.no_minmax_scores.extend(other.no_minmax_scores)
            else:
                merged.no_minmax_scores.append(other)
            return merged

        def __mul__(self, other: Any):
            return self.score * other

        def __truediv__(self, other: Any):
            return self.score / other

        def __rtruediv__(self, other: Any):
            return other / self.score

        __radd__, __rmul__ = __add__, __mul__
