"""
Value types of the search tree (row a·M).  Written from the behaviour the golden traces pin
(tests/golden/generator_trace.json, produced by running the reference's detikzify/infer/generate.py); what the
reference defines at generate.py:35-142 and callers rely on:

  NodeState     a tree position: the token prefix and the number of source lines in it.  Two states are the same
                position iff their token sequences are equal; the line count does not take part.
  WideNode      an MCTS node over a NodeState.  Every ordinary node is born with one "widen" child over the SAME
                prefix: expanding that child re-rolls from the prefix and so widens the tree there.  A node only
                counts as expanded (eligible for descent) once it has an ordinary child.
  DynMinMaxNorm rewards are min-max normalised against ALL rewards seen so far — including ones that arrive later —
                so what is back-propagated is a lazy value, evaluated when the selection reads it.

Layout choices here (not the reference's): a state caches its token tuple once (hash and equality are then tuple
operations instead of a tensor comparison plus a fresh `.tolist()` per dictionary probe: `failed_rollouts` and
`merge` probe once per generated line); the lazy reward is an immutable two-tuple ledger (rewards to normalise, plain
addends) bound to its normaliser.
"""
from __future__ import annotations

from typing import Any, Iterable, Optional, Tuple, Union

import torch

from ..mcts import Node

Numeric = Union[int, float]


class NodeState:
    """(token_ids, num_lines); identity = the token sequence."""

    __slots__ = ("token_ids", "num_lines", "_key")

    def __init__(self, token_ids: torch.Tensor, num_lines: int = 0):
        self.token_ids = token_ids
        self.num_lines = num_lines
        self._key: Optional[Tuple[int, ...]] = None

    @property
    def key(self) -> Tuple[int, ...]:
        k = self._key
        if k is None:
            ids = self.token_ids
            k = self._key = tuple(ids.tolist()) if ids.dim() else (ids.item(),)
        return k

    def __eq__(self, other: Any) -> bool:
        if isinstance(other, NodeState):
            return self.key == other.key
        ids = getattr(other, "token_ids", None)     # duck-typed states (the reference compares tensors)
        return isinstance(ids, torch.Tensor) and self.token_ids.shape == ids.shape and bool((self.token_ids == ids).all())

    def __ne__(self, other: Any) -> bool:
        return not self == other

    def __hash__(self) -> int:
        return hash(self.key)

    def __iter__(self):                             # unpacks like the (ids, lines) pairs rollout() yields
        yield self.token_ids
        yield self.num_lines

    def __repr__(self) -> str:
        return f"NodeState({len(self.key)} tokens, {self.num_lines} lines)"


class WideNode(Node):
    state: NodeState

    def __init__(self, token_ids: torch.Tensor, num_lines: int = 0, *, exploration: float = 0.6,
                 is_widen_node: bool = False):
        super().__init__(NodeState(token_ids, num_lines))
        self.discovery_factor = exploration
        self.is_widen_node = is_widen_node
        self.update_policy_value(1.0)
        if not is_widen_node:
            self._attach_widener()

    def _attach_widener(self):
        twin = WideNode(self.state.token_ids, self.state.num_lines, exploration=self.discovery_factor, is_widen_node=True)
        self.add_child(twin)

    def add_child(self, child: "WideNode"):
        if not child.is_widen_node:     # the widen twin alone never makes a node selectable for descent
            self.expanded = True
        super().add_child(child)

    def ancestors(self) -> Iterable["WideNode"]:
        node = self.parent
        while node is not None:
            yield node
            node = node.parent

    @property
    def depth(self) -> int:
        return sum(1 for _ in self.ancestors())

    @property
    def token_ids(self) -> torch.Tensor:
        return self.state.token_ids

    @property
    def num_lines(self) -> int:
        return self.state.num_lines


class LazyReward:
    """Σ normalised(reward_i) + Σ plain_j, evaluated against the normaliser's CURRENT extremes whenever it is read.
    The MCTS only ever adds these up (back-propagation: `win_value += value`, starting from the int 0) and divides or
    multiplies the total by plain numbers (UCT: `win_value / visits`), so those are the operations it has; products and
    quotients are plain floats."""

    __slots__ = ("_norm", "_rewards", "_plain")

    def __init__(self, norm: "DynMinMaxNorm", rewards: Tuple[Numeric, ...], plain: Tuple[Numeric, ...] = ()):
        self._norm, self._rewards, self._plain = norm, rewards, plain

    # the reference's attribute names, for code that inspects a back-propagated value
    @property
    def scores(self):
        return list(self._rewards)

    @property
    def no_minmax_scores(self):
        return list(self._plain)

    @property
    def score(self) -> Numeric:
        lo, hi = self._norm.bounds()
        span = hi - lo
        if self._rewards and span == 0:             # one distinct reward so far: nothing to scale against
            total = self._norm.default_value
        else:
            total = 0
            for r in self._rewards:                 # left to right, one quotient per reward (same rounding as the reference)
                total = total + (r - lo) / span
        extra = 0
        for p in self._plain:
            extra = extra + p
        return total + extra

    def __add__(self, other: Any) -> "LazyReward":
        if isinstance(other, LazyReward):
            return LazyReward(self._norm, self._rewards + other._rewards, self._plain + other._plain)
        return LazyReward(self._norm, self._rewards, self._plain + (other,))

    __radd__ = __add__

    def __mul__(self, factor: Any):
        return self.score * factor

    __rmul__ = __mul__

    def __truediv__(self, divisor: Any):
        return self.score / divisor

    def __rtruediv__(self, dividend: Any):
        return dividend / self.score

    def __repr__(self) -> str:
        return f"LazyReward({self.score!r})"


class DynMinMaxNorm:
    """norm(reward) registers the reward and returns its lazily normalised value."""

    MinMaxScore = LazyReward        # the name the reference exposes (generate.py:119)

    def __init__(self, default_value: Numeric = 0):
        self.scores = set()
        self.default_value = default_value

    def bounds(self) -> Tuple[Numeric, Numeric]:
        return min(self.scores), max(self.scores)

    def normalize(self, score: Numeric) -> LazyReward:
        self.scores.add(score)
        return LazyReward(self, (score,))

    __call__ = normalize
