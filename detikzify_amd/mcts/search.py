"""
UCT search used by the rollout loop (row a·M): `Node` and `MonteCarlo`, behavioural twins of the MCTS package the
reference vendors (ImparaAI monte-carlo-tree-search 1.3.1; reference detikzify/mcts/node.py:5-70,
detikzify/mcts/montecarlo.py:5-100).  Pinned against the reference's own modules, imported as they are, by
tests/golden/mcts_trace.json (tree statistics, preferred-child order and the consumption of `random` are identical).

Node         score(child) = win_value/visits * (+1|-1 by player) + c * policy * sqrt(ln(parent.visits)/visits) with
             visits == 0 treated as 1, policy None as 1, and NO win term for widen nodes (node.py:51-68);
             back-propagation adds the value to every ancestor (:18-23); ties break by random.choice (:36-49)
MonteCarlo   one simulate() call = `expansion_count` expansions: descend from the root through get_preferred_child
             while nodes are expanded (montecarlo.py:63-64), then expand the leaf with the user's child_finder (:68-85).
             Strictly sequential: expansion k+1 selects on the statistics expansion k back-propagated.
"""
from __future__ import annotations

import random
import time
from math import log, sqrt
from typing import Any, Callable, List, Optional


class Node:
    def __init__(self, state: Any):
        self.state = state
        self.win_value = 0
        self.policy_value: Optional[float] = None
        self.visits = 0
        self.parent: Optional["Node"] = None
        self.children: List["Node"] = []
        self.expanded = False
        self.player_number = None
        self.discovery_factor = 0.35
        self.is_widen_node = False

    # -- statistics ------------------------------------------------------------------------------
    def update_win_value(self, value):
        node: Optional[Node] = self
        while node is not None:          # iterative form of the recursive parent walk
            node.win_value += value
            node.visits += 1
            node = node.parent

    def update_policy_value(self, value):
        self.policy_value = value

    # -- structure -------------------------------------------------------------------------------
    def add_child(self, child: "Node"):
        child.parent = self
        self.children.append(child)

    def add_children(self, children):
        for child in children:
            self.add_child(child)

    # -- selection -------------------------------------------------------------------------------
    def get_score(self, root_node: "Node") -> float:
        n = self.visits or 1
        explore = self.discovery_factor * (self.policy_value or 1) * sqrt(log(self.parent.visits) / n)
        if self.is_widen_node:
            exploit = 0
        else:
            sign = 1 if self.parent.player_number == root_node.player_number else -1
            exploit = sign * self.win_value / n
        self.score = exploit + explore
        return self.score

    def get_preferred_child(self, root_node: "Node") -> "Node":
        best, best_score = [], float("-inf")
        for child in self.children:
            s = child.get_score(root_node)
            if s > best_score:
                best, best_score = [child], s
            elif s == best_score:
                best.append(child)
        return (getattr(root_node, "rng", None) or random).choice(best)       # the search's own stream when it has one (MonteCarlo(rng=...))

    def is_scorable(self) -> bool:
        return bool(self.visits) or self.policy_value is not None


class MonteCarlo:
    def __init__(self, root_node: Node, mins_timeout: Optional[float] = None, rng=None):
        """`rng`: a `random.Random` of this search's own (not in the reference, whose ties break on the process-wide `random`
        module — the default here too).  Searches that run side by side in threads draw from ONE module-level stream in whatever
        order the scheduler gives them; with a stream per search a fixed-seed parallel search no longer depends on thread timing."""
        self.root_node = root_node
        self.rng = rng or random
        if rng is not None:
            root_node.rng = rng
        self.solution = None
        self.child_finder: Optional[Callable[[Node, "MonteCarlo"], None]] = None
        self.node_evaluator: Callable[[Node, "MonteCarlo"], Optional[float]] = lambda child, mc: None
        self.stats_expansion_count = 0
        self.stats_failed_expansion_count = 0
        self.mins_timeout = mins_timeout

    # -- choices at the root ---------------------------------------------------------------------
    def make_choice(self) -> Node:
        top = max(child.visits for child in self.root_node.children)
        return self.rng.choice([c for c in self.root_node.children if c.visits == top])

    def make_exploratory_choice(self) -> Optional[Node]:
        threshold, acc = self.rng.uniform(0, 1), 0.0
        for child in self.root_node.children:
            p = child.visits / self.root_node.visits
            if acc + p >= threshold:
                return child
            acc += p
        return None

    # -- search ----------------------------------------------------------------------------------
    def simulate(self, expansion_count: Optional[int] = 1):
        started = time.time()
        done = 0
        while expansion_count is None or done < expansion_count:
            done += 1
            if self.solution is not None:
                return
            if self.mins_timeout is not None and time.time() - started > self.mins_timeout * 60:
                print("reached timelimit, stopping expansion on current node")
                return
            node = self.root_node
            while node.expanded:
                node = node.get_preferred_child(self.root_node)
            self.expand(node)

    def expand(self, node: Node):
        self.stats_expansion_count += 1
        self.child_finder(node, self)
        for child in node.children:
            value = self.node_evaluator(child, self)
            if value is not None:
                child.update_win_value(value)
            if not child.is_scorable():
                self.random_rollout(child)
                child.children = []
        if node.children:
            node.expanded = True
        else:
            self.stats_failed_expansion_count += 1

    def random_rollout(self, node: Node):
        self.child_finder(node, self)
        child = self.rng.choice(node.children)
        node.children = []
        node.add_child(child)
        value = self.node_evaluator(child, self)
        if value is not None:
            node.update_win_value(value)
        else:
            self.random_rollout(child)
