"""
Monte-Carlo tree search driver (behavioural twin of reference detikzify/mcts/montecarlo.py:5-100,
pinned by tests/golden/mcts_trace.json).  One simulate() call = `expansion_count` expansions:
descend from the root through get_preferred_child while nodes are expanded (:63-64), then expand
the leaf with the user's child_finder (:68-85).  The search is strictly sequential: expansion k+1
selects on the statistics expansion k back-propagated.
"""
from __future__ import annotations

import random
import time
from typing import Callable, Optional

from .node import Node


class MonteCarlo:
    def __init__(self, root_node: Node, mins_timeout: Optional[float] = None):
        self.root_node = root_node
        self.solution = None
        self.child_finder: Optional[Callable[[Node, "MonteCarlo"], None]] = None
        self.node_evaluator: Callable[[Node, "MonteCarlo"], Optional[float]] = lambda child, mc: None
        self.stats_expansion_count = 0
        self.stats_failed_expansion_count = 0
        self.mins_timeout = mins_timeout

    # -- choices at the root ---------------------------------------------------------------------
    def make_choice(self) -> Node:
        top = max(child.visits for child in self.root_node.children)
        return random.choice([c for c in self.root_node.children if c.visits == top])

    def make_exploratory_choice(self) -> Optional[Node]:
        threshold, acc = random.uniform(0, 1), 0.0
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
        child = random.choice(node.children)
        node.children = []
        node.add_child(child)
        value = self.node_evaluator(child, self)
        if value is not None:
            node.update_win_value(value)
        else:
            self.random_rollout(child)
