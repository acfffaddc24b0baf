"""
DetikzifyGenerator — one image, one search tree (rows a·G, a·R, a·M; the reference's semantics,
detikzify/infer/generate.py:145-353), `TikzGenerator` being the alias the north-star names:

  generate()                :209-227  early-out on EOS / max_length, the processor output for (image, text), then
                                      model.generate with bad_words_ids=[[image_token]], begin_suppress_tokens=[eos]
  newlineinfo               :229-244  vocabulary scan: which tokens contain newlines (kept per processor)
  rollout()                 :246-282  generate in a worker thread, split the token stream at newline tokens into
                                      (prefix, #lines) states on the caller's side
  child_finder() / merge()  :305-353  sqrt(n) node insertion, error-line pruning, failed-tail memo, (normalised)
                                      back-propagation
The tree's value types live in infer/tree.py, the sample() / simulate() front-end in infer/pipeline.py, several
trees in one batched decode in infer/batching.py.  The model object is detikzify_amd.model.DetikzifyForCausalLM
(HIP); nothing here touches the GPU directly.  `document_class` selects the reward back-end (TikzDocument =
latexmk, unchanged).
"""
from __future__ import annotations

import re
import threading
from collections import deque
from contextlib import nullcontext
from functools import cached_property
from math import sqrt
from time import time
from types import SimpleNamespace
from typing import Dict, Generator, List, Optional, Tuple, Type

import torch
from PIL import Image

from ..mcts import MonteCarlo
from ..util import ExplicitAbort, StreamerList, TokenStreamer, cache_cast
from ..util import unwrap_processor as unwrap
from .tikz import TikzDocument
from .tree import DynMinMaxNorm, NodeState, Numeric, WideNode


class _BackgroundCall:
    """func(*args, **kwds) in one worker thread; an exception goes to error_callback; wait() joins.  This is what the
    reference gets from `ThreadPool(processes=1).apply_async(...)` + `pending.wait()` (infer/generate.py:248-258) without
    the pool's three housekeeping threads, whose start-up and 0.1 s polling cost 50-100 ms per rollout — nothing next to a
    LaTeX run, but a sixth of a rollout that is decoded in a 64-wide batch."""

    def __init__(self, func, args=(), kwds=None, error_callback=None):
        self._func, self._args, self._kwds, self._error_callback = func, args, kwds or {}, error_callback
        self.value = None
        self._thread = threading.Thread(target=self._run, name="detikzify-rollout", daemon=True)
        self._thread.start()

    def _run(self):
        try:
            self.value = self._func(*self._args, **self._kwds)
        except BaseException as e:  # noqa: BLE001  (forwarded: the consumer of the streamer must never wait forever)
            if self._error_callback is not None:
                self._error_callback(e)

    def wait(self, timeout: Optional[float] = None):
        self._thread.join(timeout)


class DetikzifyGenerator:
    def __init__(self, model, processor, image: Optional[Image.Image], text: Optional[str] = None,
                 metric=None, compile_timeout: Optional[int] = 60, mcts_timeout: Optional[int] = None,
                 streamer=None, control: Optional[ExplicitAbort] = None, exploration: float = 0.6,
                 strict: bool = False, document_class: Type[TikzDocument] = TikzDocument, processed=None, **gen_kwargs):
        self.model, self.processor = model, processor
        self.metric, self.image, self.text = metric, image, text
        self.compile_timeout, self.mcts_timeout = compile_timeout, mcts_timeout
        self.streamer, self.exploration, self.strict = streamer, exploration, strict
        self.document_class = document_class
        self.gen_kwargs = gen_kwargs

        # processor output for (image, text): built on first use — or handed in by a caller that starts many generators on
        # the same image (simulate_parallel: one resize + normalise per image instead of two per tree, all under the GIL
        # right when every tree wants to start)
        assert processed is None or text is None, "a shared processor output is for image-only prompts"
        self._processed = processed
        self.solution: deque = deque(maxlen=1)
        self.failed_rollouts: Dict[NodeState, List[WideNode]] = {}
        self.norm = DynMinMaxNorm()
        self.control = control or ExplicitAbort()
        root_ids = (processed if processed is not None else processor(images=self.image, text=self.text, return_tensors="pt")).input_ids
        self.montecarlo = MonteCarlo(root_node=WideNode(root_ids.to(model.device).squeeze(),
                                                        exploration=self.exploration))
        self.montecarlo.child_finder = self.child_finder
        # memoise by value: token tuples / image bytes (reference :191-192)
        self.decode = cache_cast(lambda token_ids: tuple(token_ids.tolist()))(self.decode)
        self.score = cache_cast(lambda image: image.tobytes())(self.score)

    def __call__(self, *args, **kwargs):
        return self.simulate(*args, **kwargs)

    def simulate(self, expansions: Optional[Numeric] = 1) -> Generator[Tuple[Numeric, TikzDocument], None, None]:
        """One MCTS expansion per iteration; yields every rollout as (score, document)."""
        started = time()
        while expansions is None or (expansions := expansions - 1) >= 0:
            self.montecarlo.simulate()
            yield self.solution.pop()
            if self.mcts_timeout is not None and time() - started > self.mcts_timeout:
                return

    # ---- a·G -------------------------------------------------------------------------------------
    def generate(self, input_ids: torch.Tensor, streamer=None, **gen_kwargs) -> torch.Tensor:
        streamers = StreamerList(filter(bool, [streamer, self.streamer]))
        numel = input_ids.numel()
        max_length = {**self.model.generation_config.to_dict(), **self.gen_kwargs, **gen_kwargs}["max_length"]
        eos = unwrap(self.processor).tokenizer.eos_token_id
        if (numel and input_ids[-1] == eos) or numel >= max_length:
            streamers.end()
            return input_ids  # never continue past EOS / the length budget
        with torch.inference_mode():
            # image and text are fixed for the life of the generator: the reference re-runs the processor on every call
            # (:216), the result is the same tensor every time — keep it (6 ms of resize + normalise per rollout)
            enc = self._processed
            if enc is None:
                enc = self._processed = self.processor(images=self.image, text=self.text, text_kwargs={"truncation": True},
                                                       return_tensors="pt")
            adapter_kwargs = {k: v for k, v in enc.to(self.model.device).items() if k.startswith("adapter")}
            return self.model.generate(
                input_ids=input_ids.unsqueeze(0),
                bad_words_ids=[[self.model.config.image_token_id]],
                begin_suppress_tokens=[self.model.config.text_config.eos_token_id],
                pixel_values=enc.get("pixel_values"),
                streamer=streamers,
                **adapter_kwargs, **self.gen_kwargs, **gen_kwargs,
            ).squeeze()

    # ---- a·R -------------------------------------------------------------------------------------
    @cached_property
    def newlineinfo(self) -> Dict[int, SimpleNamespace]:
        """token id -> (#newlines it contains, whether it ends with one); tokens may hold several."""
        # The table depends on the tokenizer alone: it is kept on the processor, so the trees of simulate_parallel (one
        # DetikzifyGenerator each) build it once instead of once per tree (a vocabulary-sized loop of decode() calls under
        # the tokenizer lock — HF fast tokenizers are not re-entrant).
        owner = unwrap(self.processor)
        with getattr(owner, "_tok_lock", nullcontext()):
            info = getattr(owner, "_newlineinfo", None)
            if info is None:
                info = {}
                for token_id in owner.tokenizer.vocab.values():
                    text = re.sub(r"\r\n|\r", "\n", self.processor.decode([token_id]))
                    if n := text.count("\n"):
                        info[token_id] = SimpleNamespace(num_lines=n, trailing=text.endswith("\n"))
                try:
                    owner._newlineinfo = info
                except AttributeError:      # a processor type that forbids new attributes: per-generator table
                    pass
        assert info
        return info

    def rollout(self, state: NodeState) -> Generator[Tuple[torch.Tensor, int], None, None]:
        """Continue `state` to completion in a worker thread; yield one (prefix, #lines) per
        generated source line as the tokens stream in."""
        input_ids, num_lines, continuation = state.token_ids, state.num_lines, False
        streamer = TokenStreamer(flush_on=self.newlineinfo)      # same tokens, handed over one source line at a time
        pending = _BackgroundCall(
            func=self.generate, args=[input_ids], error_callback=streamer.propagate_error,
            kwds=dict(stopping_criteria=[self.control.reset()], streamer=streamer))
        try:
            prefix, line = input_ids, []
            for token in streamer:
                line.append(token)
                if nl := self.newlineinfo.get(token):
                    # a token may continue with text after its newline ("continuation")
                    num_lines += nl.num_lines - continuation
                    continuation = not nl.trailing
                    prefix = torch.cat((prefix, torch.tensor(line, device=prefix.device)))
                    line.clear()
                    yield prefix, num_lines
            if line:
                yield torch.cat((prefix, torch.tensor(line, device=prefix.device))), num_lines - continuation
        except (GeneratorExit, KeyboardInterrupt):
            self.control.abort()
            raise
        else:
            if self.control.should_stop:
                raise InterruptedError
        finally:
            pending.wait()

    def decode(self, token_ids: torch.Tensor) -> TikzDocument:
        n_prompt = len(self.montecarlo.root_node.token_ids)
        return self.document_class(
            timeout=self.compile_timeout,
            code=self.processor.decode(token_ids[n_prompt:], skip_special_tokens=True))

    def score(self, image: Image.Image) -> Numeric:
        assert self.metric
        self.metric.update(img1=image, img2=self.image, text2=self.text)
        value = self.metric.compute()
        self.metric.reset()
        return value

    def sample(self) -> TikzDocument:
        return self.decode(self.generate(input_ids=self.montecarlo.root_node.token_ids))

    # ---- a·M -------------------------------------------------------------------------------------
    def child_finder(self, node: WideNode, montecarlo: MonteCarlo):
        new_nodes: List[WideNode] = []
        rollout = self.rollout(node.state)
        for state in rollout:
            candidate = WideNode(*state, exploration=self.exploration)
            if candidate.state in self.failed_rollouts:  # known-bad tail: splice it, stop generating
                new_nodes.extend(self.failed_rollouts[candidate.state])
                rollout.close()
                break
            new_nodes.append(candidate)

        if node.is_widen_node:
            node.visits += 1
            node, new_nodes = self.merge(node.parent, new_nodes)

        tikz = self.decode((new_nodes or [node])[-1].token_ids)
        skip_idx = round(sqrt(len(new_nodes)))

        scorable = tikz.is_rasterizable and not (self.strict and tikz.compiled_with_errors)
        if scorable:
            for new_node in new_nodes[:skip_idx]:   # a chain of the first sqrt(n) line-nodes
                node.add_child(new_node)
                node = new_node
        elif errorln := min(tikz.errors or [0]):
            # keep what precedes the first located error; memoise the failing tail
            for idx, new_node in enumerate(new_nodes):
                # NB: the reference looks the 0-dim *tensor* up (generate.py:330); tensors hash by
                # identity, so this is None there — kept verbatim for identical tree statistics.
                ends_with_eol = self.newlineinfo.get(new_node.token_ids[-1])
                if new_node.num_lines < errorln and idx < skip_idx:
                    node.add_child(new_node)
                    node = new_node
                elif new_node.num_lines > errorln or (new_node.num_lines == errorln and ends_with_eol):
                    self.failed_rollouts[new_node.state] = new_nodes[idx:]
                    break

        if self.metric:
            score = self.score(tikz.rasterize()) if scorable else -1
        else:  # compiler diagnostics as the reward
            score = scorable - tikz.compiled_with_errors

        node.update_win_value(self.norm(score) if scorable and self.metric else score)
        self.solution.append((score, tikz))

    def merge(self, node: WideNode, nodes_to_merge: List[WideNode]) -> Tuple[WideNode, List[WideNode]]:
        """walk down existing children while the new chain repeats them"""
        for candidate in list(nodes_to_merge):
            match = next((c for c in node.children if c.state == candidate.state), None)
            if match is None:
                break
            node, nodes_to_merge = match, nodes_to_merge[1:]
        return node, nodes_to_merge


# the name BASELINE.json's north_star uses for the drop-in
TikzGenerator = DetikzifyGenerator
