"""
Batched decode of independent rollouts on ONE GPU (north-star: "independent rollouts
... as embarrassingly-parallel batched decodes").  The reference's tree search is sequential, so what
can be batched without touching its semantics are *independent trees* (root parallelisation, the same
divergence as sharding trees across GPUs): every tree runs the reference's logic unchanged in its own
thread (DetikzifyGenerator.rollout already generates in a worker thread, infer/generate.py:248-258), and
whenever those threads ask for their next token the BatchEngine issues ONE dtk_decode_batch step for all
of them: the weights are streamed once per step for up to 64 sequences (bytes/step = W + sum_b K*t_b).

BatchEngine      lock-step scheduler over the C ABI's slots (dtk_prefill_slot / dtk_decode_batch_*)
simulate_parallel / simulate_parallel_images   independent DetikzifyGenerator trees on one / several images
"""
from __future__ import annotations

import os
import queue
import random
import threading
import time
from collections import OrderedDict
from contextlib import contextmanager
from typing import Any, Callable, Dict, Iterator, List, Optional, Tuple


_NUDGE = object()      # wakes a thread that waits on its slot's queue without handing it a token


class _Sequence:
    def __init__(self, engine: "BatchEngine", slot: int):
        self.engine, self.slot = engine, slot

    def next_token(self) -> int:
        """pull: the calling thread gets the slot's next token (one thread hand-off per token)"""
        return self.engine._next_token(self.slot)

    def run(self, emit: Callable[[int], bool]):
        """push: `emit(token) -> stop` is called for every token of this sequence by whichever thread drives the decode
        steps; returns when emit returned True (or re-raises what it raised).  No thread hand-off per token: with dozens
        of sequences in one batch that hand-off — not the GPU step — is what bounds the step rate under the GIL."""
        self.engine._run(self.slot, emit)


class BatchEngine:
    """All methods are thread-safe; one engine per model.  A sequence is 'active' from the end of its
    prefill until it leaves.  Steps are pipelined one deep: as soon as step k has been collected, step k+1 is
    launched for the same slots, so the per-token host work (streamer, stopping criteria, Python) runs under the
    GPU's next step; a sequence that stops after token k just discards its token of step k+1 (its slot is
    recycled once that step has completed).

    Two ways to consume a sequence (they mix freely in one batch):
      * push — `seq.run(emit)`, what model.generate uses: emit(token) -> stop is called by the thread that drives
        the steps (the thread of one of the push sequences; the wheel is handed over when its sequence ends).  No
        thread is woken per token: under the GIL a hand-off per token per sequence costs more than a 64-slot GPU
        step hides.
      * pull — `seq.next_token()`: tokens travel through one SimpleQueue per slot (not through the shared condition
        variable: a notify_all on one Condition cost ~1.3 ms of lock / GIL hand-offs per step with 32 threads) and
        the last thread to run out of tokens drives the step.
    Whoever drives holds `cv`, which also serialises every other use of the context's main stream (prefill of a
    joining sequence); joins and leaves announce themselves (`pending`) and the driver lets them in between steps.
    The SelfSim ViT passes do not take `cv`: they run on their own stream under the model's ViT lock."""

    def __init__(self, model, max_batch: Optional[int] = None, share_prefix: bool = True, pipeline: bool = True,
                 gather: int = 0, gather_timeout: float = 0.5, prefix_slots: Optional[int] = None, resume_in_place: bool = True):
        n = model.num_slots()
        if n <= 0:
            raise ValueError("model was loaded without batch slots (load(..., batch_slots=N))")
        self.model = model
        # slots that can take part in a step (include/dtk.h): 4 of a context with <= 5 slots (multi-vector kernels), else one /
        # two / four 16-slot MFMA column tiles
        probe = getattr(model, "max_decode_slots", None)
        dec = int(probe()) if callable(probe) else min(n, 64 if n > 33 else (32 if n > 17 else 16))
        self.share_prefix = share_prefix and n >= 2
        self.capacity = min(dec, max_batch) if max_batch else (dec if n > dec else dec - 1)
        # Slots the decode batch does not need are the prefix cache (the highest indices: a step only runs the column
        # tiles up to its highest ACTIVE slot): each holds the KV of one image prefix ([image_token]*n + pixels), least
        # recently used first out; sequences fork it (bit-identical KV) and only prefill what follows.  With
        # one image one slot is enough; a batch of images (BASELINE config 5) wants one per image in flight
        # (load(batch_slots = rollouts + images)).  Without any spare slot the sequences still share prefixes among
        # themselves (in place / from a donor slot); only the first rollout of an image prefills in full.
        spare = n - self.capacity
        n_prefix = min(spare, 8 if prefix_slots is None else int(prefix_slots)) if self.share_prefix else 0
        self.prefix_slots: List[int] = list(range(n - n_prefix, n))
        self.prefix_cache: "OrderedDict[Tuple[int, int], int]" = OrderedDict()     # (image key, prefix length) -> prefix slot, LRU order
        self.slot_img: Dict[int, Tuple[int, int]] = {}      # slot -> (image key, prefix length) whose KV prefix it still holds
        self.joins = 0
        # A sequence whose prompt (all but its last token) is still in a free slot's KV cache — an MCTS tree coming back to a node
        # of its own previous rollout — continues there without a prefill: the slot joins the batch at once and its first step
        # forwards the last prompt token instead of sampling (dtk_resume_slot).  The reused rows are the ones the earlier
        # sequence wrote (prefill GEMMs or decode kernels), as with any same-slot prefix reuse; resume_in_place=False makes
        # every join fork the image prefix and prefill what follows, which stalls the batch ~10 ms per join.
        self.resume_in_place = resume_in_place
        self.resumes = 0
        self.last_slot: Dict[int, int] = {}                  # sequence owner -> the slot its last sequence used
        self.skip_first: set = set()                         # resumed slots whose next token is the forced prompt token: not emitted
        self.inplace_reuses = 0                              # joins that found their image prefix already in their slot
        self.prefix_encodes = 0                              # ViT + prefix prefills run for the prefix cache (diagnostics)
        self.pipeline = pipeline
        self.cv = threading.Condition()
        self.free: List[int] = list(range(self.capacity))
        self.active: set = set()
        self.ready: set = set()
        self.tokq: List["queue.SimpleQueue"] = [queue.SimpleQueue() for _ in range(n)]   # per-slot token hand-off
        self.sinks: Dict[int, Callable[[int], bool]] = {}   # push sequences: slot -> emit(token) -> stop
        self.finished: Dict[int, Optional[BaseException]] = {}   # push sequences that are done (what their emit raised)
        self.driver: Optional[int] = None               # the push sequence whose thread currently drives the steps
        self.pending = 0                                # threads that want `cv` (joins, leaves): the driver lets them in
        self._plock = threading.Lock()
        self.inflight: Optional[List[int]] = None      # slots of the launched, not yet collected step
        self.zombies: set = set()                      # left while in flight: freed when that step completes
        self.error: Optional[BaseException] = None
        self.steps = 0
        self.tokens_out = 0
        self.t_wait = self.t_launch = self.t_prefill = 0.0     # seconds inside the native calls (diagnostics)
        self.host_bound_steps = 0
        self.t_idle = 0.0                               # seconds with sequences alive but no step in flight (joins, rewards, warm start)
        self.slot_steps_short = 0                       # steps that ran with fewer than half of the decode slots active
        self._t_free_since = time.perf_counter()        # stamp of the last moment nothing was in flight (start, or a collect)
        self.t_first_launch = self.t_last_collect = None        # perf_counter stamps (diagnostics; reset by expect())
        # warm start: hold the first step until `gather` sequences have joined (rollouts started together should not
        # trickle in one pipeline flush at a time); gives up after gather_timeout seconds
        self.gather_left = min(int(gather), self.capacity)
        self.gather_deadline = time.perf_counter() + gather_timeout
        model.batch_engine = self

    @contextmanager
    def _locked(self):
        """`cv` for everybody but the driving thread: Python locks are not fair, and a driver that re-takes `cv` for the
        next step the moment it has released it would starve joining / leaving sequences — it waits while `pending` > 0"""
        with self._plock:
            self.pending += 1
        try:
            with self.cv:
                yield
        finally:
            with self._plock:
                self.pending -= 1

    def expect(self, n: int, timeout: float = 0.5):
        """n sequences are about to join (e.g. the trees of simulate_parallel): no decode step before all of
        them have been prefilled / forked, or `timeout` seconds have passed"""
        with self._locked():
            self.gather_left = min(int(n), self.capacity)
            self.gather_deadline = time.perf_counter() + timeout
            self.t_first_launch = self.t_last_collect = None
            self._t_free_since = time.perf_counter()

    def stats(self) -> Dict[str, Any]:
        return {"engine": "python", "steps": self.steps, "tokens_out": self.tokens_out, "wait_s": round(self.t_wait, 3), "launch_s": round(self.t_launch, 3),
                "prefill_s": round(self.t_prefill, 3), "host_bound_steps": self.host_bound_steps, "prefix_encodes": self.prefix_encodes,
                "inplace_reuses": self.inplace_reuses, "joins": self.joins, "resumed_in_place": self.resumes, "idle_between_steps_s": round(self.t_idle, 3),
                "steps_below_half_occupancy": self.slot_steps_short}

    def close(self):
        with self._locked():
            self._collect()
        self.model.batch_engine = None

    @contextmanager
    def sequence(self, ids, pixel_values, sampling: Dict[str, Any], owner: Optional[int] = None, **_native_only) -> Iterator[_Sequence]:
        """`owner`: whoever starts sequence after sequence (an MCTS tree: generate(sequence_owner=t)) — a join that cannot
        resume takes the slot its owner used last, so it does not overwrite a rollout ANOTHER owner may come back to.
        (max_new_tokens / stop_ids / per_token are for infer/engine.NativeBatchEngine: here the sequence's end is emit's verdict.)"""
        want = self._prefix_key(ids, pixel_values) if (self.share_prefix and pixel_values is not None) else None
        with self._locked():
            while not self.free:
                with self._plock:       # waiting for a slot is not wanting the lock: the driver must keep stepping
                    self.pending -= 1
                try:
                    self.cv.wait()
                finally:
                    with self._plock:
                        self.pending += 1
            if self.zombies and self.error is None:
                self._collect()     # a slot left while its last step was in flight is free once that step is read — and it may
                                    # be the very slot whose cache holds this prompt (a tree returning right after its rollout)
            # slot choice: a free slot that holds no prefix worth keeping; else one that already holds THIS image's prefix
            # (re-used in place); else evict the prefix of some other image
            # (lowest index first: a step only runs the 16-slot column tiles up to its highest active slot)
            order = sorted(self.free)
            slot, resume = None, False
            n_ids = int(ids.numel())
            # (an image position takes the projected patch feature, not a token embedding: the forced token must be text)
            if (self.resume_in_place and n_ids >= 2 and (want is not None or pixel_values is None)
                    and int(ids.reshape(-1)[-1]) != self.model.config.image_token_id):
                key = want[0] if want is not None else 0
                # an owner (an MCTS tree) resumes only in the slot IT used last: whether a join resumes or re-prefills must not
                # depend on which other slots happen to be free at that moment (thread timing) — resumed rows were written by
                # the decode kernels, prefilled rows by the GEMMs, so the choice is visible in the last bits of the logits and
                # a fixed-seed parallel search would stop being reproducible.  Owner-less sequences may take any free slot.
                if owner is not None:
                    candidates = [self.last_slot[owner]] if self.last_slot.get(owner) in self.free else []
                else:
                    candidates = order
                best = self.model.best_lcp_slot(candidates, ids, key) if candidates else None
                if best is not None and best[1] >= n_ids - 1:
                    slot, resume = best[0], True
            if slot is None and owner is not None and self.last_slot.get(owner) in self.free:
                slot = self.last_slot[owner]
            if slot is None:
                slot = next((f for f in order if f not in self.slot_img), None)
            if slot is None and want is not None:
                slot = next((f for f in order if self.slot_img.get(f) == want), None)
            if slot is None:
                slot = order[0]
            self.free.remove(slot)
            if owner is not None:
                self.last_slot[owner] = slot
        joined = False
        try:
            with self._locked():   # prefill needs the context exclusively (same stream as the decode steps)
                if self.error is not None:
                    raise self.error        # a native call has failed: the context is not to be touched again
                self._collect()     # a prefill drops un-collected steps on the C side: collect first
                if self.error is not None:
                    raise self.error
                t0 = time.perf_counter()
                self.model.set_sampling(slot=slot, **sampling)
                if resume:
                    self.model.resume_slot(slot, ids, want[0] if want is not None else 0)
                    self.skip_first.add(slot)
                    self.resumes += 1
                    forked = 2
                else:
                    forked = self._fork_prefix(slot, ids, pixel_values, want) if want is not None else 0
                    if want is None:
                        self.slot_img.pop(slot, None)
                if forked == 2:
                    pass    # the prompt IS the prefix (rollout from the root): KV and logits were forked, nothing to run
                elif forked:
                    self.model.prefill(ids, pixel_values, slot=slot, reuse=True)    # only the tail beyond the prefix
                else:
                    self.model.prefill(ids, pixel_values, slot=slot)
                self.t_prefill += time.perf_counter() - t0
                q = self.tokq[slot]
                while not q.empty():       # leftovers of the slot's previous sequence
                    q.get_nowait()
                self.active.add(slot)
                self.joins += 1
                self.gather_left = max(0, self.gather_left - 1)
                joined = True
                if self.driver is not None:
                    self.tokq[self.driver].put(_NUDGE)      # it may be sleeping through the warm start
            yield _Sequence(self, slot)
        finally:
            with self._locked():
                self.skip_first.discard(slot)
                if joined:
                    self.active.discard(slot)
                    self.ready.discard(slot)
                    if self.inflight is not None and slot in self.inflight:
                        self.zombies.add(slot)      # recycled by _collect()
                    else:
                        self.free.append(slot)
                    if not self.active:
                        self._collect()             # nobody left to collect the speculative step
                    elif self.driver is None:
                        self._maybe_step()          # the others may all be waiting on this one
                else:
                    self.slot_img.pop(slot, None)   # the prefill did not complete: the slot holds nobody's prefix
                    self.free.append(slot)
                self.cv.notify_all()   # a slot may have become free

    # -- called with self.cv held ---------------------------------------------------------------------
    def _prefix_key(self, ids, pixel_values):
        """(image key, prefix length) if the prompt starts with its image-token run, else None"""
        tok = self.model.config.image_token_id
        ids = ids.reshape(-1)
        n_img = int((ids == tok).sum())
        if n_img == 0 or not bool((ids[:n_img] == tok).all()):
            return None                       # the image run is not a leading prefix: no sharing
        return (self.model.image_key(pixel_values), n_img)

    def _fork_prefix(self, slot: int, ids, pixel_values, key) -> int:
        """Give `slot` the KV of its image prefix without running ViT + prefill again.  Returns 1 (prefix KV in place: the
        caller prefills what follows with reuse), 2 (prefix == whole prompt and the next-token logits were forked too) or
        0 (nobody holds this image and there is no prefix-cache slot: the caller prefills in full and becomes a donor).
        Sources, in order: a prefix-cache slot that holds this image (its fork also carries the logits); the slot itself
        if it still holds this image's prefix from its previous sequence; any other slot that still holds this image's
        prefix; otherwise the image is encoded into a free — else the least recently used — prefix-cache slot.  A batch of
        8 images x 4 rollouts encodes each image once either way; with a prefix-cache slot per image the joins of later
        rollouts are pure forks (no 1-token tail prefill to recover the logits)."""
        ids = ids.reshape(-1)
        n_img = key[1]
        src = self.prefix_cache.get(key)
        if src is None and self.slot_img.get(slot) == key:
            self.inplace_reuses += 1
            return 1                                        # in place: dtk_prefill_slot(REUSE_PREFIX) keeps the prefix rows
        self.slot_img[slot] = key
        if src is None:
            donor = next((s for s, k in self.slot_img.items() if k == key and s != slot), None)
            if donor is not None:
                self.model.kv_fork(donor, slot, n_img)      # the donor has decoded past the prefix: no logits to inherit
                return 1
            self.prefix_encodes += 1
            if not self.prefix_slots:
                return 0
            used = set(self.prefix_cache.values())
            src = next((s for s in reversed(self.prefix_slots) if s not in used), None)
            if src is None:
                _, src = self.prefix_cache.popitem(last=False)      # evict the least recently used image
            self.model.set_sampling(slot=src, do_sample=False)
            self.model.prefill(ids[:n_img], pixel_values, slot=src)
            self.prefix_cache[key] = src
        self.prefix_cache.move_to_end(key)
        self.model.kv_fork(src, slot, n_img)
        return 2 if ids.numel() == n_img else 1

    @property
    def prefix_slot(self) -> Optional[int]:
        """the (first-used) prefix-cache slot, None if the engine has none"""
        return self.prefix_slots[-1] if self.prefix_slots else None

    def _launch(self):
        # a slot whose context is full cannot take another (speculative) step; its sequence is at max_length
        lim = self.model.config.max_positions
        slots = [s for s in sorted(self.active) if self.model.lib.dtk_context_len_slot(self.model._ctx, s) < lim]
        if not slots or self.error is not None:
            return
        try:
            t0 = time.perf_counter()
            if self.t_first_launch is None:
                self.t_first_launch = t0
            if self._t_free_since is not None:
                self.t_idle += t0 - self._t_free_since
                self._t_free_since = None
            if 2 * len(slots) < self.capacity:
                self.slot_steps_short += 1
            self.model.decode_batch_launch(slots)
            self.t_launch += time.perf_counter() - t0
            self.inflight = slots
        except BaseException as e:
            self._fail(e)

    def _fail(self, e: BaseException):
        self.error = e
        for s in self.active:           # wake every waiting sequence with the error
            self.tokq[s].put(e)
        self.cv.notify_all()

    def _collect(self, dispatch: bool = True):
        """wait for the in-flight step, hand its tokens to the sequences that are still active (push sequences: through
        _dispatch, here or — dispatch=False — by the caller once it has launched the next step)"""
        pushed: List[Tuple[int, int]] = []
        if self.inflight is None:
            return pushed
        try:
            t0 = time.perf_counter()
            toks = self.model.decode_batch_wait()
            dt = time.perf_counter() - t0
            self.t_wait += dt
            self.t_last_collect = t0 + dt
            if dt < 1e-4:
                self.host_bound_steps += 1      # the GPU had already finished: this step waited for the host
            for s in self.inflight:
                if s in self.active:
                    if s in self.skip_first:        # the forced last prompt token of a resumed slot: already part of the prompt
                        self.skip_first.discard(s)
                        continue
                    if s in self.sinks:
                        pushed.append((s, toks[s]))
                    else:
                        self.tokq[s].put(toks[s])
                    self.tokens_out += 1
            self.steps += 1
        except BaseException as e:
            self._fail(e)
        for s in self.inflight:
            if s in self.zombies:
                self.zombies.discard(s)
                self.free.append(s)
        self.inflight = None
        self._t_free_since = time.perf_counter()      # cleared by the next launch (pipelined steps: microseconds later)
        self.cv.notify_all()
        if dispatch:
            self._dispatch(pushed)
            return []
        return pushed

    def _dispatch(self, pushed):
        """the per-token host work of the push sequences (append, streamer, stopping criteria), in the driving thread"""
        for s, tok in pushed:
            emit = self.sinks.get(s)
            if emit is None or s not in self.active:
                continue
            exc = None
            try:
                stop = emit(tok)
            except BaseException as e:  # noqa: BLE001  (re-raised in the sequence's own thread)
                stop, exc = True, e
            if stop:
                self.active.discard(s)
                self.ready.discard(s)
                del self.sinks[s]
                self.finished[s] = exc
                self.tokq[s].put(_NUDGE)

    def _maybe_step(self):
        """every active sequence waits for a token -> collect the step in flight (launch it first if there
        is none) and immediately launch the next one"""
        if not self.active or self.error is not None or not (self.ready >= self.active):
            return False
        if self.gather_left > 0:
            if time.perf_counter() < self.gather_deadline:
                return False            # more sequences are about to join
            self.gather_left = 0
        if self.inflight is None:
            self._launch()
        pushed = self._collect(dispatch=False)
        self.ready.clear()
        if self.pipeline:
            self._launch()              # speculative, like for the pull sequences: the push sequences' host work below
        self._dispatch(pushed)          # runs under the GPU's next step; one that stops discards its token of that step
        self.ready.update(s for s in self.sinks if s in self.active)     # push sequences never run dry
        return True

    def _run(self, slot: int, emit: Callable[[int], bool]):
        q = self.tokq[slot]
        with self._locked():
            if self.error is not None:
                raise self.error
            self.sinks[slot] = emit
            self.ready.add(slot)
            if self.driver is None:
                self.driver = slot
            else:
                self.tokq[self.driver].put(_NUDGE)      # it may have been waiting for this sequence to get ready
        try:
            while True:
                with (self.cv if self.driver == slot else self._locked()):
                    if slot in self.finished:
                        exc = self.finished.pop(slot)
                        if exc is not None:
                            raise exc
                        return
                    if self.error is not None:
                        raise self.error
                    mine = self.driver == slot
                    stepped = mine and self._maybe_step()
                if stepped:
                    while self.pending > 0:         # joins / leaves go first (they queue on `cv`, which is free now)
                        time.sleep(0.0002)
                    continue
                # not the driver, or the driver with nothing to do right now (warm start still gathering, a pull
                # sequence still busy): sleep until nudged (own completion, driver hand-over, a join, an error)
                try:
                    item = q.get(timeout=0.05 if mine else None)
                except queue.Empty:
                    continue
                if isinstance(item, BaseException):
                    raise item
        finally:
            with self._locked():
                self.sinks.pop(slot, None)
                self.finished.pop(slot, None)
                if self.driver == slot:         # hand the wheel to another push sequence that is still decoding
                    self.driver = next((s for s in self.sinks if s in self.active), None)
                    if self.driver is not None:
                        self.tokq[self.driver].put(_NUDGE)

    def _next_token(self, slot: int) -> int:
        q = self.tokq[slot]
        if q.empty():
            with self._locked():
                if self.error is not None:
                    raise self.error
                if q.empty():
                    self.ready.add(slot)
                    self._maybe_step()      # the last sequence to run dry collects / launches for everybody
        while True:                         # blocks (GIL released) until this slot's token of the next step arrives
            try:
                item = q.get(timeout=None if self.gather_left <= 0 else 0.05)
                if item is _NUDGE:
                    continue
                break
            except queue.Empty:             # warm start only: re-check the gather deadline
                with self._locked():
                    self._maybe_step()
        if isinstance(item, BaseException):
            raise item
        return item


def make_engine(model, processor=None, **kw):
    """The engine of a model's batch slots: the native run loop (infer/engine.NativeBatchEngine; its readers wake once per source
    line, so it is given the processor's newline table) unless DTK_ENGINE=python asks for the Python-driven BatchEngine (A/B runs)."""
    if os.environ.get("DTK_ENGINE", "native") == "python":
        return BatchEngine(model, **kw)
    from .engine import NativeBatchEngine
    flush = None
    if processor is not None:
        from .generate import newline_table
        flush = newline_table(processor).keys()
    return NativeBatchEngine(model, flush_tokens=flush, **kw)


def simulate_parallel(pipeline, image, trees: int, expansions_per_tree: int, seed_base: int = 1000,
                      seeds: Optional[List[int]] = None, resume_in_place: bool = True, slots: Optional[int] = None,
                      **gen_kwargs) -> Iterator[Tuple[float, Any]]:
    """Root-parallel MCTS on one GPU: `trees` independent DetikzifyGenerator searches (tree t draws its sampling seeds
    from a torch generator seeded seeds[t], default seed_base + t) decoded as one batch.  Yields (score, document)
    pairs in completion order.  trees == 1 is the unmodified sequential search.  `slots` < trees: the trees take turns in that
    many decode slots — a tree holds a slot only while it generates, so the others decode while it waits for its reward."""
    for _, score, doc in simulate_parallel_images(pipeline, [image], trees, expansions_per_tree, seed_base, seeds=seeds,
                                                  resume_in_place=resume_in_place, slots=slots, **gen_kwargs):
        yield score, doc


def simulate_parallel_images(pipeline, images, trees_per_image: int, expansions_per_tree: int, seed_base: int = 1000,
                             seeds: Optional[List[int]] = None, resume_in_place: bool = True, slots: Optional[int] = None,
                             **gen_kwargs) -> Iterator[Tuple[int, float, Any]]:
    """Several images in flight on one GPU (BASELINE config 5: 8 images x 4 rollouts): len(images) * trees_per_image
    independent searches decoded as one batch; the engine encodes every image once and forks its KV prefix into the
    slots of that image's trees.  A tree that comes back to a node of its own previous rollout continues in the slot that still
    holds it, without a prefill (BatchEngine resume_in_place; False = fork the image prefix + prefill the path on every join).  Yields (image index, score, document) in completion order.  Tree k (image k //
    trees_per_image) samples with the seed stream seeds[k] (default seed_base + k)."""
    import torch
    imgs = [pipeline.load(im) for im in images]
    with torch.inference_mode():    # what every generator of an image would compute for itself (generate.py: _processed)
        encs = [pipeline.processor(images=im, text=None, text_kwargs={"truncation": True}, return_tensors="pt") for im in imgs]
    trees = len(imgs) * trees_per_image
    if seeds is None:
        seeds = [seed_base + t for t in range(trees)]
    assert len(seeds) == trees, "one seed per tree"
    # more trees than decode slots (the model's, or `slots`): a join waits for a slot to come free (BatchEngine.sequence), i.e. for
    # another tree to finish its rollout and go off to its reward — with rewards that take seconds that keeps the batch full
    engine = make_engine(pipeline.model, processor=pipeline.processor, max_batch=min(trees, slots) if slots else trees, gather=trees,
                         resume_in_place=resume_in_place) if trees > 1 else None
    out: "queue.Queue" = queue.Queue()
    trace_path = os.environ.get("DTK_TRACE_MCTS")
    trace: Optional[List[Tuple[float, int, str]]] = [(time.perf_counter(), -1, "start")] if trace_path else None
    cancelled = threading.Event()
    generators: List[Any] = [None] * trees

    def worker(t: int):
        try:
            gen = torch.Generator().manual_seed(int(seeds[t]))
            draws = iter(lambda: int(torch.randint(0, 2 ** 62, (), generator=gen).item()), None)
            g = generators[t] = pipeline._generator(imgs[t // trees_per_image], None, False, metric=pipeline.metric,
                                                    processed=encs[t // trees_per_image], rng=random.Random(int(seeds[t])),
                                                    **gen_kwargs)      # (tie-breaks of the search from the tree's own stream too)
            base_generate = g.generate

            def generate(input_ids, **kw):      # per-tree RNG stream; a cancelled search starts no further rollout
                if cancelled.is_set():
                    raise InterruptedError("search cancelled")
                return base_generate(input_ids, seed=next(draws), sequence_owner=t, **kw)

            g.generate = generate
            if trace is not None:               # DTK_TRACE_MCTS=<file>: wall-clock of every tree's phases (tools/mcts_timeline.py)
                def timed(f, name):
                    def w(*a, **k):
                        trace.append((time.perf_counter(), t, name + "_start"))
                        try:
                            return f(*a, **k)
                        finally:
                            trace.append((time.perf_counter(), t, name + "_end"))
                    return w
                g.generate, g.score = timed(g.generate, "rollout"), timed(g.score, "reward")
                trace.append((time.perf_counter(), t, "tree_ready"))
            for score, doc in g.simulate(expansions=expansions_per_tree):
                if cancelled.is_set():
                    break
                out.put((t // trees_per_image, score, doc))
        except BaseException as e:
            if not cancelled.is_set():          # errors provoked by the cancellation itself are not results
                out.put(e)
        finally:
            out.put(None)

    threads = [threading.Thread(target=worker, args=(t,), daemon=True) for t in range(trees)]
    for th in threads:
        th.start()
    done = 0
    try:
        while done < trees:
            item = out.get()
            if item is None:
                done += 1
            elif isinstance(item, BaseException):
                raise item
            else:
                yield item
    finally:
        # The consumer may leave early (break / close / an exception of one tree): tell every tree to stop — the running
        # rollouts through their ExplicitAbort (what the reference's rollout() does on GeneratorExit, generate.py:275-277),
        # the searches through the flag — and only then take the engine away: a tree that outlived it would fall back to
        # the single-sequence path of a model other trees are still using.
        if done < trees:
            cancelled.set()
            for g in generators:
                if g is not None:
                    g.control.abort()
        deadline = time.perf_counter() + 120.0
        for th in threads:
            th.join(timeout=max(0.0, deadline - time.perf_counter()))
            while th.is_alive() and time.perf_counter() < deadline:      # a rollout that reset its abort flag after our abort()
                for g in generators:
                    if g is not None:
                        g.control.abort()
                th.join(timeout=0.05)
        alive = [th for th in threads if th.is_alive()]
        if engine is not None and not alive:
            engine.close()
            pipeline.model.last_batch_stats = engine.stats()
        if trace is not None:
            trace.append((time.perf_counter(), -1, "end"))
            import json
            with open(trace_path, "w") as f:
                json.dump(trace, f)
        if alive:
            raise RuntimeError(f"{len(alive)} search threads did not stop within 120 s of the cancellation; "
                               "the batch engine is left attached so they cannot interleave with other sequences")
