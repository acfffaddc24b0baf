"""
NativeBatchEngine — the batch of rollouts of one GPU, driven by the native run loop of libdtk_hip.so (include/dtk.h,
"dtk_engine_*"; csrc/dtk_engine.cpp).

Same contract as infer/batching.BatchEngine (the Python-driven engine it replaces as the default): every rollout is a thread
inside `model.generate` (reference detikzify/infer/generate.py:246-282 runs it in a worker thread and consumes the stream line by
line), and all of them share one pass over the weights per decode step.  What changed is who turns the crank: a native thread
launches and collects the steps (two in flight), keeps every slot's tokens in a ring and applies the sequence's stop rules (EOS,
max_length); the rollout threads block in dtk_engine_read — outside the interpreter lock — and wake once per source line, not once
per token.  Python keeps what is policy: which slot a sequence gets, which slot still holds which image prefix (fork / in-place
reuse / prefix-cache LRU), and what each join therefore has to do; the native loop executes the joins in the order they were
planned.  A slot's arithmetic does not depend on its company, so a sequence is the same tokens as under BatchEngine (tests).
"""
from __future__ import annotations

import ctypes as C
import threading
from collections import OrderedDict
from contextlib import contextmanager
from typing import Any, Callable, Dict, Iterable, Iterator, List, Optional, Tuple

import torch

from .. import _lib


class _NativeSequence:
    def __init__(self, engine: "NativeBatchEngine", slot: int):
        self.engine, self.slot = engine, slot
        self._buf = (C.c_int64 * 256)()
        self._n, self._state = C.c_int32(0), C.c_int32(0)
        self._pending: List[int] = []
        self.ended = False

    def _read(self, cap: int, timeout_ms: int) -> List[int]:
        e = self.engine
        rc = e.lib.dtk_engine_read(e._h, self.slot, self._buf, cap, C.byref(self._n), C.byref(self._state), timeout_ms)
        if rc != 0:
            raise _lib.DtkError(f"the batch engine's device failed ({rc}): {e.lib.dtk_engine_last_error(e._h).decode(errors='replace')}")
        n = self._n.value
        if self._state.value != _lib.DTK_SEQ_RUNNING:
            self.ended = True
        return self._buf[:n] if n else []

    def run(self, emit: Callable[[int], bool]):
        """`emit.many(tokens) -> stop` (or emit(token) per token) for every burst of this sequence's tokens, in this thread; returns
        when emit asked to stop, or the sequence ended by its own rules (stop id, token budget)."""
        many = getattr(emit, "many", None)
        aborted = getattr(emit, "aborted", None)
        while True:
            toks = self._read(256, 100)
            if toks:
                if many is not None:
                    if many(toks):
                        return
                else:
                    for tok in toks:
                        if emit(tok):
                            return
            elif aborted is not None and aborted():       # nothing new for 100 ms (a join's prefill, a paused batch): poll the flag
                return
            if self.ended:
                return

    def next_token(self) -> int:
        """pull: the next token of this sequence (blocks); a sequence that has ended raises StopIteration"""
        while True:
            if self._pending:
                return self._pending.pop(0)
            if self.ended:
                raise StopIteration
            self._pending = list(self._read(256, -1))


class _ScriptedOps:
    """dtk_engine_ops over a Python model object (the CPU tests' scripted device): the native loop calls back into Python for every
    device operation.  Exceptions become error codes; an AssertionError (a scripted device's protocol check) stops the engine."""

    def __init__(self, model, engine: "NativeBatchEngine"):
        self.model, self.engine = model, engine
        self._err = C.create_string_buffer(512)     # the text last_error hands to the native side (owned here)
        O = _lib.DtkEngineOps

        def guard(f):
            def g(*a):
                try:
                    r = f(*a)
                    return 0 if r is None else r
                except AssertionError as e:
                    self._err.value = f"scripted device: {e!r}".encode()[:511]
                    return -2
                except BaseException as e:  # noqa: BLE001
                    self._err.value = str(e).encode()[:511]
                    return -1
            return g

        def ids_of(ptr, n):
            return torch.tensor([ptr[i] for i in range(n)], dtype=torch.int64)

        def launch(dev, active):
            model.decode_batch_launch([j for j in range(_lib.DTK_MAX_BATCH) if active[j]])

        def wait(dev, out):
            toks = model.decode_batch_wait()
            for j in range(_lib.DTK_MAX_BATCH):
                out[j] = toks[j]

        def prefill(dev, slot, ids, T, pixels, key, flags):
            px = engine._pixels_by_key.get(int(key)) if pixels else None
            model.prefill(ids_of(ids, T), px, slot=slot, reuse=bool(flags & _lib.DTK_PREFILL_REUSE_PREFIX))

        def sampling(dev, slot, sp):
            s = sp.contents
            model.set_sampling(slot=slot, do_sample=bool(s.do_sample), temperature=s.temperature, top_p=s.top_p, top_k=s.top_k, seed=s.seed,
                               bad_ids=list(s.bad_ids[:s.n_bad]), begin_suppress_ids=list(s.begin_suppress_ids[:s.n_begin_suppress]),
                               always_suppress_ids=list(s.always_suppress_ids[:s.n_always_suppress]))

        def fork(dev, a, b, n):
            model.kv_fork(a, b, n)

        def lcp(dev, slot, ids, n, key, out):
            best = model.best_lcp_slot([slot], ids_of(ids, n), int(key))
            out[0] = best[1] if best else 0

        def resume(dev, slot, ids, n, key):
            model.resume_slot(slot, ids_of(ids, n), int(key))

        def ctxlen(dev, slot):
            return int(model.lib.dtk_context_len_slot(model._ctx, slot))

        def lasterr(dev):
            return C.addressof(self._err)

        self.keep = [O.LAUNCH(guard(launch)), O.WAIT(guard(wait)), O.PREFILL(guard(prefill)), O.SAMPLING(guard(sampling)), O.FORK(guard(fork)),
                     O.LCP(guard(lcp)), O.RESUME(guard(resume)), O.CTXLEN(ctxlen), O.LASTERR(lasterr)]
        self.ops = O(None, *self.keep, int(model.config.max_positions), int(engine.decode_slots))


class NativeBatchEngine:
    """One engine per model; every method is thread-safe.  `sequence()` is the per-rollout entry (model.generate uses it when
    model.batch_engine is set): plan the join under the engine's lock, queue it natively in plan order, wait for it outside the lock."""
    native = True

    def __init__(self, model, max_batch: Optional[int] = None, share_prefix: bool = True, pipeline: bool = True, gather: int = 0,
                 gather_timeout: float = 0.5, prefix_slots: Optional[int] = None, resume_in_place: bool = True,
                 flush_tokens: Optional[Iterable[int]] = None, flush_max: int = 32):
        n = model.num_slots()
        if n <= 0:
            raise ValueError("model was loaded without batch slots (load(..., batch_slots=N))")
        self.model = model
        self.lib = _lib.load_library()
        probe = getattr(model, "max_decode_slots", None)
        dec = int(probe()) if callable(probe) else min(n, 64 if n > 33 else (32 if n > 17 else 16))
        self.decode_slots = dec
        self.share_prefix = share_prefix and n >= 2
        self.capacity = min(dec, max_batch) if max_batch else (dec if n > dec else dec - 1)
        spare = n - self.capacity       # slots the decode batch does not need hold image prefixes (see BatchEngine)
        n_prefix = min(spare, 8 if prefix_slots is None else int(prefix_slots)) if self.share_prefix else 0
        self.prefix_slots: List[int] = list(range(n - n_prefix, n))
        self.prefix_cache: "OrderedDict[Tuple[int, int], int]" = OrderedDict()
        self.slot_img: Dict[int, Tuple[int, int]] = {}
        self.resume_in_place = resume_in_place
        self.last_slot: Dict[int, int] = {}
        self.joins = self.resumes = self.inplace_reuses = self.prefix_encodes = 0
        self.flush_max = int(flush_max)
        self._cv = threading.Condition()        # protects the bookkeeping above and `free`; joins are queued natively under it
        self.free: List[int] = list(range(self.capacity))
        self._pixels_by_key: Dict[int, Any] = {}
        self._h = C.c_void_p()
        ctx = getattr(model, "_ctx", None)
        if ctx:
            rc = self.lib.dtk_engine_create(ctx, C.byref(self._h))
            self._scripted = None
        else:
            self._scripted = _ScriptedOps(model, self)
            rc = self.lib.dtk_engine_create_ops(C.byref(self._scripted.ops), C.byref(self._h))
        if rc != 0:
            raise _lib.DtkError(f"dtk_engine_create failed ({rc})")
        if not pipeline:
            self.lib.dtk_engine_set_option(self._h, b"depth", 1)
        self.has_flush_tokens = False
        if flush_tokens is not None:
            self.set_flush_tokens(flush_tokens)
        if gather:
            self.expect(gather, gather_timeout)
        model.batch_engine = self

    # ---- configuration -------------------------------------------------------------------------------------------------------
    def set_flush_tokens(self, token_ids: Iterable[int]):
        ids = sorted({int(t) for t in token_ids})
        arr = (C.c_int64 * max(1, len(ids)))(*ids)
        if self.lib.dtk_engine_set_flush_tokens(self._h, arr, len(ids)) != 0:
            raise _lib.DtkError("dtk_engine_set_flush_tokens failed")
        self.has_flush_tokens = bool(ids)

    def expect(self, n: int, timeout: float = 0.5):
        self.lib.dtk_engine_expect(self._h, int(n), int(timeout * 1000))

    def native_stats(self) -> Dict[str, Any]:
        st = _lib.DtkEngineStats()
        self.lib.dtk_engine_get_stats(self._h, C.byref(st))
        return {k: getattr(st, k) for k, _ in st._fields_}

    def stats(self) -> Dict[str, Any]:
        st = self.native_stats() if self._h else self._final
        return {"engine": "native", "steps": st["steps"], "tokens_out": st["tokens_out"], "wait_s": round(st["wait_s"], 3),
                "launch_s": round(st["launch_s"], 3), "prefill_s": round(st["join_s"], 3), "drain_s": round(st["drain_s"], 3),
                "host_bound_steps": st["host_bound_steps"], "prefix_encodes": self.prefix_encodes, "inplace_reuses": self.inplace_reuses,
                "joins": self.joins, "resumed_in_place": self.resumes, "idle_between_steps_s": round(st["idle_s"], 3),
                "steps_below_half_occupancy": st["steps_below_half_occupancy"], "reader_wakeups": st["reader_wakeups"],
                "wasted_slot_steps": st["wasted_slot_steps"],
                "span_s": round(st["last_collect_t"] - st["first_launch_t"], 3) if st["first_launch_t"] else 0.0}

    def close(self):
        if self._h:
            self._final = self.native_stats()
            h, self._h = self._h, C.c_void_p()
            self.lib.dtk_engine_destroy(h)
        if getattr(self.model, "batch_engine", None) is self:
            self.model.batch_engine = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def prefix_slot(self) -> Optional[int]:
        return self.prefix_slots[-1] if self.prefix_slots else None

    @property
    def steps(self) -> int:
        return int((self.native_stats() if self._h else self._final)["steps"])

    # ---- joins ---------------------------------------------------------------------------------------------------------------
    def _prefix_key(self, ids, pixel_values):
        tok = self.model.config.image_token_id
        ids = ids.reshape(-1)
        n_img = int((ids == tok).sum())
        if n_img == 0 or not bool((ids[:n_img] == tok).all()):
            return None
        return (self.model.image_key(pixel_values), n_img)

    def _plan_prefix(self, j: "_lib.DtkJoin", slot: int, key) -> Callable[[], None]:
        """what a join of an image prompt into `slot` has to do to get the image prefix there (BatchEngine._fork_prefix, as a plan):
        fills j's prefix fields, returns the bookkeeping to apply once the plan is certain to run.  Sources, in order: a prefix-cache
        slot that holds this image (its fork also carries the logits); the slot itself (in place); any other slot that still holds
        the prefix (donor); else the image is encoded into a free — else the least recently used — prefix-cache slot."""
        n_img = key[1]
        j.prefix_len = n_img
        src = self.prefix_cache.get(key)
        if src is None and self.slot_img.get(slot) == key:
            j.prefix_in_place = 1
            return lambda: None
        if src is not None:
            j.prefix_src, j.prefix_src_whole = src, 1

            def apply_cached():
                self.slot_img[slot] = key
                self.prefix_cache.move_to_end(key)
            return apply_cached
        donor = next((s for s, k in self.slot_img.items() if k == key and s != slot), None)
        if donor is not None:
            j.prefix_src, j.prefix_src_whole = donor, 0
            return lambda: self.slot_img.__setitem__(slot, key)
        if not self.prefix_slots:       # nobody holds this image and there is no prefix-cache slot: full prefill, the slot becomes a donor
            j.prefix_len = 0

            def apply_full():
                self.slot_img[slot] = key
                self.prefix_encodes += 1
            return apply_full
        used = set(self.prefix_cache.values())
        src = next((s for s in reversed(self.prefix_slots) if s not in used), None)
        evict = None
        if src is None:
            evict, src = next(iter(self.prefix_cache.items()))      # least recently used image
        j.prefix_src, j.prefix_src_whole, j.prefix_encode = src, 1, 1

        def apply_encode():
            if evict is not None:
                self.prefix_cache.pop(evict, None)
            self.prefix_cache[key] = src
            self.prefix_cache.move_to_end(key)
            self.slot_img[slot] = key
            self.prefix_encodes += 1
        return apply_encode

    @contextmanager
    def sequence(self, ids, pixel_values, sampling: Dict[str, Any], owner: Optional[int] = None, max_new_tokens: Optional[int] = None,
                 stop_ids: Iterable[int] = (), per_token: bool = True) -> Iterator[_NativeSequence]:
        """`owner`: see BatchEngine.sequence.  `max_new_tokens` / `stop_ids`: the sequence's own end (the native loop stops it there);
        `per_token`: the reader is woken for every token (arbitrary stopping criteria / foreign streamers) instead of per line."""
        if not self._h:
            raise _lib.DtkError("the batch engine is closed")
        ids = ids.detach().to("cpu", torch.int64).reshape(-1).contiguous()
        n_ids = int(ids.numel())
        want = self._prefix_key(ids, pixel_values) if (self.share_prefix and pixel_values is not None) else None
        px = None
        if pixel_values is not None:
            px = pixel_values.detach().to("cpu", torch.float32).contiguous()
            if px.dim() == 4:
                if px.shape[0] != 1:
                    raise ValueError("batch size 1 only")
                px = px[0]
        j = _lib.DtkJoin()
        j.n_ids, j.ids = n_ids, ids.data_ptr()
        j.pixels = px.data_ptr() if px is not None else None
        j.image_key = self.model.image_key(pixel_values) if pixel_values is not None else 0
        j.prefix_src = -1
        j.full_flags = (_lib.DTK_PREFILL_REUSE_PREFIX | _lib.DTK_PREFILL_REUSE_IMAGE) if getattr(self.model, "reuse_prefix", False) else 0
        s = j.sampling
        s.do_sample, s.temperature = int(bool(sampling.get("do_sample", False))), float(sampling.get("temperature", 1.0))
        s.top_p, s.top_k = float(sampling.get("top_p", 1.0)), int(sampling.get("top_k", 0) or 0)
        s.seed = int(sampling.get("seed", 0)) & ((1 << 64) - 1)
        for field, cnt, name in (("bad_ids", "n_bad", "bad_ids"), ("begin_suppress_ids", "n_begin_suppress", "begin_suppress_ids"),
                                 ("always_suppress_ids", "n_always_suppress", "always_suppress_ids")):
            vals = [int(v) for v in (sampling.get(name) or ())]
            if len(vals) > 8:
                raise ValueError("at most 8 ids per suppression list")
            setattr(s, cnt, len(vals))
            arr = getattr(s, field)
            for i, v in enumerate(vals):
                arr[i] = v
        stops = [int(t) for t in stop_ids if int(t) >= 0][:8]
        j.n_stop = len(stops)
        for i, t in enumerate(stops):
            j.stop_ids[i] = t
        j.max_new_tokens = int(max_new_tokens) if max_new_tokens is not None else (1 << 30)
        j.flush_mode = 0 if per_token else 1
        j.flush_max = self.flush_max
        if self._scripted is not None and pixel_values is not None:
            self._pixels_by_key[int(j.image_key)] = pixel_values

        ticket = C.c_uint64(0)
        held_through = False
        with self._cv:
            while not self.free:
                self._cv.wait()
            order = sorted(self.free)
            slot, cands = None, []
            # resume in place: the prompt (all but its last token) is still in a free slot's cache — an MCTS tree coming back to a
            # node of its own previous rollout.  Whether that holds is decided by the native loop when the join executes
            # (dtk_slot_lcp behind every step in flight); the plan here only names where to look.  An owner looks in the slot IT used
            # last (so that resume-or-prefill never depends on which other slots happen to be free: BatchEngine.sequence).
            may_resume = (self.resume_in_place and n_ids >= 2 and (want is not None or pixel_values is None)
                          and int(ids[-1]) != self.model.config.image_token_id)
            holds = (lambda f: self.slot_img.get(f) == want) if want is not None else (lambda f: f not in self.slot_img)
            if may_resume and owner is not None:
                last = self.last_slot.get(owner)
                if last in self.free and holds(last):
                    slot, j.try_resume = last, 1
            elif may_resume:
                cands = [f for f in order if holds(f)]
            if slot is None and owner is not None and self.last_slot.get(owner) in self.free:
                slot = self.last_slot[owner]
            if slot is None:
                slot = next((f for f in order if f not in self.slot_img), None)
            if slot is None and want is not None:
                slot = next((f for f in order if self.slot_img.get(f) == want), None)
            if slot is None:
                slot = order[0]
            j.slot = slot
            if cands:
                j.try_resume, j.n_candidates = 1, len(cands)
                for i, f in enumerate(cands):
                    j.candidates[i] = f
            apply = self._plan_prefix(j, slot, want) if want is not None else (lambda: self.slot_img.pop(slot, None))
            self.free.remove(slot)
            if owner is not None:
                self.last_slot[owner] = slot
            rc = self.lib.dtk_engine_submit(self._h, C.byref(j), C.byref(ticket))
            if rc != 0:
                self.free.append(slot)
                raise _lib.DtkError(f"dtk_engine_submit failed ({rc}): {j.error_out.decode(errors='replace')}")
            if cands:
                # owner-less resume candidates: the slot is only known once the join has run, so the bookkeeping waits for it and
                # no other join is planned meanwhile (rare path: MCTS trees always carry an owner)
                held_through = True
                rc = self.lib.dtk_engine_await(self._h, ticket.value)
                if rc == 0 and j.how_out == _lib.DTK_JOIN_RESUMED:
                    if j.slot_out != slot:
                        self.free.append(slot)
                        self.free.remove(j.slot_out)
                        slot = j.slot_out
                elif rc == 0:
                    apply()
            else:
                # (an owner's try_resume join plans a fork from the cache or an in-place reuse — `holds(last)` — and both leave the
                # bookkeeping as it is if the join resumes instead: their `apply` only re-states slot_img[slot] == want)
                apply()
        if not held_through:
            rc = self.lib.dtk_engine_await(self._h, ticket.value)
        joined = rc == 0
        try:
            if not joined:
                text = j.error_out.decode(errors="replace")
                if "image patch tokens" in text:
                    raise ValueError(text[text.index("The "):] if "The " in text else text)      # the reference's own ValueErrors
                raise _lib.DtkError(f"join of a sequence failed ({rc}): {text}")
            with self._cv:
                self.joins += 1
                if j.how_out == _lib.DTK_JOIN_RESUMED:
                    self.resumes += 1
                elif j.how_out == _lib.DTK_JOIN_IN_PLACE:
                    self.inplace_reuses += 1
            yield _NativeSequence(self, slot)
        finally:
            if joined and self._h:
                self.lib.dtk_engine_leave(self._h, slot)
            with self._cv:
                if not joined:
                    self.slot_img.pop(slot, None)       # the prompt did not get there: the slot holds nobody's prefix
                    if j.prefix_encode and want is not None and self.prefix_cache.get(want) == j.prefix_src:
                        self.prefix_cache.pop(want, None)      # ... and neither does the prefix-cache slot it was to be encoded into
                self.free.append(slot)
                self._cv.notify_all()
            del ids, px     # (kept alive until here: the native join read them)
