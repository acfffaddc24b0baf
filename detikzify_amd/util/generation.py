"""
Thread/stream plumbing the generate loop drives (same contracts as reference
detikzify/util/generation.py:7-101).  Written against duck-typed protocols so neither
transformers' StoppingCriteria nor its BaseStreamer is needed on the hot path:
  stopping criterion: callable(input_ids, scores) -> bool          (ExplicitAbort :7-23)
  streamer:           .put(tensor) per token, .end() once           (TokenStreamer :25-66)
Errors raised in the generating thread travel through the queue and re-raise in the consumer
(:53, :59-66) — the rollout contract of infer/generate.py:248-258.
"""
from __future__ import annotations

from queue import Queue
from typing import Optional


class ExplicitAbort:
    """Cooperative cancel, polled once per generated token."""

    def __init__(self):
        self.should_stop = False

    def __call__(self, input_ids, scores, **kwargs) -> bool:
        return self.should_stop

    def reset(self):
        self.should_stop = False
        return self

    def abort(self):
        self.should_stop = True


class _QueueStreamer:
    _END = None

    def __init__(self, timeout: Optional[float] = None):
        self.queue: Queue = Queue()
        self.timeout = timeout

    def end(self):
        self.queue.put(self._END, timeout=self.timeout)

    def propagate_error(self, exc: BaseException):
        self.queue.put(exc, timeout=self.timeout)

    def __iter__(self):
        return self

    def __next__(self):
        item = self.queue.get(timeout=self.timeout)
        if isinstance(item, BaseException):
            raise item
        if item is self._END:
            raise StopIteration()
        return item


class TokenStreamer(_QueueStreamer):
    """Streams raw token ids; the first put() (the prompt) is dropped when skip_prompt.

    `flush_on` (optional, a container of token ids): tokens are handed to the consumer in bursts that end with one of
    these ids (and at end()).  The consumer iterates over exactly the same ids in the same order; it is merely woken
    once per burst instead of once per token.  DetikzifyGenerator.rollout acts on newline tokens only and passes its
    newline table: with dozens of rollouts decoding in one batch, a thread hand-off per token per rollout is what the
    GIL cannot keep up with (one per source line it can)."""

    def __init__(self, skip_prompt: bool = True, timeout: Optional[float] = None, flush_on=None):
        super().__init__(timeout)
        self.skip_prompt = skip_prompt
        self.next_tokens_are_prompt = True
        self.token_queue = self.queue  # reference attribute name
        self.flush_on = flush_on
        self._out: list = []        # producer side: tokens of the burst being built
        self._in: list = []         # consumer side: the burst being handed out (reversed)

    @property
    def per_token(self) -> bool:
        """without flush_on the consumer is woken per token and expects each one as it is made (no bursts from the engine either)"""
        return self.flush_on is None

    def put(self, value):
        if len(value.shape) > 1:
            if value.shape[0] > 1:
                raise ValueError("TokenStreamer only supports batch size 1")
            value = value[0]
        if self.skip_prompt and self.next_tokens_are_prompt:
            self.next_tokens_are_prompt = False
            return
        for token_id in value.tolist():
            self.put_token(token_id)

    def put_token(self, token_id: int):
        """one generated token as a plain int (the generate loop's fast path: no tensor per token)"""
        if self.flush_on is None:
            self.queue.put(token_id, timeout=self.timeout)
            return
        self._out.append(token_id)
        if token_id in self.flush_on:
            self._flush()

    def put_tokens(self, token_ids):
        """several generated tokens at once, in order (the engine's multi-step runs hand a sequence the tokens of a whole run):
        the consumer sees exactly what one put_token per id would have produced"""
        if self.flush_on is None:
            for token_id in token_ids:
                self.queue.put(token_id, timeout=self.timeout)
            return
        start, flush_on = 0, self.flush_on
        for i, token_id in enumerate(token_ids):
            if token_id in flush_on:
                self._out.extend(token_ids[start:i + 1])
                self._flush()
                start = i + 1
        self._out.extend(token_ids[start:])

    def _flush(self):
        if self._out:
            burst, self._out = self._out, []
            self.queue.put(burst, timeout=self.timeout)

    def propagate_error(self, exc: BaseException):
        self._flush()
        super().propagate_error(exc)

    def end(self):
        self.next_tokens_are_prompt = True
        self._flush()
        super().end()

    def __next__(self):
        if self._in:
            return self._in.pop()
        item = super().__next__()
        if isinstance(item, list):
            item.reverse()
            self._in = item
            return self._in.pop()
        return item


class TextIteratorStreamer(_QueueStreamer):
    """Streams decoded text (used by the web UI caller of the boundary, webui/webui.py:36-55).
    Emits text whenever the decoded suffix ends in whitespace/newline or at end()."""

    def __init__(self, tokenizer, skip_prompt: bool = True, timeout: Optional[float] = None, **decode_kwargs):
        super().__init__(timeout)
        self.tokenizer, self.skip_prompt, self.decode_kwargs = tokenizer, skip_prompt, decode_kwargs
        self.next_tokens_are_prompt = True
        self._cache, self._printed = [], 0
        self.text_queue = self.queue

    def put(self, value):
        if len(value.shape) > 1:
            if value.shape[0] > 1:
                raise ValueError("TextIteratorStreamer only supports batch size 1")
            value = value[0]
        if self.skip_prompt and self.next_tokens_are_prompt:
            self.next_tokens_are_prompt = False
            return
        self._cache.extend(value.tolist())
        text = self.tokenizer.decode(self._cache, **self.decode_kwargs)
        if text.endswith("\n"):
            out, self._cache, self._printed = text[self._printed:], [], 0
        else:
            cut = text.rfind(" ") + 1
            out, self._printed = text[self._printed:cut], max(cut, self._printed)
        if out:
            self.queue.put(out, timeout=self.timeout)

    def end(self):
        if self._cache:
            text = self.tokenizer.decode(self._cache, **self.decode_kwargs)
            if text[self._printed:]:
                self.queue.put(text[self._printed:], timeout=self.timeout)
        self._cache, self._printed, self.next_tokens_are_prompt = [], 0, True
        super().end()


class StreamerList(list):
    """Fan one generate() out to several streamers (reference :81-91)."""

    def put(self, value):
        for s in self:
            s.put(value)

    def put_token(self, token_id: int):
        """one generated token as a plain int (the generate loop's fast path); members without put_token get the HF protocol's
        one-element tensor"""
        for s in self:
            f = getattr(s, "put_token", None)
            if f is not None:
                f(token_id)
            else:
                import torch
                s.put(torch.tensor([token_id], dtype=torch.int64))

    def put_tokens(self, token_ids):
        """a burst of generated tokens, in order: exactly what one put_token per id would have produced"""
        for s in self:
            f = getattr(s, "put_tokens", None)
            if f is not None:
                f(token_ids)
            else:
                for token_id in token_ids:
                    StreamerList.put_token([s], token_id)

    @property
    def per_token(self) -> bool:
        """True if some member needs the per-token protocol (no put_tokens): bursts would delay what it shows"""
        return any(bool(getattr(s, "per_token", getattr(s, "put_tokens", None) is None)) for s in self)

    def end(self):
        for s in self:
            s.end()


def unwrap_processor(processor):
    """Adapter processors nest the real one under `.processor` (reference :93-101)."""
    while hasattr(processor, "processor"):
        processor = processor.processor
    return processor
