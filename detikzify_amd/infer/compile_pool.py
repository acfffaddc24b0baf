"""
Compile worker pool: LaTeX runs of finished rollouts in worker PROCESSES, overlapped with decoding.

The reward of a rollout is dominated by `TikzDocument.compile` — latexmk (seconds), keep-last-page, crop, rasterise
(reference detikzify/infer/tikz.py:89-156).  The reference overlaps that work only in its RL example, with a process pool
and `imap` over a batch of completions (examples/refine.py:151-185: `RewardFunc.compile` returns the rasterised image from
the worker).  Same pattern here, wired into the inference path:

  * `CompilePool(workers)`            N spawned worker processes (spawn, not fork: the parent holds a HIP runtime and
                                      threads); a job = (code, timeout) -> status, log, PNG of the raster, PDF bytes.
  * `pool.imap(codes)`                results in input order, all jobs in flight (the refine.py pattern).
  * `pooled_document_class(pool)`     a TikzDocument type whose compile() runs in the pool: pass it as `document_class` to
                                      DetikzifyPipeline / DetikzifyGenerator.  The trees of `simulate_parallel` (one thread
                                      each) then wait for their LaTeX run in a blocking `Future.result()` — no GIL, no more
                                      than `workers` TeX processes at once — while the other trees' rollouts keep decoding
                                      in the batched step.  `doc.prefetch()` starts a compile without waiting for it.

Within ONE tree nothing can be overlapped without changing the search: expansion k+1 selects on the reward of expansion k
(reference mcts/montecarlo.py:63-66).  The overlap is across trees (root parallelisation) and across the
documents of a batch (`imap`).
"""
from __future__ import annotations

import os
import threading
from concurrent.futures import CancelledError, Future, ProcessPoolExecutor
from concurrent.futures.process import BrokenProcessPool
from io import BytesIO
from multiprocessing import get_context
from typing import Iterable, Iterator, List, NamedTuple, Optional, Type

from PIL import Image

from .tikz import Output, TikzDocument


class CompiledFigure(NamedTuple):
    """what travels back from a worker (picklable)"""
    status: int
    log: str
    png: Optional[bytes]        # raster of the cropped last page, `raster_size` px on the long side (None: nothing to rasterise)
    pdf: Optional[bytes]


def _compile_job(code: str, timeout: Optional[int], raster_size: int, engines: Optional[List[str]],
                 document_class: Type[TikzDocument] = TikzDocument) -> CompiledFigure:
    """runs in a worker process: the unchanged TikzDocument compile + rasterise"""
    if engines is not None:
        TikzDocument.set_engines(engines)
    doc = document_class(code, timeout=timeout)
    out = doc.compile()
    png = pdf = None
    if out.pdf:
        image = doc.rasterize(size=raster_size, expand_to_square=False)
        if image is not None:
            buf = BytesIO()
            image.save(buf, format="PNG")
            png = buf.getvalue()
        try:
            pdf = out.pdf.tobytes()
        except Exception:  # noqa: BLE001  (a toolchain without byte export: the raster is what the reward needs)
            pdf = None
    return CompiledFigure(out.status, out.log, png, pdf)


def _warm_job(seconds: float) -> int:
    import time
    time.sleep(seconds)
    return os.getpid()


class _PooledPdf:
    """stands where TikzDocument keeps its pymupdf document: truthy, serialisable, carries the raster"""

    def __init__(self, fig: CompiledFigure):
        self._fig = fig

    def tobytes(self) -> bytes:
        if self._fig.pdf is None:
            raise ValueError("the worker's toolchain did not export the PDF bytes")
        return self._fig.pdf

    def image(self) -> Image.Image:
        return Image.open(BytesIO(self._fig.png)).convert("RGB")


class CompilePool:
    def __init__(self, workers: Optional[int] = None, raster_size: int = 420, engines: Optional[List[str]] = None,
                 document_class: Type[TikzDocument] = TikzDocument):
        """`document_class`: what a worker compiles with — TikzDocument (latexmk), or a module-level stand-in such as
        SleepingSyntheticTikzDocument (it is pickled by reference: the workers import it)"""
        self.document_class = document_class
        self.workers = workers or max(1, min(16, (os.cpu_count() or 2) // 2))
        # the workers start from a fresh interpreter: an engine list set on the parent's TikzDocument (set_engines) travels
        # with every job unless the caller names another one
        self.raster_size, self.engines = raster_size, list(TikzDocument.engines) if engines is None else engines
        self.restarts = 0
        self._lock = threading.Lock()
        # retries of jobs that were lost with a dying worker run in executors of their own (retry_isolated), several at a time: a
        # 64-tree reward wave that loses a worker loses up to 64 innocent siblings at once, and they must not queue behind one lock
        self.retry_slots = max(1, min(self.workers, 8))
        self._retry_sem = threading.BoundedSemaphore(self.retry_slots)
        self.retries = self.retries_lost = 0
        self.retry_slack_s = 60.0       # on top of the job's own timeout: process spawn, package import, rasterising
        self._isolated_all: List[ProcessPoolExecutor] = []      # the executors of the retries in progress
        self._pool = self._new_executor()

    @property
    def _isolated(self) -> Optional[ProcessPoolExecutor]:
        """the executor of the (latest) retry in progress, None if there is none"""
        with self._lock:
            return self._isolated_all[-1] if self._isolated_all else None

    def _new_executor(self) -> ProcessPoolExecutor:
        return ProcessPoolExecutor(max_workers=self.workers, mp_context=get_context("spawn"))

    def restart(self, broken: ProcessPoolExecutor) -> None:
        """a worker died (OOM-killed TeX run, crashing rasteriser): concurrent.futures marks the whole executor broken for good.
        Replace it once (the first thread that reports `broken` does; the others find a new one already in place)."""
        with self._lock:
            if self._pool is broken:
                self._pool = self._new_executor()
                self.restarts += 1
        broken.shutdown(wait=False, cancel_futures=True)

    def warm(self) -> int:
        """start every worker now (a spawned worker imports this package, ~seconds) instead of under the first rollouts;
        returns the number of distinct worker processes that answered"""
        return len({f.result() for f in [self._pool.submit(_warm_job, 0.3) for _ in range(self.workers)]})

    def submit(self, code: str, timeout: Optional[int] = 60) -> "Future[CompiledFigure]":
        for _ in range(3):
            pool = self._pool
            try:
                f = pool.submit(_compile_job, code, timeout, self.raster_size, self.engines, self.document_class)
                f._dtk_executor = pool       # so that whoever sees BrokenProcessPool on this future restarts the right executor
                return f
            except BrokenProcessPool:
                self.restart(pool)
        raise BrokenProcessPool("the compile pool's workers keep dying")

    def result(self, future: "Future[CompiledFigure]") -> Optional[CompiledFigure]:
        """the figure, or None when the job's worker (or a sibling: the executor fails every pending job) died — the pool is
        restarted for the jobs that follow; the lost job is reported as a failed compile by the caller"""
        try:
            return future.result()
        except (BrokenProcessPool, CancelledError):     # CancelledError: restart() shut the broken executor down with cancel_futures=True while this job was still queued
            self.restart(getattr(future, "_dtk_executor", self._pool))
            return None

    def retry_isolated(self, code: str, timeout: Optional[int] = 60) -> Optional[CompiledFigure]:
        """second attempt of a job that was lost with a dying worker, in an executor of its OWN (one worker, one job): the document
        that killed the worker kills only this one, so the innocent siblings that retry next to it keep their second attempt (a
        retry on the shared, restarted pool let the poison document break it again: ADVICE r4).  Up to `retry_slots` such executors
        run side by side (one lock made a lost wave of 64 retry strictly one at a time — a process spawn, the package
        import and a full TeX run each — while the decode batch sat idle).  The wait is bounded: the job's own timeout plus the
        time a fresh worker needs to start; a worker that hangs beyond it is terminated.  None = lost again (or hung)."""
        from concurrent.futures import TimeoutError as FutureTimeout
        with self._retry_sem:
            ex = ProcessPoolExecutor(max_workers=1, mp_context=get_context("spawn"))
            with self._lock:
                self.retries += 1
                self._isolated_all.append(ex)
            try:
                limit = None if timeout is None else float(timeout) + self.retry_slack_s
                return ex.submit(_compile_job, code, timeout, self.raster_size, self.engines, self.document_class).result(timeout=limit)
            except (BrokenProcessPool, CancelledError, FutureTimeout):
                with self._lock:
                    self.retries_lost += 1
                return None
            finally:
                with self._lock:
                    self._isolated_all.remove(ex)
                procs = list((getattr(ex, "_processes", None) or {}).values())     # a hung TeX run outlives shutdown(wait=False): end it
                ex.shutdown(wait=False, cancel_futures=True)
                for proc in procs:
                    if proc.is_alive():
                        proc.terminate()

    def imap(self, codes: Iterable[str], timeout: Optional[int] = 60) -> Iterator[CompiledFigure]:
        """all documents in flight at once, results in input order (multiprocessing.Pool.imap, refine.py:176)"""
        codes = list(codes)
        futures = [self.submit(code, timeout) for code in codes]
        for code, f in zip(codes, futures):
            fig = self.result(f)
            if fig is None:         # lost with a dying sibling: once more, in isolation
                fig = self.retry_isolated(code, timeout)
            yield fig if fig is not None else CompiledFigure(-1, "compile worker died", None, None)

    def close(self):
        self._pool.shutdown(wait=True, cancel_futures=True)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def pooled_document_class(pool: CompilePool) -> Type[TikzDocument]:
    """TikzDocument whose compile runs in `pool` (same interface, same status / log / errors / raster)"""

    class PooledTikzDocument(TikzDocument):
        _future: Optional[Future] = None

        def prefetch(self) -> "PooledTikzDocument":
            if self._compiled is None and self._future is None:
                self._future = pool.submit(self.code, self.timeout)
            return self

        def _compile(self) -> Output:
            fig = pool.result(self.prefetch()._future)
            if fig is None:
                # a worker died and concurrent.futures failed EVERY pending job of that executor — most of them innocent siblings
                # whose document would otherwise be cached as a failed compile (reward -1 for good: ADVICE r3).  The pool has been
                # restarted for the jobs that follow; THIS document runs once more in an executor of its own (a poison document then
                # takes nobody with it); only a second loss is reported as a failed compile.
                fig = pool.retry_isolated(self.code, self.timeout)
            if fig is None:         # the worker died under this very job twice: a failed compile, not a failed search
                return Output()
            return Output(pdf=_PooledPdf(fig) if fig.png is not None else None, status=fig.status, log=fig.log)

        def rasterize(self, size: int = 420, expand_to_square: bool = True, **_) -> Optional[Image.Image]:
            pdf = self.pdf
            if not pdf:
                return None
            from ..util import expand
            if size == pool.raster_size:
                image = pdf.image()             # the worker's raster
            elif pdf._fig.pdf is not None:      # another size: rasterise the PDF again, as the base class would (never a resample)
                image = self.toolchain.to_image(pdf, size)
            else:                               # a toolchain without byte export: the worker's raster is all there is
                image = pdf.image()
                scale = size / max(image.size)
                image = image.resize((max(1, round(image.width * scale)), max(1, round(image.height * scale))), Image.LANCZOS)
            return expand(image, size) if expand_to_square else image

    return PooledTikzDocument
