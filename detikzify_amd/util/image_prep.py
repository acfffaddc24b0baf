"""
The reward path's image preparation in worker PROCESSES.

Every rollout of a parallel search ends in a SelfSim reward (reference detikzify/evaluate/imagesim.py:91-125): the rendered figure is
trimmed, padded to a square (LANCZOS) and resized by the image processor (BICUBIC) — three Pillow calls of 3-5 ms each that HOLD the
GIL (measured on this image's Pillow 12.2: 8 threads of `Image.resize` take 8 x the time of one).  With fixed-length rollouts the 64
trees of a batch reach their rewards on the same step, so a reward WAVE cost 64 x ~10 ms of serialised Pillow work (0.3-0.5 s with
the decode batch idle: 4 waves = 2 s of BASELINE config 5's 11.5 s, profiles/r03_mcts_timeline.txt).  Here the same functions —
`util.image.expand` and the processor's own resize, same Pillow, bit-identical pixels (tests/test_host_logic.py) — run in a small pool
of spawned processes; the tree's thread waits for its result without the GIL.  What stays in the parent: `load` (a copy for RGB
figures) and the processor's rescale / normalise (a numpy gather that releases the GIL).

`shared_pool()` is the process-wide pool, OPT-IN: DTK_REWARD_PREP_WORKERS=N starts N workers, unset / 0 = off.  Measured: a wave of
64 rewards on an 8-core box 345 -> 227 ms with 6 workers; BASELINE config 5 on the MI355X box's 256-thread host 20.0 / 23.9 rollouts/s
without and 19.9 / 23.7 with 16 workers (profiles/r05_bench_config5_prep_pool_*.json) — that search's wall is decode wait, not
rewards, so the default stays off.  `infer.batching.simulate_parallel_images` attaches the pool to the pipeline's metric when one is
configured and more than one tree runs.
The workers import PIL, numpy and this package's `util` only (50 ms; no torch, no HIP).
"""
from __future__ import annotations

import atexit
import os
import queue
import threading
from multiprocessing import get_context
from typing import Optional, Tuple

import numpy as np
from PIL import Image

from .image import expand


def _expand_and_resize(mode: str, size: Tuple[int, int], data: bytes, w: int, h: int, resample: int) -> Tuple[bytes, Tuple[int, ...]]:
    """ImageSim.get_vision_features' expand(trim) followed by DetikzifyImageProcessor._to_numpy + _resize, on one RGB figure"""
    im = Image.frombytes(mode, size, data)
    ex = expand(im, max(im.size), do_trim=True)
    arr = np.array(ex.convert("RGB") if ex.mode != "RGB" else ex)
    out = np.array(Image.fromarray(arr).resize((w, h), resample=Image.Resampling(resample), reducing_gap=None))
    return out.tobytes(), out.shape


def _worker_main(conn) -> None:
    """one request at a time over the worker's own pipe: a job tuple -> (bytes, shape) | the exception it raised; "ping" -> pid"""
    while True:
        try:
            msg = conn.recv()
        except (EOFError, OSError):
            return
        if msg is None:
            return
        try:
            conn.send(os.getpid() if msg == "ping" else _expand_and_resize(*msg))
        except Exception as e:  # noqa: BLE001  (the parent does the job in-thread instead)
            try:
                conn.send(e)
            except Exception:  # noqa: BLE001
                return


class PrepPool:
    """N spawned worker processes, each behind its own duplex pipe; a calling thread takes an idle worker, sends one job, waits for
    the answer (without the GIL) and puts the worker back.  (Not concurrent.futures: its executor can leave a future unresolved —
    and its manager thread stuck at interpreter exit — when a worker dies while a 0.5 MB job is being queued.)  A worker that dies,
    hangs or answers with an exception costs nothing but the fallback: expand_and_resize returns None and the caller does the
    work in-thread; dead workers are dropped, and a pool without workers says so (`broken`)."""

    def __init__(self, workers: int, timeout: float = 20.0):
        self.workers = int(workers)
        self.timeout = float(timeout)
        self.jobs = 0
        self.broken = False
        self._alive = 0
        self._lock = threading.Lock()
        self._idle: "queue.Queue" = queue.Queue()
        self._all = []
        ctx = get_context("spawn")
        for _ in range(self.workers):
            parent, child = ctx.Pipe()
            proc = ctx.Process(target=_worker_main, args=(child,), daemon=True)
            proc.start()
            child.close()
            self._all.append((proc, parent))
            self._idle.put((proc, parent))
            self._alive += 1

    def _drop(self, proc, conn) -> None:
        try:
            conn.close()
        except OSError:
            pass
        if proc.is_alive():
            proc.kill()
        with self._lock:
            self._alive -= 1
            if self._alive <= 0:
                self.broken = True

    def _call(self, msg):
        """the worker's answer, or None (no worker came free in time / the worker died, hung or raised)"""
        if self.broken:
            return None
        try:
            proc, conn = self._idle.get(timeout=self.timeout)
        except queue.Empty:
            return None
        try:
            conn.send(msg)
            if not conn.poll(self.timeout):
                raise TimeoutError
            out = conn.recv()
        except (EOFError, OSError, TimeoutError):
            self._drop(proc, conn)
            return None
        self._idle.put((proc, conn))
        return None if isinstance(out, BaseException) else out

    def warm(self) -> int:
        """wait until every worker has started (a spawned interpreter + imports, ~0.3 s each, in parallel) instead of paying for it
        under the first reward wave; returns the number of workers that answered"""
        pids = set()
        grabbed = []
        for _ in range(self._alive):
            try:
                grabbed.append(self._idle.get(timeout=self.timeout))
            except queue.Empty:
                break
        for proc, conn in grabbed:
            try:
                conn.send("ping")
                if not conn.poll(max(self.timeout, 60.0)):
                    raise TimeoutError
                pids.add(conn.recv())
                self._idle.put((proc, conn))
            except (EOFError, OSError, TimeoutError):
                self._drop(proc, conn)
        return len(pids)

    def expand_and_resize(self, image: Image.Image, w: int, h: int, resample: int) -> Optional[np.ndarray]:
        """uint8 [h, w, 3] of an RGB figure, or None when no worker could do it (the caller then does the work inline)"""
        if self.broken or image.mode != "RGB":
            return None
        out = self._call((image.mode, image.size, image.tobytes(), int(w), int(h), int(resample)))
        if out is None:
            return None
        with self._lock:
            self.jobs += 1
        data, shape = out
        return np.frombuffer(data, dtype=np.uint8).reshape(shape)

    def close(self):
        self.broken = True
        for proc, conn in self._all:
            try:
                conn.send(None)
            except (OSError, ValueError):
                pass
        for proc, conn in self._all:
            proc.join(timeout=1.0)
            if proc.is_alive():
                proc.kill()
            try:
                conn.close()
            except OSError:
                pass


_SHARED: Optional[PrepPool] = None
_LOCK = threading.Lock()


def default_workers() -> int:
    env = os.environ.get("DTK_REWARD_PREP_WORKERS")
    return max(0, int(env)) if env else 0


def shared_pool() -> Optional[PrepPool]:
    """the process-wide pool, started on first use; None when not configured (DTK_REWARD_PREP_WORKERS unset / 0) or broken"""
    global _SHARED
    with _LOCK:
        if _SHARED is None:
            n = default_workers()
            if n <= 0:
                return None
            _SHARED = PrepPool(n)
            atexit.register(_SHARED.close)
        return None if _SHARED.broken else _SHARED
