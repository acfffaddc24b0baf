"""The native run loop of a batch (include/dtk.h "dtk_engine_*", csrc/dtk_engine.cpp, infer/engine.NativeBatchEngine) on the CPU: the
loop itself is the shipped native code; the device under it is the scripted device of tests/test_generate_loop.py, reached through
dtk_engine_create_ops (every device operation of the loop calls back into that Python object).  What this pins without a GPU:

  * a sequence decoded through the native loop is, token for token, the sequence decoded alone and the sequence decoded through the
    Python-driven BatchEngine — with threads queueing for slots, prefix-cache forks, donors, in-place reuse, resumes in place;
  * the loop ends a sequence itself at its stop id / token budget and never launches a step past the budget;
  * readers are woken per source line (flush tokens), per flush_max tokens, or per token when a caller-supplied object needs every
    token (stopping criteria, foreign streamers) — whose exact HF semantics are kept;
  * failures: a device error reaches every reader and every later join; a failing streamer / criterion frees its slot only;
  * joins execute in plan order (an image's encode into the prefix cache always precedes the forks planned after it).
"""
import ctypes as C
import threading
import time

import pytest
import torch

from detikzify_amd import _lib
from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
from detikzify_amd.infer.batching import BatchEngine, simulate_parallel
from detikzify_amd.infer.engine import NativeBatchEngine
from detikzify_amd.util import ExplicitAbort, TokenStreamer

from .helpers import fake_processor, sketch_image
from .test_generate_loop import EOS, IMG, NIMG, VOCAB, ScriptedDevice, _prompt, _Recorder


def _threads(n, target):
    errs = []

    def guarded(k):
        try:
            target(k)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=guarded, args=(k,)) for k in range(n)]
    [t.start() for t in ths]
    [t.join(timeout=90) for t in ths]
    assert not any(t.is_alive() for t in ths), "threads hang"
    assert not errs, errs[:1]


def test_native_sequences_decode_exactly_as_alone_and_as_the_python_engine():
    proc = fake_processor(VOCAB, NIMG)
    jobs = []
    for j in range(18):
        ids, px = _prompt(proc, j % 3, extra=[40 + j, 50 + j][: j % 3])
        jobs.append((ids, px, 100 + j))
    kw = dict(bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS], do_sample=True, max_length=NIMG + 60)
    alone_dev = ScriptedDevice()
    alone = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw) for i, p, s in jobs]

    results = {}
    for make in (NativeBatchEngine, BatchEngine):
        dev = ScriptedDevice(slots=5)
        eng = make(dev, max_batch=4)
        got = [None] * len(jobs)

        def worker(k):
            for j in range(k, len(jobs), 6):
                i, p, s = jobs[j]
                got[j] = dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)
        _threads(6, worker)                       # more threads than slots
        st = eng.stats()
        eng.close()
        assert dev.batch_engine is None and not dev.bpending
        for a, g in zip(alone, got):
            assert torch.equal(a, g)
        assert st["joins"] == 18 and dev.forks + st["inplace_reuses"] + st["resumed_in_place"] == 18
        assert dev.prefills == st["prefix_encodes"] + dev.tail_prefills and st["prefix_encodes"] <= 18
        results[make.__name__] = st
    nat = results["NativeBatchEngine"]
    assert nat["engine"] == "native" and nat["tokens_out"] == sum(a.shape[1] - j[0].numel() for a, j in zip(alone, jobs))
    # the loop woke the 18 readers far fewer times than there were tokens (no flush tokens set: bursts of flush_max = 32)
    assert nat["reader_wakeups"] <= 18 * 3 < nat["tokens_out"]


@pytest.mark.parametrize("with_owner", [True, False])
def test_native_resume_in_place_for_returning_sequences(with_owner):
    """test_returning_sequences_resume_in_the_slot_that_holds_their_prompt on the native loop: the resume is decided by the loop
    itself when the join executes (dtk_slot_lcp behind the steps in flight).  An owner looks in its own last slot only; an
    owner-less sequence in every free slot that holds its image."""
    proc = fake_processor(VOCAB, NIMG)
    kw = dict(bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS], do_sample=True, max_length=NIMG + 60)
    firsts = [(*_prompt(proc, j % 2, extra=[60 + j]), 300 + j) for j in range(4)]
    alone_dev = ScriptedDevice()
    out1 = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)[0] for i, p, s in firsts]
    seconds = [(o[: NIMG + 1 + max(1, (o.numel() - NIMG - 1) // 2)], p, 400 + j) for j, (o, (_, p, _)) in enumerate(zip(out1, firsts))]
    out2 = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)[0] for i, p, s in seconds]
    for resume in (True, False):
        dev = ScriptedDevice(slots=5)
        eng = NativeBatchEngine(dev, max_batch=4, resume_in_place=resume)
        got = {}

        def worker(j):
            for wave, (i, p, s) in enumerate((firsts[j], seconds[j])):
                got[(j, wave)] = dev.generate(input_ids=i[None], pixel_values=p, seed=s, sequence_owner=j if with_owner else None, **kw)[0]
        _threads(4, worker)
        st = eng.stats()
        eng.close()
        for j in range(4):
            assert torch.equal(got[(j, 0)], out1[j]) and torch.equal(got[(j, 1)], out2[j]), (resume, j)
        if resume:
            assert st["resumed_in_place"] == dev.resumes == 4 and dev.tail_prefills <= 4
        else:
            assert st["resumed_in_place"] == dev.resumes == 0 and dev.tail_prefills >= 4
        assert sorted(eng.free) == [0, 1, 2, 3]


class _BurstRecorder:
    """a streamer that speaks the burst protocol and remembers how the tokens arrived"""

    def __init__(self):
        self.bursts, self.prompt, self.ended = [], None, False

    def put(self, value):
        self.prompt = value.reshape(-1).tolist()

    def put_token(self, t):
        self.bursts.append([t])

    def put_tokens(self, ts):
        self.bursts.append(list(ts))

    def end(self):
        self.ended = True


def test_readers_wake_once_per_source_line():
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=9, max_positions=NIMG + 200)
    eng = NativeBatchEngine(dev, max_batch=8, flush_tokens=dev.newline, flush_max=24)
    ids, px = _prompt(proc, 4)
    kw = dict(bad_words_ids=[[IMG]], suppress_tokens=[EOS], eos_token_id=-1, do_sample=True, max_new_tokens=150)
    alone = [ScriptedDevice(max_positions=NIMG + 200).generate(input_ids=ids[None], pixel_values=px, seed=70 + k, **kw) for k in range(8)]
    recs = [_BurstRecorder() for _ in range(8)]
    streams = [TokenStreamer(flush_on=set(dev.newline)) for _ in range(8)]
    seen = [None] * 8
    eng.expect(8, timeout=60)

    def worker(k):
        consumer = threading.Thread(target=lambda: seen.__setitem__(k, list(streams[k])))
        consumer.start()
        from detikzify_amd.util import StreamerList
        out = dev.generate(input_ids=ids[None], pixel_values=px, seed=70 + k, streamer=StreamerList([recs[k], streams[k]]), **kw)
        consumer.join(timeout=30)
        assert torch.equal(out, alone[k])
    _threads(8, worker)
    st = eng.stats()
    eng.close()
    nl = set(dev.newline)
    for k in range(8):
        new = alone[k][0, NIMG:].tolist()
        assert recs[k].ended and recs[k].prompt == ids.tolist()
        assert [t for b in recs[k].bursts for t in b] == new == seen[k]         # same tokens, same order, nothing lost
        for b in recs[k].bursts[:-1]:           # every burst but the last: whole lines (several, if the reader was slow), or the cap
            assert b[-1] in nl or len(b) >= 24, b
    lines = sum(sum(t in nl for t in a[0, NIMG:].tolist()) for a in alone)
    assert st["reader_wakeups"] <= lines + 8 * (150 // 24 + 2) and st["reader_wakeups"] < st["tokens_out"] // 2
    assert st["steps"] <= 150 + 4       # the eight sequences moved together


def test_per_token_objects_keep_their_exact_semantics_on_the_native_loop():
    """stopping criteria and foreign streamers need every token as it is made: the loop wakes their reader per token; the sequence
    stops exactly where the criterion fires even though the device has run ahead (those tokens are discarded)"""
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=4)
    eng = NativeBatchEngine(dev, max_batch=3)
    ids, px = _prompt(proc, 0)
    T = ids.numel()
    kw = dict(input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS], do_sample=True, seed=21)
    full = ScriptedDevice().generate(max_new_tokens=30, **kw)
    seen = []

    def criterion(input_ids, scores):
        seen.append(input_ids.shape[1])
        assert scores is None and input_ids[0, :T].tolist() == ids.tolist()
        return input_ids.shape[1] >= T + 7
    rec = _Recorder()
    out = dev.generate(max_new_tokens=30, stopping_criteria=[criterion], streamer=rec, **kw)
    assert torch.equal(out, full[:, : T + 7]) and seen == list(range(T + 1, T + 8))
    assert [e[1] for e in rec.events[1:-1]] == [[t] for t in full[0, T:T + 7].tolist()] and rec.events[-1] == ("end",)
    # an abort that is already set stops after the first burst; one set from outside stops a running sequence
    ctl = ExplicitAbort()
    ctl.abort()
    assert dev.generate(max_new_tokens=30, stopping_criteria=[ctl], **kw).shape[1] <= T + 30
    ctl2 = ExplicitAbort()
    slow = ScriptedDevice.decode_batch_wait
    dev.decode_batch_wait = lambda: (time.sleep(0.002), slow(dev))[1]
    timer = threading.Timer(0.03, ctl2.abort)
    timer.start()
    cut = dev.generate(max_new_tokens=120, max_length=None, stopping_criteria=[ctl2], **{**kw, "seed": 22})
    timer.join()
    ref = ScriptedDevice().generate(max_new_tokens=120, **{**kw, "seed": 22})
    assert T < cut.shape[1] < ref.shape[1] and torch.equal(cut, ref[:, : cut.shape[1]])
    # plain TokenStreamer (no flush_on): its consumer gets every token as an int, as it is made
    ts = TokenStreamer()
    th = threading.Thread(target=lambda: dev.generate(max_new_tokens=6, streamer=ts, **kw))
    th.start()
    streamed = list(ts)
    th.join()
    assert streamed == full[0, T:T + 6].tolist()
    eng.close()
    assert sorted(eng.free) == [0, 1, 2]


def test_a_failing_streamer_or_criterion_frees_its_slot_only_native():
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=4)
    eng = NativeBatchEngine(dev, max_batch=3)
    ids, px = _prompt(proc, 1)
    kw = dict(bad_words_ids=[[IMG]], suppress_tokens=[EOS], do_sample=True)
    good = ScriptedDevice().generate(input_ids=ids[None], pixel_values=px, seed=11, max_new_tokens=40, **kw)

    class BadStreamer(_Recorder):
        def put(self, value):
            super().put(value)
            if len(self.events) == 4:
                raise KeyError("consumer went away")

    def bad_criterion(input_ids, scores):
        if input_ids.shape[1] >= NIMG + 5:
            raise ZeroDivisionError("criterion bug")
        return False
    out, errs = {}, {}

    def run(name, **extra):
        try:
            out[name] = dev.generate(input_ids=ids[None], pixel_values=px, max_new_tokens=40, **kw, **extra)
        except BaseException as e:  # noqa: BLE001
            errs[name] = e
    ths = [threading.Thread(target=run, args=("good",), kwargs=dict(seed=11)),
           threading.Thread(target=run, args=("streamer",), kwargs=dict(seed=12, streamer=BadStreamer())),
           threading.Thread(target=run, args=("criterion",), kwargs=dict(seed=13, stopping_criteria=[bad_criterion]))]
    [t.start() for t in ths]
    [t.join(timeout=60) for t in ths]
    assert not any(t.is_alive() for t in ths)
    assert isinstance(errs.get("streamer"), KeyError) and isinstance(errs.get("criterion"), ZeroDivisionError)
    assert "good" not in errs and torch.equal(out["good"], good)
    again = dev.generate(input_ids=ids[None], pixel_values=px, seed=11, max_new_tokens=40, **kw)
    eng.close()
    assert torch.equal(again, good) and sorted(eng.free) == [0, 1, 2]


def test_bench_shaped_batch_of_64_threads_two_passes_native():
    """64 threads call generate() at once, twice: the first step waits for all of them (dtk_engine_expect), the sequences then move
    together — one step per token for all 64 — and no step is launched past a sequence's budget"""
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=65, max_positions=NIMG + 64)
    eng = NativeBatchEngine(dev, max_batch=64, flush_tokens=dev.newline)
    ids, px = _prompt(proc, 2)
    outs = {}

    def one(i):
        outs[i] = dev.generate(input_ids=ids[None], pixel_values=px, seed=5000 + i, do_sample=True, bad_words_ids=[[IMG]],
                               suppress_tokens=[EOS], eos_token_id=-1, max_new_tokens=48)
    steps0 = 0
    for rep in range(2):
        outs.clear()
        eng.expect(64, timeout=60)
        _threads(64, one)
        st = eng.stats()
        assert len(outs) == 64 and all(o.shape == (1, NIMG + 48) for o in outs.values())
        assert st["steps"] - steps0 <= 48 + 3, st["steps"] - steps0
        steps0 = st["steps"]
        assert all(len(dev.ctx[s]) == NIMG + 48 for s in range(64))        # exactly the budget: nothing speculative beyond it
    st = eng.stats()
    eng.close()
    assert st["wasted_slot_steps"] == 0 and st["tokens_out"] == 2 * 64 * 48
    assert len(eng.prefix_cache) == 1 and dev.forks == 128 and sorted(eng.free) == list(range(64))
    alone = ScriptedDevice(max_positions=NIMG + 64).generate(input_ids=ids[None], pixel_values=px, seed=5007, do_sample=True, bad_words_ids=[[IMG]],
                                                             suppress_tokens=[EOS], eos_token_id=-1, max_new_tokens=48)
    assert torch.equal(outs[7], alone)


def test_device_failure_reaches_every_reader_and_every_later_join():
    proc = fake_processor(VOCAB, NIMG)

    class Dying(ScriptedDevice):
        steps = 0

        def decode_batch_wait(self):
            Dying.steps += 1
            if Dying.steps > 10:
                raise AssertionError("HIP error 719 in k_gemv_b")
            return super().decode_batch_wait()
    dev = Dying(slots=4)
    eng = NativeBatchEngine(dev, max_batch=3)
    ids, px = _prompt(proc, 0)
    errs = []

    def one(k):
        try:
            dev.generate(input_ids=ids[None], pixel_values=px, seed=k, do_sample=True, suppress_tokens=[EOS], max_new_tokens=100)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)
    ths = [threading.Thread(target=one, args=(k,)) for k in range(3)]
    [t.start() for t in ths]
    [t.join(timeout=60) for t in ths]
    assert not any(t.is_alive() for t in ths) and len(errs) == 3 and all("719" in str(e) for e in errs), errs
    with pytest.raises(_lib.DtkError, match="719"):     # the engine stays failed: a later join is told why
        dev.generate(input_ids=ids[None], pixel_values=px, max_new_tokens=3)
    eng.close()
    assert sorted(eng.free) == [0, 1, 2]


def test_a_failing_join_fails_alone():
    """a prompt the device refuses (the reference's own ValueError for a broken image-token run) fails ITS generate(); the sequences
    that are decoding go on, the slot is free again, and an image whose encode failed is not remembered as cached"""
    proc = fake_processor(VOCAB, NIMG)

    class Picky(ScriptedDevice):
        def prefill(self, input_ids, pixel_values=None, **kw):
            if int(input_ids.reshape(-1)[-1]) == 77:
                raise ValueError("The image patch tokens should be consecutive.")
            return super().prefill(input_ids, pixel_values, **kw)
    dev = Picky(slots=4)
    eng = NativeBatchEngine(dev, max_batch=3)
    ids, px = _prompt(proc, 0)
    bad_ids, bad_px = _prompt(proc, 1, extra=[77])
    kw = dict(suppress_tokens=[EOS], do_sample=True, max_new_tokens=60)
    good = ScriptedDevice().generate(input_ids=ids[None], pixel_values=px, seed=5, **kw)
    box = {}

    def ok():
        box["good"] = dev.generate(input_ids=ids[None], pixel_values=px, seed=5, **kw)
    th = threading.Thread(target=ok)
    th.start()
    with pytest.raises(ValueError, match="image patch tokens should be consecutive"):
        dev.generate(input_ids=bad_ids[None], pixel_values=bad_px, seed=6, **kw)
    th.join(timeout=60)
    assert torch.equal(box["good"], good)
    # image 1's prefix WAS encoded into a prefix-cache slot before its tail failed: a later rollout of image 1 forks it
    ids1, px1 = _prompt(proc, 1)
    assert torch.equal(dev.generate(input_ids=ids1[None], pixel_values=px1, seed=8, **kw),
                       ScriptedDevice().generate(input_ids=ids1[None], pixel_values=px1, seed=8, **kw))
    eng.close()
    assert sorted(eng.free) == [0, 1, 2]


def test_both_engines_run_the_same_parallel_search(monkeypatch):
    """simulate_parallel with a fixed seed under DTK_ENGINE=python and under the native loop: the same rollouts and scores"""
    proc = fake_processor(VOCAB, NIMG)
    image = sketch_image(9, 96)

    def run(kind):
        monkeypatch.setenv("DTK_ENGINE", kind)
        dev = ScriptedDevice(slots=10)
        pipe = DetikzifyPipeline(dev, proc, metric="fast", document_class=SyntheticTikzDocument, max_length=NIMG + 40, compile_timeout=None)
        res = sorted((doc.code, score) for score, doc in simulate_parallel(pipe, image, trees=8, expansions_per_tree=3))
        return res, dev.last_batch_stats
    py, st_py = run("python")
    nat, st_nat = run("native")
    assert len(py) == 24 and py == nat
    assert st_py["engine"] == "python" and st_nat["engine"] == "native" and st_nat["joins"] == st_py["joins"]
    assert st_nat["resumed_in_place"] == st_py["resumed_in_place"]      # resume-or-prefill is a property of the tree, not of timing


def test_engine_c_abi_refuses_misuse():
    lib = _lib.load_library()
    h = C.c_void_p()
    assert lib.dtk_engine_create(None, C.byref(h)) == -1
    assert lib.dtk_engine_create_ops(None, C.byref(h)) == -1
    ops = _lib.DtkEngineOps()           # no callbacks
    assert lib.dtk_engine_create_ops(C.byref(ops), C.byref(h)) == -1
    dev = ScriptedDevice(slots=4)
    eng = NativeBatchEngine(dev, max_batch=3)
    n, st, buf = C.c_int32(0), C.c_int32(0), (C.c_int64 * 4)()
    assert lib.dtk_engine_read(eng._h, 0, buf, 4, C.byref(n), C.byref(st), 0) == -3          # no sequence in the slot yet
    assert lib.dtk_engine_read(eng._h, 99, buf, 4, C.byref(n), C.byref(st), 0) == -1
    assert lib.dtk_engine_set_option(eng._h, b"depth", 3) == -1 and lib.dtk_engine_set_option(eng._h, b"nope", 1) == -1
    assert lib.dtk_engine_set_flush_tokens(eng._h, (C.c_int64 * 1)(-5), 1) == -1
    ids = torch.tensor([IMG] * NIMG, dtype=torch.int64)
    j = _lib.DtkJoin()
    j.slot, j.n_ids, j.ids, j.prefix_src, j.max_new_tokens = 7, NIMG, ids.data_ptr(), -1, 4       # slot 7 of 4 decoding slots
    assert lib.dtk_engine_join(eng._h, C.byref(j)) == -3 and b"not a free decoding slot" in j.error_out
    j.slot, j.n_ids = 0, 0
    assert lib.dtk_engine_join(eng._h, C.byref(j)) == -1
    assert lib.dtk_engine_get_stats(eng._h, None) == -1 and lib.dtk_engine_leave(eng._h, -1) == -1
    eng.close()
    eng.close()         # idempotent
    with pytest.raises(_lib.DtkError, match="closed"):
        with eng.sequence(ids, None, {}):
            pass
    # struct layouts: the ctypes mirrors against the C compiler's own sizeof / offsetof
    assert lib.dtk_abi_struct_size(6) == C.sizeof(_lib.DtkJoin) and lib.dtk_abi_struct_size(7) == C.sizeof(_lib.DtkEngineStats)
    assert lib.dtk_abi_struct_size(8) == C.sizeof(_lib.DtkEngineOps)
    assert lib.dtk_abi_struct_size(9) == _lib.DtkJoin.sampling.offset and lib.dtk_abi_struct_size(10) == _lib.DtkJoin.error_out.offset
