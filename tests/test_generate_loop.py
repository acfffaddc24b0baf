"""Host side of the generate / rollout / parallel-MCTS path, on the CPU: the REAL `DetikzifyForCausalLM.generate`,
`BatchEngine`, `DetikzifyGenerator`, `DetikzifyPipeline`, `ImageSim` and `simulate_parallel` run on top of a scripted
device (the C-ABI calls of modeling.py replaced by a toy language model whose next token is a pure function of
(image, context, sampling seed)).  What this pins without a GPU:

  * the HF generate protocol the reference relies on (infer/generate.py:218-227, util/generation.py:25-66): the
    streamer sees the prompt once, then one token at a time, then end(); every stopping criterion is evaluated on every
    token with the ids so far; EOS / max_length / max_new_tokens stop exactly there; one decode step is kept in flight
    and never more than the length budget allows
  * a sequence decoded in a KV slot of the batch engine (threads, prefix-cache fork, in-place reuse, donors, slot
    recycling) gets exactly the tokens it gets when decoded alone — the scripted device asserts that every
    prefix-reusing prefill really finds that image's prefix in the slot
  * root-parallel trees share one metric object: every yielded score is the score of ITS document
"""
import threading
import time
import zlib
from types import SimpleNamespace

import pytest
import torch

from detikzify_amd.evaluate.imagesim import ImageSim
from detikzify_amd.infer import DetikzifyGenerator, DetikzifyPipeline, SyntheticTikzDocument
from detikzify_amd.infer.batching import BatchEngine, simulate_parallel
from detikzify_amd.model.modeling import DetikzifyForCausalLM, GenerationConfig
from detikzify_amd.util import ExplicitAbort, TokenStreamer

from .helpers import fake_processor, sketch_image

IMG, EOS, VOCAB, NIMG = 1, 2, 512, 12


class _Lib:
    def __init__(self, dev):
        self.dev = dev

    def dtk_context_len_slot(self, ctx, s):
        return len(self.dev.ctx[s])

    def dtk_max_decode_slots(self, ctx):       # the rule of include/dtk.h: <= 5 slots -> 0..3 decode; else 16 / 32 / 64 column slots
        n = self.dev.slots
        return min(n, 4 if n <= 5 else (64 if n > 33 else (32 if n > 17 else 16)))


class _Vision:
    """pooled 'features' = 4x4 average pool of the pixels; the sleep releases the GIL like the ctypes call does"""

    def pooled_only(self, pixel_values):
        time.sleep(0.002)
        return torch.nn.functional.adaptive_avg_pool2d(pixel_values.float(), 4).flatten() + 1.5

    def __call__(self, pixel_values, **_):
        return SimpleNamespace(pooler_output=self.pooled_only(pixel_values)[None], last_hidden_state=None)


class ScriptedDevice(DetikzifyForCausalLM):
    """DetikzifyForCausalLM without a GPU: everything above the C-ABI wrappers is the shipped code"""
    SINGLE = -1

    def __init__(self, slots=0, max_positions=160):      # no super().__init__: that one creates the device context
        cfg = SimpleNamespace(image_token_id=IMG, eos_token_id=EOS, pad_token_id=0, bos_token_id=1, vocab=VOCAB,
                              max_positions=max_positions, pooling_mode="cos")
        cfg.text_config = cfg
        self.config = cfg
        self.generation_config = GenerationConfig(eos_token_id=EOS, pad_token_id=0, bos_token_id=1)
        self.name_or_path = "scripted"
        self.model = SimpleNamespace(vision_model=_Vision())
        self.reuse_prefix, self.batch_engine, self._weights_ready = False, None, True
        self._vit_lock, self._single_busy = threading.RLock(), threading.Lock()
        self.lib, self._ctx, self.slots = _Lib(self), None, slots
        self.ctx, self.img, self.samp, self.gen0 = {}, {}, {}, {}
        self.pending, self.bpending = [], []
        self.max_in_flight = self.launches = self.prefills = self.forks = self.tail_prefills = 0
        self.forced, self.resumes = {}, 0
        tok = fake_processor(VOCAB, NIMG).tokenizer
        self.newline = [i for i, t in enumerate(tok._id2tok) if "\n" in t and i > 2]
        self.plain = [i for i, t in enumerate(tok._id2tok) if "\n" not in t and i > 2]

    # ---- the toy LM ---------------------------------------------------------------------------------------------
    def _next(self, s):
        if s in self.forced:            # a resumed slot forwards the last token of its prompt instead of sampling
            self.ctx[s].append(self.forced.pop(s))
            return self.ctx[s][-1]
        sp, ctx = self.samp[s], self.ctx[s]
        seed = sp.get("seed", 0) if sp.get("do_sample") else 0
        h = zlib.crc32(repr((self.img[s], ctx, seed)).encode())
        r, pick = (h & 0xFFFF) / 65536.0, h >> 16
        first = len(ctx) == self.gen0[s]
        if r < 0.05 and not (first and EOS in sp.get("begin_suppress_ids", ())) and EOS not in sp.get("always_suppress_ids", ()):
            tok = EOS
        elif r < 0.4:
            tok = self.newline[pick % len(self.newline)]
        else:
            tok = self.plain[pick % len(self.plain)]
        assert tok not in sp.get("bad_ids", ())
        ctx.append(tok)
        return tok

    # ---- C-ABI wrappers of modeling.py, scripted ------------------------------------------------------------------
    def num_slots(self):
        return self.slots

    def set_sampling(self, do_sample=False, temperature=1.0, top_p=1.0, top_k=0, seed=0, bad_ids=(), begin_suppress_ids=(),
                     always_suppress_ids=(), slot=None):
        self.samp[self.SINGLE if slot is None else slot] = dict(
            do_sample=do_sample, seed=seed, bad_ids=tuple(bad_ids), begin_suppress_ids=tuple(begin_suppress_ids),
            always_suppress_ids=tuple(always_suppress_ids))

    def prefill(self, input_ids, pixel_values=None, return_logits=False, reuse=None, slot=None):
        assert not self.bpending, "prefill while a batched step is un-collected"
        s = self.SINGLE if slot is None else slot
        ids = [int(t) for t in input_ids.reshape(-1)]
        key = self.image_key(pixel_values) if pixel_values is not None else 0
        if reuse:       # the engine claims this slot already holds the image prefix: check it
            n = next((i for i, t in enumerate(ids) if t != IMG), len(ids))
            assert self.img.get(s) == key and self.ctx[s][:n] == ids[:n], "prefix reuse without the prefix in the slot"
            self.tail_prefills += 1
        self.pending.clear()
        self.ctx[s], self.img[s], self.gen0[s] = ids, key, len(ids)
        self.prefills += 1

    def kv_fork(self, src_slot, dst_slot, n_tokens):
        assert len(self.ctx[src_slot]) >= n_tokens
        self.ctx[dst_slot], self.img[dst_slot] = self.ctx[src_slot][:n_tokens], self.img[src_slot]
        self.gen0[dst_slot] = n_tokens
        self.forks += 1

    def best_lcp_slot(self, slots, ids, key=0):
        ids, best = [int(t) for t in ids.reshape(-1)], None
        for s in slots:
            if s not in self.ctx or self.img.get(s) != key:
                continue
            n = next((i for i, (a, b) in enumerate(zip(self.ctx[s], ids)) if a != b), min(len(self.ctx[s]), len(ids)))
            if n > (best[1] if best else 0):
                best = (s, n)
        return best

    def resume_slot(self, slot, ids, key=0):
        ids = [int(t) for t in ids.reshape(-1)]
        assert self.img.get(slot) == key and self.ctx[slot][:len(ids) - 1] == ids[:-1], "resume without the prompt in the slot"
        assert not any(slot in step for step in self.bpending), "resume of a slot that is part of an un-collected step"
        self.ctx[slot], self.gen0[slot], self.forced[slot] = ids[:-1], len(ids), ids[-1]
        self.resumes += 1

    def decode_launch(self):
        assert len(self.ctx[self.SINGLE]) < self.config.max_positions
        self.pending.append(self._next(self.SINGLE))
        self.launches += 1
        self.max_in_flight = max(self.max_in_flight, len(self.pending))

    def decode_wait(self):
        return self.pending.pop(0)

    def decode_batch_launch(self, active_slots):
        slots = list(active_slots)
        assert len(self.bpending) < 2 and slots == sorted(slots)
        self.bpending.append({s: self._next(s) for s in slots})

    def decode_batch_wait(self):
        time.sleep(0.0002)
        step = self.bpending.pop(0)
        return [step.get(s, -1) for s in range(64)]


def _prompt(proc, image_seed, extra=()):
    enc = proc(images=sketch_image(image_seed, 96), return_tensors="pt")
    ids = torch.cat([enc.input_ids[0], torch.tensor(list(extra), dtype=torch.int64)])
    return ids, enc.pixel_values


class _Recorder:
    def __init__(self):
        self.events = []

    def put(self, value):
        self.events.append(("put", value.reshape(-1).tolist()))

    def end(self):
        self.events.append(("end",))


def test_generate_protocol_streamer_criteria_and_length_budget():
    dev, proc = ScriptedDevice(), fake_processor(VOCAB, NIMG)
    ids, px = _prompt(proc, 0)
    T = ids.numel()
    seen = []

    def criterion(input_ids, scores):
        seen.append(input_ids.shape[1])
        assert scores is None and input_ids[0, :T].tolist() == ids.tolist()
        return False

    rec = _Recorder()
    out = dev.generate(input_ids=ids[None], pixel_values=px, bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS],
                       streamer=rec, stopping_criteria=[criterion], max_length=T + 40, do_sample=True, seed=7)
    new = out[0, T:].tolist()
    assert out.shape[0] == 1 and out[0, :T].tolist() == ids.tolist() and 1 <= len(new) <= 40
    assert (new[-1] == EOS or len(new) == 40) and EOS not in new[:-1] and new[0] != EOS and IMG not in new
    assert rec.events[0] == ("put", ids.tolist()) and rec.events[-1] == ("end",)
    assert [e[1] for e in rec.events[1:-1]] == [[t] for t in new]          # one token per put, in order
    assert seen == list(range(T + 1, T + len(new) + 1))                    # every token, with the ids so far
    assert dev.max_in_flight == 2 and dev.launches <= min(40, len(new) + 1)   # one step ahead, never past the budget

    # same seed -> same tokens; max_new_tokens wins over max_length; EOS-free run stops at the budget exactly
    again = dev.generate(input_ids=ids[None], pixel_values=px, bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS],
                         max_length=T + 40, do_sample=True, seed=7)
    assert torch.equal(again, out)
    dev.launches = 0
    fixed = dev.generate(input_ids=ids, pixel_values=px, suppress_tokens=[EOS], max_new_tokens=9, max_length=5, eos_token_id=-1)
    assert fixed.shape == (1, T + 9) and dev.launches == 9
    assert dev.generate(input_ids=ids[None], pixel_values=px, max_length=T).shape == (1, T)      # nothing to generate
    assert dev.generate(input_ids=ids[None], pixel_values=px, max_length=10 ** 6, suppress_tokens=[EOS],
                        eos_token_id=-1).shape[1] == dev.config.max_positions                    # clipped to the KV capacity

    # a criterion that fires stops after that token; ExplicitAbort is polled; TokenStreamer gets plain ints
    stop_at = dev.generate(input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS], max_new_tokens=30,
                           stopping_criteria=[lambda i, s: i.shape[1] >= T + 4])
    assert stop_at.shape[1] == T + 4
    ctl = ExplicitAbort()
    ctl.abort()
    assert dev.generate(input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS], max_new_tokens=30,
                        stopping_criteria=[ctl]).shape[1] == T + 1
    ts = TokenStreamer()
    th = threading.Thread(target=lambda: dev.generate(input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS],
                                                      max_new_tokens=6, streamer=ts, do_sample=True, seed=3))
    th.start()
    streamed = list(ts)
    th.join()
    assert len(streamed) == 6 and all(isinstance(t, int) for t in streamed)

    # a second generate() on the same (engine-less) model while one is running is refused, not interleaved
    gate, inside = threading.Event(), threading.Event()
    blocker = threading.Thread(target=lambda: dev.generate(
        input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS], max_new_tokens=3,
        stopping_criteria=[lambda i, s: (inside.set(), gate.wait(10), False)[2]]))
    blocker.start()
    assert inside.wait(10)
    with pytest.raises(Exception, match="concurrent generate"):
        dev.generate(input_ids=ids[None], pixel_values=px, max_new_tokens=3)
    gate.set()
    blocker.join()
    assert dev.generate(input_ids=ids[None], pixel_values=px, suppress_tokens=[EOS], max_new_tokens=3).shape[1] == T + 3

    with pytest.raises(ValueError):
        dev.generate(input_ids=torch.stack([ids, ids]), pixel_values=px)
    with pytest.raises(NotImplementedError):
        dev.generate(input_ids=ids[None], pixel_values=px, bad_words_ids=[[3, 4]])
    dev._weights_ready = False
    with pytest.raises(Exception, match="no weights"):
        dev.generate(input_ids=ids[None], pixel_values=px)


def test_sequences_in_engine_slots_decode_exactly_as_alone():
    """the Python-driven BatchEngine (one launch / wait pair per token, the round-3 loop; tests/test_native_engine.py runs the same
    jobs through the native run loop): a sequence decoded in a slot is the ids it gets alone"""
    proc = fake_processor(VOCAB, NIMG)
    jobs = []       # (prompt ids, pixels, seed): 3 images, prompts = the bare image prefix or prefix + a few tokens
    for j in range(18):
        ids, px = _prompt(proc, j % 3, extra=[40 + j, 50 + j][: j % 3])
        jobs.append((ids, px, 100 + j))
    kw = dict(bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS], do_sample=True, max_length=NIMG + 60)
    alone_dev = ScriptedDevice()
    alone = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw) for i, p, s in jobs]

    dev = ScriptedDevice(slots=5)
    eng = BatchEngine(dev, max_batch=4)
    got, errs = [None] * len(jobs), []

    def worker(k):
        try:
            for j in range(k, len(jobs), 6):
                i, p, s = jobs[j]
                got[j] = dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)
        except BaseException as e:  # noqa: BLE001
            errs.append(e)

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(6)]     # more threads than slots
    [t.start() for t in ths]
    [t.join(timeout=60) for t in ths]
    assert not any(t.is_alive() for t in ths) and not errs, errs[:1]
    eng.close()
    assert dev.batch_engine is None and not dev.bpending
    for a, g in zip(alone, got):
        assert torch.equal(a, g)
    st = eng.stats()
    # every join got its image prefix without a full prefill: one ViT + prefix prefill per image CHANGE of the prefix
    # cache at most, everything else forked / reused in place; the scripted device has checked every reuse claim
    assert st["joins"] == 18 and dev.forks + st["inplace_reuses"] + st["resumed_in_place"] == 18
    assert dev.prefills == st["prefix_encodes"] + dev.tail_prefills and st["prefix_encodes"] <= 18


def test_parallel_trees_share_one_metric_but_never_a_score():
    """simulate_parallel hands the SAME ImageSim to every tree; each tree runs update -> compute -> reset on its own
    thread (reference infer/generate.py:293-298).  Every yielded score must be the similarity of its own document."""
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=7)
    pipe = DetikzifyPipeline(dev, proc, metric="model", document_class=SyntheticTikzDocument, max_length=NIMG + 50,
                             compile_timeout=None)
    assert isinstance(pipe.metric, ImageSim) and pipe.metric.mode == "cos"
    image = sketch_image(5, 96)
    compute = pipe.metric.compute
    pipe.metric.compute = lambda: (time.sleep(0.003), compute())[1]     # widen the update -> compute -> reset window
    res = list(simulate_parallel(pipe, image, trees=6, expansions_per_tree=5))
    assert len(res) == 30 and dev.batch_engine is None
    check = ImageSim.from_detikzify(dev, proc)
    ref_img = pipe.load(image)
    scored = 0
    for score, doc in res:
        if doc.is_rasterizable:
            assert score == pytest.approx(check.get_similarity(doc.rasterize(), ref_img), abs=1e-12), doc.code[:40]
            scored += 1
        else:
            assert score == -1
    assert scored >= 10 and pipe.metric.n_samples == 0
    assert 6 <= dev.last_batch_stats["joins"] <= 30 and dev.last_batch_stats["prefix_encodes"] == 1   # a rollout from a finished node never reaches the model

    # one tree = the sequential search of the reference, through the same code path and the same device
    seq = list(simulate_parallel(pipe, image, trees=1, expansions_per_tree=4))
    gen = DetikzifyGenerator(dev, proc, image=ref_img, metric=pipe.metric, **pipe.gen_kwargs)
    assert len(seq) == 4 and len(list(gen.simulate(expansions=2))) == 2


def test_more_trees_than_decode_slots_take_turns():
    """8 trees over 3 decode slots (simulate_parallel(slots=3)): a tree holds a slot only while it generates and gives it up for
    its reward, so the trees take turns; every rollout of every tree arrives, never more than 3 sequences decode at once, and —
    every tree drawing its sampling seeds AND the tie-breaks of its search from its own stream (MonteCarlo(rng=...)) — the
    rollouts are the ones a slot per tree produces, run after run, whatever the thread timing."""
    proc = fake_processor(VOCAB, NIMG)
    image = sketch_image(9, 96)

    def run(slots):
        dev = ScriptedDevice(slots=10)
        pipe = DetikzifyPipeline(dev, proc, metric="fast", document_class=SyntheticTikzDocument, max_length=NIMG + 40, compile_timeout=None)
        peak = [0]
        launch = dev.decode_batch_launch
        dev.decode_batch_launch = lambda active: (peak.__setitem__(0, max(peak[0], len(active))), launch(active))[1]
        res = sorted((doc.code, score) for score, doc in simulate_parallel(pipe, image, trees=8, expansions_per_tree=3, slots=slots))
        return res, peak[0], dev.last_batch_stats

    crowded, peak, st = run(3)
    roomy, peak_all, _ = run(None)
    assert len(crowded) == len(roomy) == 24 and peak <= 3 < peak_all <= 8 and st["joins"] >= 8
    assert crowded == roomy == run(None)[0] == run(5)[0]          # a fixed-seed parallel search is reproducible


def test_returning_sequences_resume_in_the_slot_that_holds_their_prompt():
    """A sequence whose prompt is a path inside an earlier sequence of the same image (an MCTS tree returning to a node of its
    previous rollout) finds the free slot whose cache still holds that path and continues there: no fork, no tail prefill; the
    first token of its first step is the forced last prompt token and never reaches the caller.  Token for token what the same
    prompt and seed produce alone; resume_in_place=False takes the fork + prefill path to the same tokens."""
    proc = fake_processor(VOCAB, NIMG)
    kw = dict(bad_words_ids=[[IMG]], begin_suppress_tokens=[EOS], do_sample=True, max_length=NIMG + 60)
    firsts = [(*_prompt(proc, j % 2, extra=[60 + j]), 300 + j) for j in range(4)]
    alone_dev = ScriptedDevice()
    out1 = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)[0] for i, p, s in firsts]
    seconds = [(o[: NIMG + 1 + max(1, (o.numel() - NIMG - 1) // 2)], p, 400 + j) for j, (o, (_, p, _)) in enumerate(zip(out1, firsts))]
    out2 = [alone_dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)[0] for i, p, s in seconds]
    for resume in (True, False):
        dev = ScriptedDevice(slots=5)
        eng = BatchEngine(dev, max_batch=4, resume_in_place=resume)
        got, errs = {}, []

        def worker(j):
            try:
                for wave, (i, p, s) in enumerate((firsts[j], seconds[j])):
                    got[(j, wave)] = dev.generate(input_ids=i[None], pixel_values=p, seed=s, **kw)[0]
            except BaseException as e:  # noqa: BLE001
                errs.append(e)

        ths = [threading.Thread(target=worker, args=(j,)) for j in range(4)]
        [t.start() for t in ths]
        [t.join(timeout=60) for t in ths]
        assert not any(t.is_alive() for t in ths) and not errs, errs[:1]
        eng.close()
        for j in range(4):
            assert torch.equal(got[(j, 0)], out1[j]) and torch.equal(got[(j, 1)], out2[j]), (resume, j)
        st = eng.stats()
        if resume:      # every second-wave prompt was still cached in a free slot (4 slots, 4 threads: nobody evicted it)
            assert st["resumed_in_place"] == dev.resumes == 4 and dev.tail_prefills <= 4
        else:
            assert st["resumed_in_place"] == dev.resumes == 0 and dev.tail_prefills >= 4


def test_device_error_surfaces_from_every_layer_without_hanging():
    """a failing native call inside a batched step reaches the caller of simulate_parallel (engine -> sequence ->
    generate in the rollout's worker thread -> streamer.propagate_error -> the tree's thread -> the result queue)"""
    proc = fake_processor(VOCAB, NIMG)

    class Dying(ScriptedDevice):
        steps = 0

        def decode_batch_wait(self):
            Dying.steps += 1
            if Dying.steps > 25:
                raise RuntimeError("HIP error 719 in k_gemv_b")
            return super().decode_batch_wait()

    dev = Dying(slots=5)
    pipe = DetikzifyPipeline(dev, proc, metric="fast", document_class=SyntheticTikzDocument, max_length=NIMG + 50,
                             compile_timeout=None)
    done, box = threading.Event(), []

    def run():
        try:
            list(simulate_parallel(pipe, sketch_image(6, 96), trees=4, expansions_per_tree=50))
        except BaseException as e:  # noqa: BLE001
            box.append(e)
        done.set()

    threading.Thread(target=run, daemon=True).start()
    assert done.wait(timeout=60), "simulate_parallel hangs after a device error"
    assert box and "HIP error 719" in str(box[0])
    assert dev.batch_engine is None

    # single sequence: the error comes out of generate() itself
    class DyingSingle(ScriptedDevice):
        def decode_wait(self):
            raise RuntimeError("HIP error 700")
    ids, px = _prompt(proc, 0)
    with pytest.raises(RuntimeError, match="700"):
        DyingSingle().generate(input_ids=ids[None], pixel_values=px, max_new_tokens=5)


@pytest.mark.parametrize("slots", [16, 13, 6])
def test_several_images_in_flight_keep_their_own_prefix_and_reward(slots):
    """BASELINE config 5 on one GPU: 4 images x 3 trees in one batched decode (16 slots: a prefix-cache slot per image; 13: one prefix-cache slot,
    the rest through donors / in place; 6 slots: the trees queue for slots, no prefix cache at all).  The scripted device checks every prefix-reuse claim against
    the image actually held by the slot; every score must be the similarity to the tree's OWN image."""
    from detikzify_amd.infer.batching import simulate_parallel_images
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=slots)
    pipe = DetikzifyPipeline(dev, proc, metric="model", document_class=SyntheticTikzDocument, max_length=NIMG + 40,
                             compile_timeout=None)
    images = [sketch_image(20 + i, 96) for i in range(4)]
    res = list(simulate_parallel_images(pipe, images, trees_per_image=3, expansions_per_tree=3))
    assert len(res) == 36 and sorted({k for k, _, _ in res}) == [0, 1, 2, 3]
    check = ImageSim.from_detikzify(dev, proc)
    refs = [pipe.load(im) for im in images]
    for k, score, doc in res:
        if doc.is_rasterizable:
            sims = [check.get_similarity(doc.rasterize(), r) for r in refs]
            assert score == pytest.approx(sims[k], abs=1e-12)
    st = dev.last_batch_stats
    if slots >= 13:
        assert st["prefix_encodes"] == 4        # every image encoded once (prefix cache, then donors / in place)
    if slots == 16:                             # a prefix-cache slot per image: every join is a fork — or a resume in place
        assert dev.forks + st["resumed_in_place"] == st["joins"] and dev.prefills == 4 + dev.tail_prefills      # tails: rollouts from inner nodes
    assert st["prefix_encodes"] + dev.forks + st["inplace_reuses"] + st["resumed_in_place"] >= st["joins"]


def test_a_failing_streamer_or_criterion_frees_its_slot_and_leaves_the_others_alone():
    """exceptions raised by caller-supplied objects inside generate() (streamer.put, a stopping criterion) leave through
    generate(); in the engine the sequence's slot is recycled and the other sequences keep decoding"""
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=4)
    eng = BatchEngine(dev, max_batch=3)
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
    # the engine is still usable and every decode slot is free again
    again = dev.generate(input_ids=ids[None], pixel_values=px, seed=11, max_new_tokens=40, **kw)
    eng.close()
    assert torch.equal(again, good) and sorted(eng.free) == [0, 1, 2] and not eng.zombies and eng.error is None


def test_bench_shaped_batch_of_64_threads_two_passes():
    """what bench.py's batched phase does: 64 threads call generate() at once (engine.expect holds the first step until
    all have joined), twice; every sequence gets its full length and the batch stays full"""
    proc = fake_processor(VOCAB, NIMG)
    dev = ScriptedDevice(slots=65, max_positions=NIMG + 64)
    eng = BatchEngine(dev, max_batch=64)
    ids, px = _prompt(proc, 2)
    outs = {}

    def one(i):
        outs[i] = dev.generate(input_ids=ids[None], pixel_values=px, seed=5000 + i, do_sample=True, bad_words_ids=[[IMG]],
                               suppress_tokens=[EOS], eos_token_id=-1, max_new_tokens=48)
    for rep in range(2):
        outs.clear()
        steps0 = eng.steps
        eng.expect(64, timeout=60)      # (bench.py uses the 0.5 s default; a loaded CI box may need longer to start 64 threads)
        ths = [threading.Thread(target=one, args=(i,)) for i in range(64)]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        assert not any(t.is_alive() for t in ths)
        assert len(outs) == 64 and all(o.shape == (1, NIMG + 48) for o in outs.values())
        assert eng.steps - steps0 <= 48 + 3, eng.steps - steps0        # lock-step: one step per token for all 64
    eng.close()
    assert len(eng.prefix_cache) == 1 and dev.forks == 128 and sorted(eng.free) == list(range(64))


def test_pipeline_simulate_with_parallel_trees_by_expansions_and_by_timeout():
    proc = fake_processor(VOCAB, NIMG)
    pipe = DetikzifyPipeline(ScriptedDevice(slots=4), proc, metric="fast", document_class=SyntheticTikzDocument,
                             max_length=NIMG + 30, compile_timeout=None)
    image = sketch_image(9, 96)
    assert len(list(pipe.simulate(image, expansions=2, trees=3))) == 6
    assert len(list(pipe.simulate(image, expansions=2))) == 2                 # the sequential search of the reference
    t0 = time.perf_counter()
    got = list(pipe.simulate(image, timeout=0.4, trees=3))                    # every tree stops after its own time budget
    assert got and time.perf_counter() - t0 < 20


def test_leaving_a_parallel_search_early_stops_every_tree_before_the_engine_goes():
    """`for s, d in pipe.simulate(trees=N): break` with no expansion limit: the trees must not keep searching in the
    background, and none may outlive the batch engine (it would fall onto the single-sequence path of a model in use)"""
    dev = ScriptedDevice(slots=4)
    pipe = DetikzifyPipeline(dev, fake_processor(VOCAB, NIMG), metric="fast", document_class=SyntheticTikzDocument,
                             max_length=NIMG + 30, compile_timeout=None)
    before = {t.ident for t in threading.enumerate()}
    stream = pipe.simulate(sketch_image(9, 96), trees=3)          # expansions=None, timeout=None: endless
    first = [next(stream) for _ in range(2)]
    assert len(first) == 2
    t0 = time.perf_counter()
    stream.close()
    assert time.perf_counter() - t0 < 30
    time.sleep(0.1)
    leftover = [t for t in threading.enumerate() if t.ident not in before and t.is_alive()]
    assert not leftover, leftover
    assert dev.batch_engine is None                               # closed, after the trees had stopped
    launches = dev.launches
    time.sleep(0.2)
    assert dev.launches == launches and not dev.bpending          # nothing decodes in the background
    assert len(list(pipe.simulate(sketch_image(9, 96), expansions=1, trees=2))) == 2      # and the model is usable again


def test_generate_refuses_hf_arguments_it_does_not_implement():
    dev = ScriptedDevice()
    ids, px = _prompt(fake_processor(VOCAB, NIMG), 3)
    kw = dict(input_ids=ids, pixel_values=px, max_new_tokens=4, do_sample=False)
    dev.generate(**kw, num_beams=1, repetition_penalty=1.0, use_cache=True, attention_mask=None, logits_processor=[])
    for bad in (dict(num_beams=4), dict(repetition_penalty=1.2), dict(num_return_sequences=2), dict(min_length=5),
                dict(no_repeat_ngram_size=3), dict(penalty_alpha=0.6), dict(logits_processor=[object()])):
        with pytest.raises(NotImplementedError):
            dev.generate(**kw, **bad)
    with pytest.raises(TypeError):
        dev.generate(**kw, adapter_input_ids=None)
    # several EOS ids (HF allows a list): generation stops on any of them
    out = dev.generate(**{**kw, "max_new_tokens": 40}, eos_token_id=[EOS, *dev.newline])
    assert int(out[0, -1]) in {EOS, *dev.newline} and out.shape[1] < NIMG + 40


def test_generate_protocol_matches_hf_generate(golden_dir):
    """tests/golden/generate_protocol.json: what the installed HF GenerationMixin.generate (the call the reference makes)
    hands to a streamer and to a stopping criterion — (1, T) prompt once, then a (1,) int64 CPU tensor per token, end();
    the criterion after every token with the (1, length) ids so far.  Our generate() does exactly the same."""
    import json

    from tests.golden.make_golden import ProtocolRecorder
    golden = json.loads((golden_dir / "generate_protocol.json").read_text())
    dev, proc = ScriptedDevice(), fake_processor(VOCAB, NIMG)
    ids = torch.tensor([[IMG, 5, 6, 7]])        # 4 prompt tokens like the golden (no image: the scripted LM needs none)
    for name, kw, stop in (("max_new_tokens", dict(max_new_tokens=6), None), ("criterion_stops", dict(max_new_tokens=20), 7)):
        st, cr = ProtocolRecorder(), ProtocolRecorder(stop)
        out = dev.generate(input_ids=ids, do_sample=False, streamer=st, stopping_criteria=[cr], suppress_tokens=[EOS], **kw)
        assert list(out.shape) == golden[name]["out_shape"]
        assert st.events == golden[name]["streamer"] and cr.events == golden[name]["criterion"], name


def example_loader(args, local_rank):
    """DTK_EXAMPLE_LOADER hook of examples/mcts_multi_gpu.py (tests/test_compile_pool.py): the scripted device + the toy processor"""
    return ScriptedDevice(slots=min(64, args.trees) + 1), fake_processor(VOCAB, NIMG)
