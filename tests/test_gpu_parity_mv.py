"""
GPU parity tests (-m gpu) of the multi-vector decode step: contexts with at most 5 slots (<= 4 decoding + a prefix slot) decode
with k_gemv_mv (csrc/kernels_decode_mv.hip) — the single-sequence GEMVs carrying 1, 2 or 4 input vectors — instead of a
16-column MFMA tile.  These are the per-rank shapes of BASELINE config 4 at N = 4 / 8 (4 / 2 trees per rank: reference
examples/eval.py:80-83).  Covered here at toy size: the kernel against its single-sequence twin bit for bit, the step against the
CPU oracle, independence of a slot from the vector count and the active set, fp8 rows, GQA, the host engine on top of it; the
full-size comparison with the CPU oracle is tests/test_gpu_parity_batched.py (slot counts 3 and 5 of its parametrisation).
Tolerances as in tests/test_gpu_parity.py.
"""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampling
from oracle.model import DetikzifyOracle
from oracle.ops import f32_to_bits, rb
from oracle.synth import tensor_specs
from tests.helpers import TINY_CFG, TINY_V2_CFG, engines, rel_l2, sketch_image


def weights_from_device(model, cfg):
    return {name: model.read_tensor(name).float().reshape(shape) for name, shape, _, _ in tensor_specs(cfg)}


@pytest.fixture(scope="module")
def tiny_mv():
    from detikzify_amd.model import load
    model, proc = load("detikzify-tiny", synthetic=1234, batch_slots=5)
    assert model.max_decode_slots() == 4          # slot 4 is prefix-only in this family
    return model, proc


def _prompts(proc, n=4):
    out = []
    for k, extra in enumerate(([], [70, 300, 41], [9] * 17, [5, 6])[:n]):
        enc = proc(images=sketch_image(10 + k, 96), return_tensors="pt")
        out.append((torch.cat([enc.input_ids[0], torch.tensor(extra, dtype=torch.long)]), enc.pixel_values))
    return out


@pytest.mark.parametrize("N,K", [(96, 256), (64, 4096), (33, 11008), (50, 360), (130, 2048)])
@pytest.mark.parametrize("mode", [0, 1])
def test_multi_vector_gemv_is_the_single_sequence_gemv_per_vector(tiny_mv, N, K, mode):
    """dtk_op_gemv_mv (k_gemv_mv, 1 / 2 / 4 vectors) against dtk_op_gemv (k_gemv) vector by vector: the same lane / chunk
    order, the same wave reduction, the same block size for the RMSNorm prologue -> identical bits; the result of a vector
    does not depend on how many vectors travel with it."""
    model, _ = tiny_mv
    g = torch.Generator().manual_seed(N * 31 + K + mode)
    W = rb(torch.randn(N, K, generator=g) * 0.05)
    X = rb(torch.randn(4, K, generator=g))
    nw = rb(1 + 0.1 * torch.randn(K, generator=g))
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    Wb, Xb, nb_ = f32_to_bits(W), f32_to_bits(X), f32_to_bits(nw)
    single = []
    for b in range(4):
        out = np.empty(N, dtype=np.uint16)
        xb = np.ascontiguousarray(Xb[b])
        model._check(model.lib.dtk_op_gemv(model._ctx, p(Wb), p(xb), p(nb_), N, K, mode, 1e-6, p(out)), "dtk_op_gemv")
        single.append(out)
    for nvec in (1, 2, 4):
        out = np.empty((nvec, N), dtype=np.uint16)
        xin = np.ascontiguousarray(Xb[:nvec])
        model._check(model.lib.dtk_op_gemv_mv(model._ctx, p(Wb), p(xin), p(nb_), N, K, mode, 1e-6, nvec, p(out)), "dtk_op_gemv_mv")
        for b in range(nvec):
            assert np.array_equal(out[b], single[b]), (nvec, b, int((out[b] != single[b]).sum()))


def test_multi_vector_step_tracks_oracle_and_is_batch_invariant(tiny_mv):
    """the <= 4-slot step: four sequences of different lengths / images in one step.  (1) per-slot logits follow the CPU oracle
    (teacher forced, the bounds of the single-sequence path); (2) a slot's tokens and logits are bit-identical whether it decodes
    alone (1 vector), next to one other slot (2) or three (4), in any slot; (3) slot 4 of the 5-slot context cannot decode."""
    from detikzify_amd._lib import DtkError
    model, proc = tiny_mv
    oracle = DetikzifyOracle(TINY_CFG, weights_from_device(model, TINY_CFG), precision="bf16")
    prompts = _prompts(proc)
    n_steps = 12
    for s, (ids, px) in enumerate(prompts):
        model.set_sampling(do_sample=False, bad_ids=[1], begin_suppress_ids=[2], slot=s)
        model.prefill(ids, px, slot=s)
    toks, logit_log = [[] for _ in prompts], [[] for _ in prompts]
    for step in range(n_steps):
        # 4 vectors; 4 vectors with a hole; 2 vectors (slot 1 returns after three steps away: its own context, untouched)
        active = [0, 1, 2, 3] if step < 6 else ([0, 2, 3] if step < 9 else [0, 1])
        model.decode_batch_launch(active)
        out = model.decode_batch_wait()
        for s in active:
            toks[s].append(out[s])
            logit_log[s].append(model.get_logits_slot(s))
        assert all(out[j] == -1 for j in range(16) if j not in active)
    worst, flips = 0.0, 0
    for s, (ids, px) in enumerate(prompts):
        logits = oracle.prefill(ids, px[0])
        for i, t in enumerate(toks[s]):
            rt = sampling.greedy(logits, [1], [2], i == 0)
            if rt != t:
                top2 = torch.topk(sampling.mask_scores(logits, [1], [2], i == 0), 2)[0]
                assert float(top2[0] - top2[1]) <= 2 * float(top2[0].abs()) * 2.0 ** -7 + 1e-6, (s, i, t, rt)
                flips += 1
            logits = oracle.step(t)
            worst = max(worst, rel_l2(logit_log[s][i], logits))
    print(f"multi-vector step: worst logits rel_l2 {worst:.2e}, {flips} near-tie flips over {sum(map(len, toks))} tokens")
    assert worst < 1e-2 and flips <= 3
    # vector-count / active-set / slot-index invariance: sequence 2 alone in slot 0 (1 vector), next to another slot in slot 1
    # (2 vectors) and in slot 3 (4 vectors) reproduces its tokens and logits bit for bit
    ids, px = prompts[2]
    for slot, others in ((0, []), (1, [0]), (3, [0, 1])):
        for o in others:
            model.set_sampling(do_sample=False, bad_ids=[1], begin_suppress_ids=[2], slot=o)
            model.prefill(prompts[0][0], prompts[0][1], slot=o)
        model.set_sampling(do_sample=False, bad_ids=[1], begin_suppress_ids=[2], slot=slot)
        model.prefill(ids, px, slot=slot)
        for i in range(len(toks[2])):
            model.decode_batch_launch([slot] + others)
            assert model.decode_batch_wait()[slot] == toks[2][i], (slot, i)
            assert torch.equal(model.get_logits_slot(slot), logit_log[2][i]), (slot, i)
    model.set_sampling(do_sample=False, slot=4)
    model.prefill(ids, px, slot=4)                     # prefix-only slot: prefill and fork are fine ...
    model.kv_fork(4, 0, ids.numel())
    with pytest.raises(DtkError, match="cannot decode"):
        model.decode_batch_launch([4])                 # ... decoding is not


def test_multi_vector_step_equals_plain_launches_and_the_mfma_family_within_rounding(tiny_mv):
    """graph replay == plain launches bit for bit; against the MFMA family (option mv_slots = 0 on the same context) the logits
    agree to accumulation order (rel-L2 1e-2: two bf16 pipelines) with the same greedy tokens away from near-ties"""
    model, proc = tiny_mv
    (ids, px), (ids_b, px_b) = _prompts(proc, 2)

    def run(steps=10):
        for s, (i_, p_) in enumerate(((ids, px), (ids_b, px_b))):
            model.set_sampling(do_sample=False, bad_ids=[1], begin_suppress_ids=[2], slot=s)
            model.prefill(i_, p_, slot=s)
        toks, logs = [], []
        for _ in range(steps):
            model.decode_batch_launch([0, 1])
            toks.append(model.decode_batch_wait()[:2])
            logs.append([model.get_logits_slot(0), model.get_logits_slot(1)])
        return toks, logs

    t_graph, l_graph = run()
    model.set_graph_mode(0)
    try:
        t_plain, l_plain = run()
    finally:
        model.set_graph_mode(1)
    assert t_graph == t_plain and all(torch.equal(a, b) for x, y in zip(l_graph, l_plain) for a, b in zip(x, y))
    model.set_option("mv_slots", 0)
    try:
        assert model.max_decode_slots() == 5
        t_mfma, l_mfma = run()
    finally:
        model.set_option("mv_slots", 4)
    assert rel_l2(l_graph[0][0], l_mfma[0][0]) < 1e-2 and rel_l2(l_graph[0][1], l_mfma[0][1]) < 1e-2
    for shape_opt in ("mv_shape_qkv", "mv_shape_o", "mv_shape_gu", "mv_shape_down", "mv_shape_lm_head"):
        for shape in (0, 1, 2, 3):              # every block shape of every role: the same bits (a row's chunk order is the lane's)
            model.set_option(shape_opt, shape)
            try:
                t_s, l_s = run(4)
            finally:
                model.set_option(shape_opt, -1)
            assert t_s == t_graph[:4], (shape_opt, shape)
            assert all(torch.equal(a, b) for x, y in zip(l_s, l_graph[:4]) for a, b in zip(x, y)), (shape_opt, shape)
    for threads in (256, 1024):                 # attention block shapes: keys are scored in the same order -> same tokens, logits to rounding
        model.set_option("mv_tail_threads", threads)
        try:
            t_s, l_s = run(4)
        finally:
            model.set_option("mv_tail_threads", 512)
        assert all(rel_l2(a, b) < 1e-2 for x, y in zip(l_s, l_graph[:4]) for a, b in zip(x, y)), threads


def test_multi_vector_fp8_rows_and_gqa():
    """fp8 rows (weight_format="fp8") and the GQA / llama3-rope / 128k-style multi-block sampler family through the
    multi-vector step: per-slot logits against the CPU oracle on the effective weights; slot invariance bit for bit."""
    from detikzify_amd.model import load
    for name, cfg, wf, seed in (("detikzify-tiny", TINY_CFG, "fp8", 1234), ("detikzify-tiny-v2", TINY_V2_CFG, "bf16", 4321)):
        model, proc = load(name, synthetic=seed, weight_format=wf, batch_slots=3)
        assert model.max_decode_slots() == 3
        img_tok = model.config.image_token_id
        oracle = DetikzifyOracle(cfg, weights_from_device(model, cfg), precision="bf16")
        size = 96 if cfg is TINY_CFG else 84
        encs = [proc(images=sketch_image(20 + k, size), return_tensors="pt") for k in range(2)]
        prompts = [(torch.cat([e.input_ids[0], torch.tensor(x, dtype=torch.long)]), e.pixel_values) for e, x in zip(encs, ([], [7, 8, 9]))]
        for s, (ids, px) in enumerate(prompts):
            model.set_sampling(do_sample=False, bad_ids=[img_tok], slot=s)
            model.prefill(ids, px, slot=s)
        toks, logs = [[], []], [[], []]
        for _ in range(10):
            model.decode_batch_launch([0, 1])
            out = model.decode_batch_wait()
            for s in (0, 1):
                toks[s].append(out[s]); logs[s].append(model.get_logits_slot(s))
        worst = 0.0
        for s, (ids, px) in enumerate(prompts):
            logits = oracle.prefill(ids, px[0])
            for i, t in enumerate(toks[s]):
                rt = sampling.greedy(logits, [img_tok], [], False)
                if rt != t:
                    top2 = torch.topk(sampling.mask_scores(logits, [img_tok], [], False), 2)[0]
                    assert float(top2[0] - top2[1]) <= 2 * float(top2[0].abs()) * 2.0 ** -7 + 1e-6, (name, s, i, t, rt)
                logits = oracle.step(t)
                worst = max(worst, rel_l2(logs[s][i], logits))
        print(f"multi-vector step, {name} {wf}: worst logits rel_l2 vs oracle {worst:.2e}")
        assert worst < 1e-2
        ids, px = prompts[1]
        model.set_sampling(do_sample=False, bad_ids=[img_tok], slot=0)
        model.prefill(ids, px, slot=0)
        for i in range(10):
            model.decode_batch_launch([0])
            assert model.decode_batch_wait()[0] == toks[1][i]
            assert torch.equal(model.get_logits_slot(0), logs[1][i])
        del model


@pytest.mark.parametrize("BatchEngine", engines(), ids=lambda c: c.__name__)
def test_engine_and_parallel_trees_on_the_multi_vector_step(tiny_mv, BatchEngine):
    """the host stack on a 5-slot context: BatchEngine capacity 4 + one prefix slot; model.generate from four threads == each
    prompt generated alone; kv_fork / resume_slot in this family; simulate_parallel with 4 trees (config 4 at N = 4)"""
    from detikzify_amd.infer import DetikzifyPipeline, SyntheticTikzDocument
    from detikzify_amd.infer.batching import simulate_parallel
    model, proc = tiny_mv
    prompts = _prompts(proc)
    kw = dict(do_sample=True, temperature=0.8, top_p=0.95, top_k=0, max_new_tokens=24, bad_words_ids=[[1]],
              begin_suppress_tokens=[2], eos_token_id=-1)
    engine = BatchEngine(model)
    try:
        assert engine.capacity == 4 and engine.prefix_slots == [4]
        alone = [model.generate(input_ids=ids[None], pixel_values=px, seed=50 + i, **kw)[0].tolist() for i, (ids, px) in enumerate(prompts)]
        res, steps0 = [None] * 4, engine.steps

        def run(i):
            ids, px = prompts[i]
            res[i] = model.generate(input_ids=ids[None], pixel_values=px, seed=50 + i, **kw)[0].tolist()
        engine.expect(4, timeout=30.0)
        ths = [threading.Thread(target=run, args=(i,)) for i in range(4)]
        [t.start() for t in ths]
        [t.join(timeout=120) for t in ths]
        assert res == alone
        assert engine.steps - steps0 <= 24 + 2
    finally:
        engine.close()
    # fork of a whole prefilled prompt decodes at once and tracks its source; resume_slot continues bit for bit
    (ids, px) = prompts[1]
    key = model.image_key(px)
    for s in (0, 1):
        model.set_sampling(do_sample=False, bad_ids=[1], slot=s)
    model.prefill(ids, px, slot=0)
    model.kv_fork(0, 1, ids.numel())
    first = []
    for _ in range(8):
        model.decode_batch_launch([0, 1])
        out = model.decode_batch_wait()
        assert out[0] == out[1]
        first.append(out[0])
    model.resume_slot(1, torch.cat([ids, torch.tensor(first[:4])]), key)
    model.decode_batch_launch([1])
    assert model.decode_batch_wait()[1] == first[3]              # the forced last prompt token
    for i in range(4, 8):
        model.decode_batch_launch([1])
        assert model.decode_batch_wait()[1] == first[i]          # the same rows, the same kernels: the same continuation
    pipe = DetikzifyPipeline(model, proc, metric="model", document_class=SyntheticTikzDocument, max_length=70)
    res = list(simulate_parallel(pipe, sketch_image(8, 128), trees=4, expansions_per_tree=2))
    assert len(res) == 8 and all(-1.0 <= s <= 1.0 + 1e-6 for s, _ in res) and model.batch_engine is None
