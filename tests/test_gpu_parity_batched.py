"""
GPU parity (-m gpu) of the ROLLOUTS/sec kernel set at FULL size against the CPU oracle, and of the decode step at long
contexts.  Round 2 pinned the 64-slot kernels (k_gemv_b / k_gemv_bx / resid_split / k_attn_tail_b<256>, shared-prefix reads)
only transitively (tiny model vs oracle, then device vs device); these tests put the shipped default configuration of a
64-slot, full-depth step next to `DetikzifyOracle` directly.

Reference anchors: the forward the oracle restates is detikzify/model/v1/modeling_detikzify.py:218-283, the sampler
configuration is the one detikzify/infer/generate.py:218-227 passes to HF `generate`.

Tolerances (stated where used):
  * logits of a step: the device may be no further from the fp32 oracle than 1.5 x the bf16-policy oracle is, + 2e-3
    (two correct bf16 pipelines random-walk apart with depth);
  * greedy tokens: identical, except where the oracle's own top-2 gap is within 2 bf16 ulps of the top logit (near-tie),
    at most n // 8 of those;
  * sampled tokens: EXACTLY the oracle sampler's counter-based draw from the device's logits of that step (integer work);
  * the peaked-logits weight set: 16 of 16 greedy tokens identical, no near-tie rule.
"""
import gc
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampling
from oracle.model import DetikzifyOracle
from tests.helpers import rel_l2, sketch_image
from tests.test_gpu_parity import weights_from_device

ULP = 2.0 ** -7          # one bf16 ulp relative to the value's binade top (8 significant bits)
GAP_BINS = (0.0, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, float("inf"))


def top2_gap_ulps(logits, bad, begin, first):
    """top-1 minus top-2 of the processed scores, in bf16 ulps of the top logit"""
    top2 = torch.topk(sampling.mask_scores(logits, bad, begin, first), 2)[0]
    return float(top2[0] - top2[1]) / (float(top2[0].abs()) * ULP + 1e-30)


def histogram(gaps):
    counts = [0] * (len(GAP_BINS) - 1)
    for g in gaps:
        for b in range(len(counts)):
            if GAP_BINS[b] <= g < GAP_BINS[b + 1]:
                counts[b] += 1
                break
    return " ".join(f"[{GAP_BINS[b]:g},{GAP_BINS[b + 1]:g}):{c}" for b, c in enumerate(counts))


def oracle_snapshot(o):
    return list(o.llm.k), list(o.llm.v), o.llm.pos


def oracle_restore(o, snap):
    o.llm.k, o.llm.v, o.llm.pos = list(snap[0]), list(snap[1]), snap[2]


@pytest.mark.parametrize("name,weight_format", [("detikzify-ds-7b", "bf16"), ("detikzify-cl-7b", "fp8")])
def test_batched_headline_matches_cpu_oracle(name, weight_format):
    """The 64-slot batched decode step exactly as `bench.py`'s rollouts/sec phases run it — full depth, 65 slots allocated
    (64 decoding + the prefix-cache slot), every default (k_gemv_bx for gate/up and lm_head, resid_split for o_proj / down,
    k_attn_tail_b<256>, forked slots reading the shared image prefix from the source slot) — against the CPU oracle.
    The image prefix is prefilled once into slot 64 and forked into slots 0..63; 8 SAMPLED steps with per-slot seeds make
    the 64 contexts diverge, then 8 GREEDY steps.  For slots {0, 17, 40, 63} (one per 16-slot MFMA column tile) every step's
    logits are compared with the oracle teacher-forced on the device's tokens (fp32 envelope on slot 40, bf16-policy
    distance on all four); every sampled token of all 64 slots must equal the oracle sampler's draw from the device's
    logits; greedy tokens of the four slots follow the near-tie rule."""
    from detikzify_amd.model import load
    t_start = time.perf_counter()
    model, proc = load(name, synthetic=1234, max_positions=512, weight_format=weight_format, batch_slots=65)
    try:
        cfg = model.config.oracle_dict()
        w = weights_from_device(model, cfg)
        enc = proc(images=sketch_image(0, 224), return_tensors="pt")
        ids, px = enc.input_ids[0], enc.pixel_values
        n_img = ids.numel()
        img_tok, eos = cfg["image_token_id"], 2
        NS, SRC, N_SAMPLED, N_GREEDY = 64, 64, 8, 8
        watch, watch32 = (0, 17, 40, 63), (40,)
        assert model.num_slots() >= 65

        model.set_sampling(do_sample=False, slot=SRC)
        dev_prefill = model.prefill(ids, px, slot=SRC, return_logits=True)
        for s in range(NS):
            model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, top_k=0, seed=4242 + s, bad_ids=[img_tok],
                               begin_suppress_ids=[eos], slot=s)
            model.kv_fork(SRC, s, n_img)          # whole-prefix fork: KV rows and the next-token logits
        assert torch.equal(model.get_logits_slot(63), dev_prefill)

        toks = [[] for _ in range(NS)]
        logit_log = {s: [] for s in watch}        # logits AFTER step i of slot s
        prev = [dev_prefill] * NS
        draws_checked = 0
        for i in range(N_SAMPLED + N_GREEDY):
            if i == N_SAMPLED:
                for s in range(NS):               # draw counter restarts: begin-suppress applies to this step again
                    model.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=[eos], slot=s)
            model.decode_batch_launch(range(NS))
            out = model.decode_batch_wait()
            for s in range(NS):
                toks[s].append(out[s])
                if i < N_SAMPLED:
                    rt, _ = sampling.draw(prev[s], 0.8, 0, 0.95, 4242 + s, i, [img_tok], [eos], i == 0)
                    assert out[s] == rt, f"slot {s} sampled draw {i}: device {out[s]}, oracle draw from the device's logits {rt}"
                    draws_checked += 1
            prev = [model.get_logits_slot(s) for s in range(NS)]
            for s in watch:
                logit_log[s].append(prev[s])
            assert all(model.context_len_slot(s) == n_img + i + 1 for s in watch)
        assert len({tuple(t[:N_SAMPLED]) for t in toks}) > NS // 2, "the per-slot seeds did not make the contexts diverge"

        o16 = DetikzifyOracle(cfg, w, precision="bf16")
        ref = o16.prefill(ids, px[0])
        snap16 = oracle_snapshot(o16)
        o32 = DetikzifyOracle(cfg, w, precision="fp32")
        truth = o32.prefill(ids, px[0])
        snap32 = oracle_snapshot(o32)
        e_dev, e_orc = rel_l2(dev_prefill, truth), rel_l2(ref, truth)
        assert e_dev < 1.5 * e_orc + 2e-3

        worst_ratio, worst_r16, near_ties, near_tie_steps, identical, gaps = 0.0, 0.0, 0, 0, 0, []
        for s in watch:
            oracle_restore(o16, snap16)
            if s in watch32:
                oracle_restore(o32, snap32)
            logits = ref
            for i, t in enumerate(toks[s]):
                if i >= N_SAMPLED:
                    first = i == N_SAMPLED
                    gaps.append(top2_gap_ulps(logits, [img_tok], [eos], first))
                    near_tie_steps += gaps[-1] <= 2.0 + 1e-3
                    top2 = torch.topk(sampling.mask_scores(logits, [img_tok], [eos], first), 2)[1].tolist()
                    if top2[0] == t:
                        identical += 1
                    else:       # a flip: only at a near-tie, and only to the oracle's runner-up
                        assert gaps[-1] <= 2.0 + 1e-3 and t == top2[1], (s, i, t, top2, gaps[-1])
                        near_ties += 1
                else:
                    gaps.append(top2_gap_ulps(logits, [img_tok], [eos], i == 0))
                logits = o16.step(t)
                r16 = rel_l2(logit_log[s][i], logits)
                worst_r16 = max(worst_r16, r16)
                assert r16 < 3e-2, (s, i, r16)
                if s in watch32:
                    t32 = o32.step(t)
                    d, o = rel_l2(logit_log[s][i], t32), rel_l2(logits, t32)
                    worst_ratio = max(worst_ratio, d / (1.5 * o + 2e-3))
                    assert d < 1.5 * o + 2e-3, (s, i, d, o)
        n_greedy_total = N_GREEDY * len(watch)
        # uniform synthetic rows put the oracle's own top-2 within 2 bf16 ulps in ~30 % of the steps (histogram below); two correct
        # bf16 pipelines order such a pair either way, so the budget is counted against the NEAR-TIE steps, not against all steps
        # (the peaked weight set below has no near-ties and demands 16 of 16)
        budget = max(1, (3 * near_tie_steps + 3) // 4)
        assert near_ties <= budget, f"{near_ties} flips in {near_tie_steps} near-tie steps of {n_greedy_total} greedy steps: too many"
        print(f"batched {name}{' fp8' if weight_format == 'fp8' else ''}, 64 slots x {N_SAMPLED + N_GREEDY} steps: prefill logits vs fp32: "
              f"device {e_dev:.2e} oracle {e_orc:.2e}; slots {watch}: step logits dev-vs-bf16-oracle worst {worst_r16:.2e}, worst "
              f"ratio to the fp32 envelope {worst_ratio:.2f}; greedy {identical}/{n_greedy_total} identical ({near_ties} flips to the runner-up in {near_tie_steps} near-tie steps); "
              f"{draws_checked} sampled draws exact (64 slots x {N_SAMPLED}); oracle top-2 gap histogram (bf16 ulps of the top logit, "
              f"{len(gaps)} steps): {histogram(gaps)}; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()


def test_long_context_steps_match_cpu_oracle():
    """ds-7b decode steps at contexts ~700 (where bench.py's 512-token rollouts end) and ~1900 (the API allows 2048) against
    the CPU oracle: the prompt is the image prefix + seeded text tokens; the oracle prefills it ONCE (KV kept across the two
    checkpoints), the device prefills to the checkpoint and decodes 3 greedy steps there (the attention kernel walks 700 /
    1900 keys of KV the prefill GEMMs wrote).  Same logits envelope as the short-context tests, tokens by the near-tie rule."""
    from detikzify_amd.model import load
    t_start = time.perf_counter()
    model, proc = load("detikzify-ds-7b", synthetic=1234, max_positions=2048)
    try:
        cfg = model.config.oracle_dict()
        w = weights_from_device(model, cfg)
        enc = proc(images=sketch_image(0, 224), return_tensors="pt")
        ids, px = enc.input_ids[0], enc.pixel_values
        img_tok, eos = cfg["image_token_id"], 2
        g = torch.Generator().manual_seed(7)
        text = torch.randint(3, cfg["vocab"] - 1, (1900,), generator=g)
        text[text == img_tok] = 5
        full = torch.cat([ids, text])
        o16, o32 = DetikzifyOracle(cfg, w, precision="bf16"), DetikzifyOracle(cfg, w, precision="fp32")
        model.set_sampling(do_sample=False, bad_ids=[img_tok])
        report, done = [], 0
        for T in (700, 1900):
            prompt = full[:T]
            dev = model.prefill(prompt, px, return_logits=True)
            if done == 0:
                ref, truth = o16.prefill(prompt, px[0]), o32.prefill(prompt, px[0])
            else:       # extend both oracles from where the previous checkpoint's prompt ended
                for o in (o16, o32):
                    o.llm.truncate(done)
                h16 = o16.llm.forward(o16.llm.embed(prompt[done:]))
                h32 = o32.llm.forward(o32.llm.embed(prompt[done:]))
                ref, truth = o16.llm.logits(h16[-1]), o32.llm.logits(h32[-1])
            done = T
            e_dev, e_orc = rel_l2(dev, truth), rel_l2(ref, truth)
            assert e_dev < 1.5 * e_orc + 2e-3, (T, e_dev, e_orc)
            logits, worst, ties = ref, 0.0, 0
            for i in range(3):
                model.decode_launch()
                t = model.decode_wait()
                rt = sampling.greedy(logits, [img_tok], [], False)
                if rt != t:
                    assert top2_gap_ulps(logits, [img_tok], [], False) <= 2.0 + 1e-3, (T, i, t, rt)
                    ties += 1
                lg = model.get_logits()
                logits, t32 = o16.step(t), o32.step(t)
                d, o = rel_l2(lg, t32), rel_l2(logits, t32)
                worst = max(worst, d / (1.5 * o + 2e-3))
                assert d < 1.5 * o + 2e-3, (T, i, d, o)
            assert model.context_len() == T + 3
            report.append(f"context {T}: prefill logits vs fp32: device {e_dev:.2e} oracle {e_orc:.2e}; 3 decode steps worst ratio to the "
                          f"envelope {worst:.2f}, {3 - ties}/3 tokens identical")
        print("ds-7b long-context decode vs CPU oracle: " + "; ".join(report) + f"; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()


PEAKED_SEED, PEAKED_BETA = 0, 2.0      # chosen by tools/peaked_seed_search.py on the CPU oracle (every top-2 gap >= MIN_PEAKED_GAP ulps)
MIN_PEAKED_GAP = 4.0


def test_peaked_logits_weight_set_is_token_identical():
    """A second synthetic weight set whose logits are PEAKED (tests/helpers.py::peaked_lm_head: lm_head rows scaled by
    log-normal powers of two, exact in bf16): the uniform set gives 32 k equal-variance logits, i.e. a top-2 gap below one
    bf16 ulp of the top logit in ~10 % of the steps, and the near-tie rule of the other tests then forgives a mismatch.  Here
    the oracle's own top-2 gap is >= 4 ulps at every one of the 16 steps (asserted), so there is nothing to forgive:
    ds-7b, full depth, 16 of 16 greedy tokens must be identical — on the single-sequence decode graph AND in a 64-slot
    batched step (slot 37, its 63 neighbours decoding other contexts)."""
    from detikzify_amd.model import load
    from tests.helpers import peaked_lm_head
    t_start = time.perf_counter()
    model, proc = load("detikzify-ds-7b", synthetic=1234, max_positions=512, batch_slots=65)
    try:
        cfg = model.config.oracle_dict()
        head = model.read_tensor("lm_head.weight").float().reshape(cfg["vocab"], cfg["hidden"])
        model.load_tensor("lm_head.weight", peaked_lm_head(head, PEAKED_BETA, PEAKED_SEED).to(torch.bfloat16))
        w = weights_from_device(model, cfg)
        enc = proc(images=sketch_image(0, 224), return_tensors="pt")
        ids, px = enc.input_ids[0], enc.pixel_values
        n_img, img_tok, eos, N = ids.numel(), cfg["image_token_id"], 2, 16
        # single sequence
        model.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=[eos])
        model.prefill(ids, px)
        single = []
        for _ in range(N):
            model.decode_launch()
            single.append(model.decode_wait())
        # 64-slot batch: slot 37 greedy, the others sampling their own continuations
        model.set_sampling(do_sample=False, slot=64)
        model.prefill(ids, px, slot=64)
        for s in range(64):
            if s == 37:
                model.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=[eos], slot=s)
            else:
                model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=99 + s, bad_ids=[img_tok], begin_suppress_ids=[eos], slot=s)
            model.kv_fork(64, s, n_img)
        batched = []
        for _ in range(N):
            model.decode_batch_launch(range(64))
            batched.append(model.decode_batch_wait()[37])
        o16 = DetikzifyOracle(cfg, w, precision="bf16")
        logits, gaps, ref = o16.prefill(ids, px[0]), [], []
        for i in range(N):
            gaps.append(top2_gap_ulps(logits, [img_tok], [eos], i == 0))
            ref.append(sampling.greedy(logits, [img_tok], [eos], i == 0))
            logits = o16.step(ref[-1])
        print(f"peaked weight set (lm_head rows x 2^round({PEAKED_BETA} z), seed {PEAKED_SEED}): oracle top-2 gaps in bf16 ulps "
              f"{' '.join(f'{g:.0f}' for g in gaps)} (histogram {histogram(gaps)}); greedy tokens single {sum(a == b for a, b in zip(single, ref))}/{N}, "
              f"batched slot 37 {sum(a == b for a, b in zip(batched, ref))}/{N} identical; {time.perf_counter() - t_start:.0f} s")
        assert min(gaps) >= MIN_PEAKED_GAP, f"the weight set is not peaked enough on this host's oracle: min gap {min(gaps):.2f} ulps"
        assert single == ref, (single, ref)
        assert batched == ref, (batched, ref)
    finally:
        del model
        gc.collect()
