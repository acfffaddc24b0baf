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
  * greedy tokens: identical, except where the oracle's own top-2 gap is within 2 bf16 ulps of the top logit (near-tie) AND the
    device's token is one the oracle scores within 2 ulps of its best; at most every second near-tie step may flip; ONE flip per
    run may lie between 2 and 3 ulps (a flip proves that the two logits' errors add up to the margin: 1.5 ulps each);
  * sampled tokens: EXACTLY the oracle sampler's counter-based draw from the device's logits of that step (integer work);
  * the peaked-logits weight set: every judged position identical (64 distinct contexts), no near-tie rule;
  * fp8 matrix-core steps (fp8 models, MXFP8 activations): the same rules against the oracle that quantises the same activations
    (LlamaOracle.act_quant), the tie unit widened as stated at MX_TIE.
"""
import gc
import time

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import sampling
from oracle.model import DetikzifyOracle
from tests.helpers import ENVELOPE, SLACK_LOGITS, rel_l2, sketch_image
from tests.test_gpu_parity import weights_from_device

ULP = 2.0 ** -7          # one bf16 ulp relative to the value's binade top (8 significant bits)
GAP_BINS = (0.0, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, float("inf"))
MX_TIE = 16.0            # near-tie unit of the fp8 matrix-core steps: bf16 ulps of the top logit (16 = 12.5 % of it, about the measured distance between the device and the quantising oracle, 0.14 rel-L2; round 4: worst flip 6.7 behind)
MX_R16 = 0.3             # sanity bound on that distance (two MXFP8 pipelines decorrelate to the quantisation noise: tests/test_gpu_parity_mx.py; the fp32 envelope is the test)


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


def _phases_vs_oracle(name, weight_format, batch_slots, phases, watch, watch32, max_positions=512, private_tail=0, tail_watch=(), act_fp8=None):
    """One context of `batch_slots` slots: the image prefix is prefilled once into the last slot and forked into every decoding
    slot; then `phases` = [(active slots, sampled steps, greedy steps, slots the step's kernels must cover)] run one after the
    other — which kernels a step runs is decided by its highest active slot (MFMA family: 1 / 2 / 4 column tiles of 16) or by the
    number of vectors (multi-vector family: 1 / 2 / 4), asserted per step through dtk_stats.last_batch_step_slots.  Every
    sampled token of every active slot must equal the oracle sampler's draw from the device's logits; for the `watch` slots
    every step's logits are compared with the CPU oracle teacher-forced on the device's tokens (bf16-policy distance on all,
    fp32 envelope on `watch32`), greedy tokens by the near-tie rule.  `private_tail` > 0: all slots then sample that many more
    tokens (device only), the oracle re-reads the whole sequence of the `tail_watch` slots in one pass and 4 greedy steps are
    compared there: attention over hundreds of PRIVATE keys per slot that the decode kernels themselves appended."""
    from detikzify_amd.model import load
    from tests.fullsize import host_side
    t_start = time.perf_counter()
    model, proc = load(name, synthetic=1234, max_positions=max_positions, weight_format=weight_format, batch_slots=batch_slots)
    try:
        if act_fp8 is not None:
            model.set_option("act_fp8", int(act_fp8))
        hs = host_side(model, proc, name, weight_format)       # weights + the prefix prefill of both oracles, shared by the model's full-size tests
        cfg, w, o16, o32 = hs.oracles(model)
        ids, px = hs.ids, hs.px
        n_img = ids.numel()
        img_tok, eos = cfg["image_token_id"], 2
        SRC = batch_slots - 1
        NS = min(SRC, model.max_decode_slots())
        seed_of = lambda s: 4242 + s
        model.set_sampling(do_sample=False, slot=SRC)
        dev_prefill = model.prefill(ids, px, slot=SRC, return_logits=True)
        for s in range(NS):
            model.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=[eos], slot=s)
            model.kv_fork(SRC, s, n_img)          # whole-prefix fork: KV rows and the next-token logits
        assert torch.equal(model.get_logits_slot(NS - 1), dev_prefill)

        log = {s: [] for s in range(NS)}          # per slot: (sampled?, first step after set_sampling?, token)
        logit_log = {s: [] for s in watch}        # logits AFTER each step of a watched slot
        prev = {s: dev_prefill for s in range(NS)}
        draws_checked, kinds, mx = 0, [], None
        for active, n_sampled, n_greedy, kind in phases:
            active = list(active)
            assert max(active) < NS
            for sampled, n in ((True, n_sampled), (False, n_greedy)):
                if not n:
                    continue
                for s in active:                  # (the draw counter restarts: begin-suppress applies to the next step again)
                    if sampled:
                        model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, top_k=0, seed=seed_of(s), bad_ids=[img_tok],
                                           begin_suppress_ids=[eos], slot=s)
                    else:
                        model.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=[eos], slot=s)
                for i in range(n):
                    model.decode_batch_launch(active)
                    out = model.decode_batch_wait()
                    got_kind = model.stats()["last_batch_step_slots"]
                    assert got_kind == kind, f"{len(active)} active slots up to {max(active)}: the step ran the {got_kind}-slot kernels, expected {kind}"
                    step_mx = bool(model.stats()["last_batch_step_fp8_mfma"])     # fp8 model, MFMA family: the fp8 matrix cores on MXFP8 activations
                    assert mx is None or mx == step_mx, "the kernel family of a context must not change with the active set"
                    mx = step_mx
                    for s in active:
                        if sampled:
                            rt, _ = sampling.draw(prev[s], 0.8, 0, 0.95, seed_of(s), i, [img_tok], [eos], i == 0)
                            assert out[s] == rt, f"slot {s} sampled draw {i} ({kind}-slot kernels): device {out[s]}, oracle draw from the device's logits {rt}"
                            draws_checked += 1
                        log[s].append((sampled, i == 0, out[s]))
                        prev[s] = model.get_logits_slot(s)
                        if s in logit_log:
                            logit_log[s].append(prev[s])
                kinds.append(kind)
        assert all(model.context_len_slot(s) == n_img + len(log[s]) for s in watch)
        if NS >= 8:
            assert len({tuple(t for _, _, t in log[s][:4]) for s in range(NS)}) > NS // 2, "the per-slot seeds did not make the contexts diverge"

        ref, truth, snap16, snap32 = hs.ref, hs.truth, hs.snap16, hs.snap32
        e_dev, e_orc, e_pair = rel_l2(dev_prefill, truth), rel_l2(ref, truth), rel_l2(dev_prefill, ref)
        assert e_dev < ENVELOPE * e_orc + SLACK_LOGITS

        # fp8 matrix-core steps (csrc/kernels_batch_mx.hip): the bf16-policy oracle quantises the same activations to MXFP8 from here
        # on (LlamaOracle.act_quant; prefill above ran with bf16 activations on both sides); the fp32 oracle stays the unquantised
        # truth.  e4m3 activations make the two pipelines random-walk apart in steps of 2^-4 instead of 2^-8, so the near-tie rule
        # is stated in the unit that fits: MX_TIE bf16 ulps of the top logit (asserted), and the sanity bound on the distance widens.
        assert mx == bool(act_fp8 and weight_format == "fp8" and batch_slots > 5), "MXFP8 activations are opt-in (dtk_set_option act_fp8): only then the MFMA-family step of an fp8 model runs on the fp8 matrix cores"
        o16.llm.act_quant = bool(mx)
        tie = MX_TIE if mx else 2.0
        worst_ratio, worst_r16, near_ties, near_tie_steps, identical, n_greedy_total, gaps, worst_behind, wide_flips = 0.0, 0.0, 0, 0, 0, 0, [], 0.0, 0
        worst_b16 = 0.0
        for s in watch:
            # the slot's tokens teacher-forced in ONE oracle pass each (rows16[i] / rows32[i] = logits after log[s][i]'s token)
            oracle_restore(o16, snap16)
            rows16 = o16.extend([t for _, _, t in log[s]])
            if mx:      # ... and once more with bf16 activations: the distance MXFP8 activations put between the device and the reference's arithmetic is ASSERTED (MX_BOUND)
                o16.llm.act_quant = False
                oracle_restore(o16, snap16)
                rows_b16 = o16.extend([t for _, _, t in log[s]])
                o16.llm.act_quant = True
                for i in range(len(log[s])):
                    rb16 = rel_l2(logit_log[s][i], rows_b16[i])
                    worst_b16 = max(worst_b16, rb16)
                    assert rb16 <= MX_BOUND, (s, i, rb16)
            if s in watch32:
                oracle_restore(o32, snap32)
                rows32 = o32.extend([t for _, _, t in log[s]])
            logits = ref
            for i, (sampled, first, t) in enumerate(log[s]):
                gaps.append(top2_gap_ulps(logits, [img_tok], [eos], first))
                if not sampled:
                    n_greedy_total += 1
                    near_tie_steps += gaps[-1] <= tie + 1e-3
                    masked = sampling.mask_scores(logits, [img_tok], [eos], first)
                    best = float(masked.max())
                    if float(masked[t]) == best and t == int(torch.nonzero(masked == best)[0]):
                        identical += 1      # the oracle's argmax (lowest index among exactly equal maxima, like torch.argmax and the device)
                    else:
                        # a flip: only at a near-tie, and only to a token the ORACLE itself scores within 2 bf16 ulps of its best (the
                        # runner-up, or — bf16 logits tie exactly now and then — any of several tokens at that distance)
                        # A flip proves |err(t)| + |err(best)| >= the oracle's margin between the two (the device ranked them the other
                        # way round), err = device logit - oracle logit: `tie` ulps = one ulp on each of the two bf16 logits.  The
                        # 64-slot v2-8b run (128 k candidates per step) produced one flip at 2.03 ulps in round 4: a flip may reach
                        # 1.5 x tie, but only ONE per run may lie beyond `tie` (counted, asserted below, printed).
                        behind = (best - float(masked[t])) / (abs(best) * ULP + 1e-30)
                        assert gaps[-1] <= 1.5 * tie + 1e-3 and behind <= 1.5 * tie + 1e-3, (s, i, t, torch.topk(masked, 3), gaps[-1], behind)
                        near_ties += 1
                        wide_flips += behind > tie + 1e-3
                        worst_behind = max(worst_behind, behind)
                logits = rows16[i]
                r16 = rel_l2(logit_log[s][i], logits)
                worst_r16 = max(worst_r16, r16)
                # sanity bound on the distance between the two bf16 pipelines (the fp32 envelope on `watch32` is the real test): what
                # the prefill of this model showed, with headroom — ds-7b ~2.9e-2, the 128 k-vocabulary v2-8b ~3.4e-2
                assert r16 < (MX_R16 if mx else max(3e-2, 1.25 * e_pair)), (s, i, r16, e_pair)
                if s in watch32:
                    t32 = rows32[i]
                    d, o = rel_l2(logit_log[s][i], t32), rel_l2(logits, t32)
                    worst_ratio = max(worst_ratio, d / (ENVELOPE * o + SLACK_LOGITS))
                    assert d < ENVELOPE * o + SLACK_LOGITS, (s, i, d, o)
        # uniform synthetic rows put the oracle's own top-2 within 2 bf16 ulps in ~25 % of the steps (histogram below); two correct
        # bf16 pipelines order such a pair either way, so flips are counted against the NEAR-TIE steps — at most every second one
        # (was three of four) — and must go to the oracle's runner-up (asserted above).  Whether the flips lean one
        # way is measured over 256 steps by test_greedy_margins_are_not_biased_against_the_oracle.
        budget = max(1, (near_tie_steps + 1) // 2)
        assert near_ties <= budget, f"{near_ties} flips in {near_tie_steps} near-tie steps of {n_greedy_total} greedy steps: too many"
        assert wide_flips <= 1, f"{wide_flips} flips beyond {tie:g} ulps (worst {worst_behind:.2f})"
        tail_report = ""
        if private_tail:
            everyone = list(range(NS))
            for s in everyone:
                model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, top_k=0, seed=seed_of(s) + 1000, bad_ids=[img_tok, eos], slot=s)
            seqs = {s: [t for _, _, t in log[s]] for s in tail_watch}
            model.decode_batch_launch(everyone)
            for i in range(private_tail):          # one step in flight, like the engine drives it
                if i + 1 < private_tail:
                    model.decode_batch_launch(everyone)
                out = model.decode_batch_wait()
                for s in tail_watch:
                    seqs[s].append(out[s])
            for s in everyone:
                model.set_sampling(do_sample=False, bad_ids=[img_tok], slot=s)
            tail_logits = {s: [model.get_logits_slot(s)] for s in tail_watch}
            tail_toks = {s: [] for s in tail_watch}
            for _ in range(4):
                model.decode_batch_launch(everyone)
                out = model.decode_batch_wait()
                for s in tail_watch:
                    tail_toks[s].append(out[s])
                    tail_logits[s].append(model.get_logits_slot(s))
            t_worst, t_same = 0.0, 0
            for s in tail_watch:
                full = torch.cat([ids, torch.tensor(seqs[s], dtype=torch.long)])
                assert model.context_len_slot(s) == full.numel() + 4 and full.numel() - n_img >= private_tail
                # both oracles read the slot's whole private sequence in one pass behind the shared image prefix, then the 4 compared
                # tokens in a second one
                o16.llm.act_quant = False
                oracle_restore(o16, snap16)
                oracle_restore(o32, snap32)
                lg16, lg32 = o16.extend(seqs[s], last_only=True), o32.extend(seqs[s], last_only=True)
                nx16, nx32 = o16.extend(tail_toks[s]), o32.extend(tail_toks[s])
                for i in range(5):
                    d, o = rel_l2(tail_logits[s][i], lg32), rel_l2(lg16, lg32)
                    t_worst = max(t_worst, d / (ENVELOPE * o + SLACK_LOGITS))
                    assert d < ENVELOPE * o + SLACK_LOGITS, (s, i, d, o)
                    if i == 4:
                        break
                    t = tail_toks[s][i]
                    rt = sampling.greedy(lg16, [img_tok], [], False)
                    if rt != t:
                        assert top2_gap_ulps(lg16, [img_tok], [], False) <= 2.0 + 1e-3, (s, i, t, rt)
                    else:
                        t_same += 1
                    lg16, lg32 = nx16[i], nx32[i]
            tail_report = (f"; after {private_tail} more sampled tokens per slot (contexts {n_img + len(log[tail_watch[0]]) + private_tail}, all but {n_img} keys private): "
                           f"slots {tuple(tail_watch)} 5 logit checks worst ratio to the envelope {t_worst:.2f}, {t_same}/{4 * len(tail_watch)} greedy tokens identical")
        o16.llm.act_quant = False
        mx_note = (f" [fp8 matrix cores, MXFP8 activations (opt-in): oracle act_quant, tie rule {tie:g} ulps, worst flip {worst_behind:.1f} ulps behind; "
                   f"vs the bf16-ACTIVATION oracle worst {worst_b16:.2e}, asserted <= {MX_BOUND}]") if mx else ""
        print(f"batched {name}{' fp8' if weight_format == 'fp8' else ''}{mx_note}, {batch_slots} slots, step kinds {kinds}: prefill logits vs fp32: "
              f"device {e_dev:.2e} oracle {e_orc:.2e}; slots {tuple(watch)}: step logits dev-vs-bf16-oracle worst {worst_r16:.2e}, worst "
              f"ratio to the fp32 envelope {worst_ratio:.2f}; greedy {identical}/{n_greedy_total} identical ({near_ties} flips to the runner-up in {near_tie_steps} near-tie steps, "
              f"worst {worst_behind:.2f} ulps behind, {wide_flips} beyond {tie:g}); "
              f"{draws_checked} sampled draws exact; oracle top-2 gap histogram (bf16 ulps of the top logit, "
              f"{len(gaps)} steps): {histogram(gaps)}{tail_report}; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()


R64, R32, R16 = range(64), range(32), range(16)


@pytest.mark.parametrize("name,weight_format,act_fp8", [("detikzify-ds-7b", "bf16", None), ("detikzify-cl-7b", "fp8", None), ("detikzify-cl-7b", "fp8", 1)])
def test_batched_headline_matches_cpu_oracle(name, weight_format, act_fp8):
    """The batched decode step exactly as `bench.py`'s rollouts/sec phases run it — full depth, 65 slots allocated (64 decoding
    + the prefix-cache slot), every default — against the CPU oracle, at EVERY column-tile count: 64 active slots (4 tiles:
    k_gemv_bl / k_gemv_bkl / k_resid_norm_b, the 64-slot kernels), then 32 (2 tiles: k_gemv_b<.., NT = 2>, the shape of BASELINE
    config 5 at N = 8: 1 image x 32 rollouts per rank, reference examples/eval.py:80-83), then 16 (1 tile: config 4 at N = 1 and
    config 5's ragged tail).  A step's kernels depend only on its highest active slot (csrc/dtk_api.hip: nt_step), asserted per
    step; slots 0 and 9 are followed through all three shapes.  ds-7b continues with 500 sampled tokens per slot and compares
    4 more steps there (>= 500 private keys per slot: k_attn_tail_b's steady state)."""
    ds = name == "detikzify-ds-7b"
    _phases_vs_oracle(name, weight_format, 65,
                      phases=[(R64, 4, 4, 64), (R32, 2, 2, 32), (R16, 2, 2, 16)],
                      watch=(0, 9, 40) if act_fp8 else (0, 9, 17, 40, 63), watch32=(9,), max_positions=1024 if ds else 512,      # (the opt-in MXFP8 run reads every watched slot twice: three are enough there)
                      private_tail=500 if ds else 0, tail_watch=(40,) if ds else (), act_fp8=act_fp8)


@pytest.mark.parametrize("name,weight_format", [("detikzify-ds-7b", "bf16"), ("detikzify-cl-7b", "fp8")])
def test_few_slot_contexts_match_cpu_oracle(name, weight_format):
    """Contexts of at most 5 slots decode with the multi-vector kernels (csrc/kernels_decode_mv.hip: the single-sequence GEMVs
    carrying 1 / 2 / 4 vectors) — the per-rank shape of BASELINE config 4 at N = 4 / 8 (4 / 2 trees per rank).  Full depth against
    the CPU oracle: 4 vectors, then 2, then 1 (slot 0 is followed through all three)."""
    _phases_vs_oracle(name, weight_format, 5,
                      phases=[(range(4), 3, 3, 4), (range(2), 2, 2, 2), (range(1), 1, 2, 1)],
                      watch=(0, 1, 3), watch32=(0,))


def test_v2_8b_batched_matches_cpu_oracle():
    """The repo's default family (reference detikzify/model/modeling_detikzify.py:119-271: LLaMA-3.1-8B decoder, GQA 32 / 8,
    128 256-token vocabulary) at 64 slots, full depth: the GQA-fused attention blocks and the multi-block sampler (7 kernels x
    64 slots) against the CPU oracle — every sampled draw of every slot exact."""
    _phases_vs_oracle("detikzify-v2-8b", "bf16", 65, phases=[(R64, 3, 3, 64)], watch=(0, 40), watch32=(40,))


N_LONG = 8       # decode steps per long-context checkpoint (round 3: 3)
LONG_CONTEXT_KNOWN_WIDE = 2.5      # bf16 ulps: the one measured flip beyond the 2-ulp rule (2.30, round 5), see the test


def test_long_context_steps_match_cpu_oracle():
    """ds-7b decode steps at contexts ~700 (where bench.py's 512-token rollouts end) and ~1900 (the API allows 2048) against
    the CPU oracle: the prompt is the image prefix + seeded text tokens; the oracle extends the shared prefix prefill ONCE per
    checkpoint (KV kept across the two), the device prefills to the checkpoint and decodes 8 greedy steps there (the attention
    kernel walks 700 / 1900 keys of KV the prefill GEMMs wrote); the 8 tokens are teacher-forced in one oracle pass.  Same logits
    envelope as the short-context tests, tokens by the near-tie rule."""
    from detikzify_amd.model import load
    from tests.fullsize import host_side
    t_start = time.perf_counter()
    model, proc = load("detikzify-ds-7b", synthetic=1234, max_positions=2048)
    try:
        hs = host_side(model, proc, "detikzify-ds-7b", "bf16")
        cfg, w, o16, o32 = hs.oracles(model)
        ids, px = hs.ids, hs.px
        img_tok, eos = cfg["image_token_id"], 2
        g = torch.Generator().manual_seed(7)
        text = torch.randint(3, cfg["vocab"] - 1, (1900,), generator=g)
        text[text == img_tok] = 5
        full = torch.cat([ids, text])
        model.set_sampling(do_sample=False, bad_ids=[img_tok])
        report, done, wide = [], ids.numel(), 0  # both oracles hold the image prefix (tests/fullsize.py)
        for T in (700, 1900):
            prompt = full[:T]
            dev = model.prefill(prompt, px, return_logits=True)
            for o in (o16, o32):                 # extend both oracles from where the previous checkpoint's prompt ended
                o.llm.truncate(done)
            ref, truth = o16.extend(prompt[done:], last_only=True), o32.extend(prompt[done:], last_only=True)
            done = T
            e_dev, e_orc = rel_l2(dev, truth), rel_l2(ref, truth)
            assert e_dev < ENVELOPE * e_orc + SLACK_LOGITS, (T, e_dev, e_orc)
            toks, dev_logits = [], []
            for i in range(N_LONG):
                model.decode_launch()
                toks.append(model.decode_wait())
                dev_logits.append(model.get_logits())
            rows16, rows32 = o16.extend(toks), o32.extend(toks)
            logits, worst, ties = ref, 0.0, 0
            for i, t in enumerate(toks):
                rt = sampling.greedy(logits, [img_tok], [], False)
                if rt != t:
                    # the near-tie rule of _phases_vs_oracle: a flip proves that the two logits' errors add up to the oracle's margin — within
                    # 2 bf16 ulps.  ONE named exception: round 5's driver-order run flipped a pair 2.30 ulps apart at the
                    # first checkpoint (context 701 of the 700-token prompt); a single flip there may reach LONG_CONTEXT_KNOWN_WIDE.
                    gap = top2_gap_ulps(logits, [img_tok], [], False)
                    assert gap <= (LONG_CONTEXT_KNOWN_WIDE if T == 700 else 2.0) + 1e-3, (T, i, t, rt, gap)
                    wide += gap > 2.0 + 1e-3
                    ties += 1
                logits, t32 = rows16[i], rows32[i]
                d, o = rel_l2(dev_logits[i], t32), rel_l2(logits, t32)
                worst = max(worst, d / (ENVELOPE * o + SLACK_LOGITS))
                assert d < ENVELOPE * o + SLACK_LOGITS, (T, i, d, o)
            assert model.context_len() == T + N_LONG
            report.append(f"context {T}: prefill logits vs fp32: device {e_dev:.2e} oracle {e_orc:.2e}; {N_LONG} decode steps worst ratio to the "
                          f"envelope {worst:.2f}, {N_LONG - ties}/{N_LONG} tokens identical")
        assert wide <= 1, f"{wide} token flips beyond 2 bf16 ulps"
        print("ds-7b long-context decode vs CPU oracle: " + "; ".join(report) + f"; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()


PEAKED_SEED, PEAKED_BETA = 0, 2.0      # the weight set (tests/helpers.py::peaked_lm_head)
PEAKED_PREFIX, PEAKED_CONTEXTS, PEAKED_WINDOW = 48, 64, 7
PEAKED_KNOWN_WIDE = 8.0      # bf16 ulps: at most PEAKED_MAX_WIDE judged positions per run may differ, each only where the oracle's own gap is below this.  The scale
PEAKED_MAX_WIDE = 2          # is the two pipelines' measured distance: 2.7-2.9e-2 rel-L2 between device and bf16-oracle logits at full depth (asserted by the envelope
                             # tests), carried by the rows this head scales by 2^8 = 7 ulps of those rows.  Measured flips: v2-8b position 78 at 2.69 ulps (round 5);
                             # with the sliced-K prefill GEMMs (another fp32 summation order of the prefix, same error against the fp32 oracle) cl-7b fp8
                             # positions 54 and 62 at 4.44 and 6.10 ulps, ds-7b one at 2-4; the rounds before happened to draw none above 2.7
PEAKED_VS_FP32_SLACK = 2     # positions by which the device may trail the bf16 oracle in agreeing with the fp32 oracle's greedy token (64 contexts per run)
PEAKED_MIN_JUDGED = 0.85     # of the 64 contexts the ORACLE's own top-2 gap must leave at >= 2 ulps (a property of the sequence, not of the device: ds-1.3b 55 of 64)


@pytest.mark.parametrize("name,weight_format", [("detikzify-ds-7b", "bf16"), ("detikzify-ds-1.3b", "bf16"), ("detikzify-cl-7b", "fp8"),
                                                ("detikzify-v2-8b", "bf16")])
def test_peaked_logits_weight_set_is_token_identical(name, weight_format):
    """A second synthetic weight set whose logits are PEAKED (lm_head rows scaled by log-normal powers of two, exact in bf16): the
    uniform set gives 32 k equal-variance logits — a top-2 gap below 2 bf16 ulps in ~25 % of the steps — and the near-tie rule of
    the other tests then forgives a mismatch.  Round 3 compared 16 plain greedy tokens here and the sequence fell into a two-token
    cycle after 5 steps (6 distinct contexts; sampling does not help either: the distribution is so peaked that T = 6 draws the same
    two tokens).  Now the decode is greedy UNDER A MOVING BAN: at every step the last 7 generated tokens are banned (bad_words_ids, the
    processor the reference itself configures: detikzify/infer/generate.py:218-227), so no token returns within 8 steps and every
    context is new — nothing is searched for, the only seed is the weight set's.  48 tokens of run-in, then at each of the next 64
    positions (64 distinct contexts, asserted) the device's token must be the argmax of the CPU oracle's logits under the same ban,
    the oracle reading the whole sequence in one pass.  Every BASELINE model at full depth (round 6: ds-1.3b = config 2, cl-7b with
    fp8 weights = config 5 and v2-8b as well as ds-7b), once on the single-sequence graph and once in slot 37 of a 64-slot batched
    step whose neighbours sample.  Positions where the oracle's own top-2 gap is below 2 ulps are reported and excluded; at least
    85 % must remain.  This is north_star's "token-identical under greedy decode", literally, on a head where it CAN hold.  Beside it the
    symmetric statement that needs no tie rule: over ALL 64 contexts of a run the device's token differs from the fp32 oracle's greedy token
    no more often (+ PEAKED_VS_FP32_SLACK) than the bf16 oracle's own argmax does — two bf16 roundings of the same arithmetic."""
    from detikzify_amd.model import load
    from tests.helpers import peaked_lm_head
    t_start = time.perf_counter()
    model, proc = load(name, synthetic=1234, max_positions=512, batch_slots=65, weight_format=weight_format)
    try:
        from tests.fullsize import host_side
        hs = host_side(model, proc, name, weight_format)       # BEFORE the head is replaced: the shared entry holds the seed-1234 weights
        cfg = model.config.oracle_dict()
        head = peaked_lm_head(hs.w["lm_head.weight"], PEAKED_BETA, PEAKED_SEED)
        model.load_tensor("lm_head.weight", head.to(torch.bfloat16))
        assert torch.equal(model.read_tensor("lm_head.weight").float().reshape(head.shape), head)
        ids, px = hs.ids, hs.px
        n_img, img_tok, N = ids.numel(), cfg["image_token_id"], PEAKED_PREFIX + PEAKED_CONTEXTS
        bans_at = lambda toks, k: [img_tok] + toks[max(0, k - PEAKED_WINDOW):k]
        runs = []           # (label, tokens)
        toks = []
        model.set_sampling(do_sample=False, bad_ids=[img_tok])
        model.prefill(ids, px)
        for k in range(N):
            model.set_sampling(do_sample=False, bad_ids=bans_at(toks, k))
            model.decode_launch()
            toks.append(model.decode_wait())
        runs.append(("single sequence", toks))
        model.set_sampling(do_sample=False, slot=64)
        model.prefill(ids, px, slot=64)
        for s_ in range(64):
            model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=99 + s_, bad_ids=[img_tok], slot=s_)
            model.kv_fork(64, s_, n_img)
        toks = []
        for k in range(N):
            model.set_sampling(do_sample=False, bad_ids=bans_at(toks, k), slot=37)
            model.decode_batch_launch(range(64))
            toks.append(model.decode_batch_wait()[37])
        if toks != runs[0][1] or name in ("detikzify-ds-7b", "detikzify-cl-7b"):      # (a sequence the oracle has judged already is not judged twice: one oracle pass less for the smaller runs)
            runs.append(("slot 37 of 64", toks))
        else:
            print(f"{name}: slot 37 of a 64-slot step produced the single-sequence run's {N} tokens exactly")

        _, _, o16, o32 = hs.oracles(model, override={"lm_head.weight": head}, fp32=True)    # the prefix KV does not depend on the head
        snap, snap32 = oracle_snapshot(o16), oracle_snapshot(o32)
        report, all_gaps = [], []
        for label, toks in runs:
            oracle_restore(o16, snap)
            h = o16.llm.forward(o16.llm.embed(torch.tensor(toks, dtype=torch.long)))
            oracle_restore(o32, snap32)
            h32 = o32.llm.forward(o32.llm.embed(torch.tensor(toks, dtype=torch.long)))
            d_dev = d_orc = 0          # positions where the device's token / the bf16 oracle's argmax is not the fp32 oracle's argmax
            # 64 distinct contexts by construction; the ban window (7 = what dtk_sampling's 8 bad ids leave next to the image token)
            # guarantees 8 distinct tokens among them, the peaked head gives few more (round 4: 11) — that is the point of the set
            assert len({tuple(toks[:k]) for k in range(PEAKED_PREFIX, N)}) == PEAKED_CONTEXTS and len(set(toks[PEAKED_PREFIX:])) > PEAKED_WINDOW, label
            same = judged = wide = 0
            for k in range(PEAKED_PREFIX, N):        # position k: context = image + toks[:k]; oracle logits from the state after toks[k - 1]
                ref = o16.llm.logits(h[k - 1])
                bans = bans_at(toks, k)
                a32 = sampling.greedy(o32.llm.logits(h32[k - 1]), bans, [], False)
                d_dev += toks[k] != a32
                d_orc += sampling.greedy(ref, bans, [], False) != a32
                gap = top2_gap_ulps(ref, bans, [], False)
                all_gaps.append(gap)
                if gap < 2.0:
                    continue
                judged += 1
                a_orc = sampling.greedy(ref, bans, [], False)
                if toks[k] != a_orc:
                    # the peaked rows that win are the ones scaled by 2^8: their logit carries the SAME relative error as any other
                    # (a few per cent of the dot product = several ulps), so two of them a few ulps apart can still change places — at most
                    # PEAKED_MAX_WIDE times per run, and only below PEAKED_KNOWN_WIDE
                    assert gap < PEAKED_KNOWN_WIDE and wide < PEAKED_MAX_WIDE, (label, k, toks[k], a_orc, gap)
                    wide += 1
                    continue
                same += 1
            assert same + wide == judged
            assert judged >= PEAKED_MIN_JUDGED * PEAKED_CONTEXTS, (label, judged)
            # the symmetric statement, no tie rule in it: both pipelines round the same fp32 arithmetic, so against the fp32 oracle's greedy
            # token (all 64 contexts, near-ties included) the device must not be wrong more often than the bf16 oracle is
            assert d_dev <= d_orc + PEAKED_VS_FP32_SLACK, (label, d_dev, d_orc)
            report.append(f"{label}: vs the fp32 oracle's token the device differs at {d_dev}, the bf16 oracle at {d_orc} of {PEAKED_CONTEXTS}; {same}/{judged} tokens identical ({PEAKED_CONTEXTS - judged} positions below 2 ulps excluded, {wide} flip(s) at 2-8 ulps, {len(set(toks[PEAKED_PREFIX:]))} distinct tokens)")
        print(f"{name}{' fp8' if weight_format == 'fp8' else ''}: peaked weight set (lm_head rows x 2^round({PEAKED_BETA} z)), greedy under a moving ban of the last {PEAKED_WINDOW} tokens, {PEAKED_PREFIX} + {PEAKED_CONTEXTS} tokens: "
              + "; ".join(report) + f"; oracle top-2 gap histogram ({len(all_gaps)} contexts): {histogram(all_gaps)}; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()


def test_greedy_margins_are_not_biased_against_the_oracle():
    """Is the device's rounding one-sided?  ds-7b width at 4 layers (a cheap oracle), 64-slot batched step, 4 watched slots x 64
    sampled steps = 256 distinct contexts.  At every step the oracle's top-2 pair (a, b) is looked up in the device's logits of the
    same context: margin_dev = dev[a] - dev[b] against margin_orc = orc[a] - orc[b] > 0.  Two correct bf16 pipelines scatter
    around each other, so the sign of (margin_dev - margin_orc) must be balanced (|n+ - n-| <= 4 sqrt(n): a pipeline that
    truncated where the reference rounds, or dropped a rounding point, would push every margin the same way), and the argmax
    may differ only where margin_orc is within 2 bf16 ulps — at most half of those."""
    from detikzify_amd.model.config import preset
    from detikzify_amd.model.modeling import DetikzifyForCausalLM
    t_start = time.perf_counter()
    cfg_dev = preset("detikzify-ds-7b")
    cfg_dev.layers, cfg_dev.max_positions, cfg_dev.batch_slots = 4, 256, 65
    model = DetikzifyForCausalLM(cfg_dev, 0)
    try:
        model.fill_synthetic(4321)
        cfg = model.config.oracle_dict()
        w = weights_from_device(model, cfg, skip_prefix="vision_model.")
        img_tok = cfg["image_token_id"]
        g = torch.Generator().manual_seed(3)
        ids = torch.randint(3, cfg["vocab"] - 1, (48,), generator=g)
        ids = ids[ids != img_tok]
        watch, STEPS = (0, 21, 42, 63), 64
        model.set_sampling(do_sample=False, slot=64)
        dev0 = model.prefill(ids, None, slot=64, return_logits=True)
        for s_ in range(64):
            model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, seed=700 + s_, bad_ids=[img_tok], slot=s_)
            model.kv_fork(64, s_, ids.numel())
        # the device first (64 steps, tokens and logits of the watched slots kept), then each watched slot's tokens teacher-forced
        # through the oracle in ONE pass: orc_rows[s][k] = the oracle's logits for the context the device held before step k
        dev_rows = {s_: [dev0] for s_ in watch}
        toks = {s_: [] for s_ in watch}
        for _ in range(STEPS):
            model.decode_batch_launch(range(64))
            out = model.decode_batch_wait()
            for s_ in watch:
                toks[s_].append(out[s_])
                dev_rows[s_].append(model.get_logits_slot(s_))
        o16 = DetikzifyOracle(cfg, w, precision="bf16")
        orc0 = o16.prefill(ids, None)
        snap = oracle_snapshot(o16)
        plus = minus = equal = flips = near = wide_flips = 0
        for s_ in watch:
            oracle_restore(o16, snap)
            orc_rows = [orc0] + list(o16.extend(toks[s_][:-1]))
            for k in range(STEPS):       # compare the logits both sides hold for the SAME context
                masked = sampling.mask_scores(orc_rows[k], [img_tok], [], False)
                top = torch.topk(masked, 2)
                a, b = top[1].tolist()
                m_orc = float(top[0][0] - top[0][1])
                m_dev = float(dev_rows[s_][k][a] - dev_rows[s_][k][b])
                gap = m_orc / (float(top[0][0].abs()) * ULP + 1e-30)
                near += gap <= 2.0 + 1e-3
                if m_dev < 0 or (m_dev == 0 and b < a):
                    # the 2-ulp rule, with ONE named exception per run below 3 ulps (measured with the sliced-K prefill GEMMs: slot 21, tokens 7294 / 3633, 2.48 ulps)
                    if gap > 2.0 + 1e-3:
                        assert gap <= 3.0 and wide_flips == 0, (s_, a, b, m_orc, m_dev, gap)
                        wide_flips += 1
                    flips += 1
                plus, minus, equal = plus + (m_dev > m_orc), minus + (m_dev < m_orc), equal + (m_dev == m_orc)
        n = plus + minus
        print(f"margin sign test, ds-7b width x 4 layers, 64-slot step, {len(watch) * STEPS} contexts: device margin above the oracle's {plus}, below {minus}, "
              f"equal {equal}; {flips} argmax flips in {near} near-tie contexts; {time.perf_counter() - t_start:.0f} s")
        assert n >= 128 and abs(plus - minus) <= 4.0 * n ** 0.5, (plus, minus)
        assert flips - wide_flips <= max(1, (near + 1) // 2), (flips, near)
    finally:
        del model
        gc.collect()


MX_BOUND = 0.20          # asserted: rel-L2 between the logits of a step with MXFP8 activations and the same step with bf16 activations, full depth
MX_CONTEXTS, MX_STEPS = 32, 16


def test_mxfp8_activations_against_bf16_activations():
    """What the opt-in fp8 matrix-core step (dtk_set_option act_fp8 = 1: MXFP8 activations, csrc/kernels_batch_mx.hip) costs against
    the default bf16-activation step of the SAME fp8-weight model — cl-7b at full depth, 65 slots, device against device on identical
    token sequences: 32 slots sample 16 tokens each (T = .8, top-p .95, the pipeline's defaults: detikzify/infer/generate.py:218-227)
    with bf16 activations on the uniform synthetic head; those 32 x 16 = 512 contexts are then teacher-forced (dtk_resume_slot forces
    the next token) through 32 other slots with MXFP8 activations, their KV cache built by the MXFP8 steps themselves.  The same 512
    contexts are then forced through BOTH modes under the PEAKED head (tests/helpers.py::peaked_lm_head: the uniform one puts the top
    two of 32 k logits within 2 bf16 ulps in a quarter of the steps, so its greedy agreement says little; sampling under the peaked
    head would not make the contexts differ — it draws the same token everywhere).  Per context: rel-L2 of the logits (ASSERTED
    <= MX_BOUND, SURVEY §7: "fp8 parity = bounded logit error"), greedy agreement, KL(T = .8) of the two next-token distributions,
    and whether the sampler's draw (same seed, same counter) picks the same token.  The figures are the reason the path is opt-in;
    DESIGN.md quotes them."""
    from detikzify_amd.model import load
    from tests.fullsize import host_side
    from tests.helpers import peaked_lm_head
    t_start = time.perf_counter()
    model, proc = load("detikzify-cl-7b", synthetic=1234, max_positions=512, weight_format="fp8", batch_slots=65)
    try:
        hs = host_side(model, proc, "detikzify-cl-7b", "fp8")
        cfg = model.config.oracle_dict()
        ids, px = hs.ids, hs.px
        n_img, img_tok, key = ids.numel(), cfg["image_token_id"], model.image_key(px)
        A, SRC = list(range(MX_CONTEXTS)), 64
        seed_of = lambda s: 300 + s

        def prefix():
            model.set_sampling(do_sample=False, slot=SRC)
            model.prefill(ids, px, slot=SRC)

        def forced(mode, base, toks):
            """the 512 contexts teacher-forced through slots base .. base + 31 with act_fp8 = mode: logits after every forced token"""
            model.set_option("act_fp8", mode)
            for s in A:
                model.set_sampling(do_sample=False, bad_ids=[img_tok], slot=base + s)
                model.kv_fork(SRC, base + s, n_img)
            out_logits = {s: [] for s in A}
            for k in range(MX_STEPS):
                for s in A:      # the slot holds prefix + toks[:k]; the next step is forced to emit toks[k] and runs the forward on it
                    model.resume_slot(base + s, torch.cat([ids, torch.tensor(toks[s][:k + 1], dtype=torch.long)]), key)
                model.decode_batch_launch([base + s for s in A])
                out = model.decode_batch_wait()
                st = model.stats()
                assert st["last_batch_step_fp8_mfma"] == mode and st["last_batch_step_slots"] == (64 if base else 32)
                for s in A:
                    assert out[base + s] == toks[s][k], (mode, s, k)
                    out_logits[s].append(model.get_logits_slot(base + s))
            return out_logits

        # the contexts: sampled with bf16 activations under the uniform head
        model.set_option("act_fp8", 0)
        prefix()
        for s in A:
            model.set_sampling(do_sample=True, temperature=0.8, top_p=0.95, top_k=0, seed=seed_of(s), bad_ids=[img_tok], slot=s)
            model.kv_fork(SRC, s, n_img)
        toks, free_run = {s: [] for s in A}, {s: [] for s in A}
        for k in range(MX_STEPS):
            model.decode_batch_launch(A)
            out = model.decode_batch_wait()
            assert model.stats()["last_batch_step_fp8_mfma"] == 0
            for s in A:
                toks[s].append(out[s])
                free_run[s].append(model.get_logits_slot(s))
        assert len({tuple(toks[s][:4]) for s in A}) > MX_CONTEXTS // 2, "the per-slot seeds did not make the contexts diverge"
        reports = []
        for label in ("uniform head", "peaked head"):
            if label == "uniform head":
                LA, LB = free_run, forced(1, 32, toks)
            else:
                model.load_tensor("lm_head.weight", peaked_lm_head(hs.w["lm_head.weight"], PEAKED_BETA, PEAKED_SEED).to(torch.bfloat16))
                prefix()
                LA = forced(0, 0, toks)
                LB = forced(1, 32, toks)
            worst = mean = kl_sum = kl_max = 0.0
            same_greedy = same_draw = n = 0
            for s in A:
                for k in range(MX_STEPS):
                    a, b = LA[s][k], LB[s][k]
                    r = rel_l2(b, a)
                    worst, mean = max(worst, r), mean + r
                    ma, mb = sampling.mask_scores(a, [img_tok], [], False), sampling.mask_scores(b, [img_tok], [], False)
                    same_greedy += int(torch.argmax(ma)) == int(torch.argmax(mb))
                    pa, lb = torch.softmax(ma.double() / 0.8, -1), torch.log_softmax(mb.double() / 0.8, -1)
                    live = pa > 0           # (banned ids: p = 0 and log q = -inf on both sides)
                    kl = float((pa[live] * (torch.log(pa[live]) - lb[live])).sum())
                    kl_sum, kl_max = kl_sum + kl, max(kl_max, kl)
                    da, _ = sampling.draw(a, 0.8, 0, 0.95, seed_of(s), k + 1, [img_tok], [], False)
                    db, _ = sampling.draw(b, 0.8, 0, 0.95, seed_of(s), k + 1, [img_tok], [], False)
                    same_draw += da == db
                    n += 1
            assert worst <= MX_BOUND, f"{label}: logits with MXFP8 activations are {worst:.3f} rel-L2 from the bf16-activation step (bound {MX_BOUND})"
            # (the draw agreement is reported, not asserted: the uniform head spreads T = .8 / top-p .95 over thousands of tokens of ~1e-4
            # mass each, so any perturbation of the CDF moves the draw — round 5 measured 54 / 512 there)
            reports.append(f"{label}: {n} contexts, logits rel-L2 mean {mean / n:.3e} worst {worst:.3e} (asserted <= {MX_BOUND}); greedy token identical "
                           f"{same_greedy}/{n} = {same_greedy / n:.3f}; KL(T=.8) mean {kl_sum / n:.3e} max {kl_max:.3e}; same sampled token under the same draw {same_draw}/{n} = {same_draw / n:.3f}")
        print("MXFP8 activations (act_fp8 = 1) against bf16 activations (default), cl-7b fp8 full depth, same token sequences: " + "; ".join(reports)
              + f"; {time.perf_counter() - t_start:.0f} s")
    finally:
        del model
        gc.collect()
