"""
GPU parity tests (-m gpu) of the grouped shared-prefix attention of the batched decode step (round 5:
csrc/kernels_batch_decode.hip::k_attn_prefix_g + the group table of BatchState, csrc/dtk_api.hip::dtk_decode_batch_launch).

Semantics under test: the HF attention under the KV cache the reference runs per sequence
(/root/reference detikzify/model/v1/modeling_detikzify.py:191-200 via transformers LlamaAttention; rollouts of one image:
detikzify/infer/generate.py:246-282).  The rollouts of one image hold bit-identical copies of the image prefix (dtk_kv_fork), so
the scores against it are computed once per GROUP of <= 16 forks on the matrix cores; a slot's private keys follow per slot.

  * against the per-slot walk (prefix_mfma = 0) on the same contexts: identical greedy tokens, logits within 1e-2 rel-L2 (fp32
    summation order + the bf16 hi/lo split of the probabilities), slots without a shared prefix bit-identical;
  * a slot's logits are BIT-IDENTICAL whether it decodes alone, in its own group, or next to 60 other slots of three images —
    grouping is by (share_src, share_len), properties of the slot alone;
  * sources with more than 16 forks (several groups), a singleton group, a source that decodes itself (not a member), more
    images than DTK_PFX_GROUPS allows (the overflow takes the per-slot walk), every key-split count, GQA (tiny-v2);
  * against the CPU oracle: the full-size tests of tests/test_gpu_parity_batched.py run this path by default (65-slot contexts).
The path is the default of contexts with 64 decoding slots (dtk_create: prefix kernel on, 2-wave tail blocks); smaller contexts
walk every slot's whole context in 4-wave blocks (profiles/r05g_step_bench.txt has the measurement behind that split).
"""
import gc

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.helpers import rel_l2, sketch_image

N_GROUPS = 16        # DTK_PFX_GRID (csrc/common.h): group rows of the prefix kernel's grid


ULP = 2.0 ** -7      # one bf16 ulp relative to the value's binade top
EXTRA = {0: 141, 1: 60, 18: 150, 19: 101}          # text tokens behind the image tokens of prompt k: prefixes of 153 / 72 keys (3 / 2 tiles of 64, none a
                                                     # multiple); prompts 18 / 19 (the 19th / 20th group of the overflow test): 162 / 113 keys


def _load(name, slots=65):
    from detikzify_amd.model import load
    model, proc = load(name, synthetic=1234, batch_slots=slots, max_positions=320)
    return model, proc


def _image_prompt(proc, k, size, vocab):
    enc = proc(images=sketch_image(20 + k, size), return_tensors="pt")
    g = torch.Generator().manual_seed(100 + k)
    extra = torch.randint(6, vocab - 1, (EXTRA.get(k, 1 + k % 3),), generator=g)
    return torch.cat([enc.input_ids[0], extra]), enc.pixel_values


def _setup(model, proc, size, layout, img_tok):
    """layout: list of (source slot, prompt index, [fork slots]); the sources are prefilled with an image prompt + text tokens
    (prefix lengths that are no multiple of the 64-key tile: 153, 72, 13-15 keys), the forks share the whole prompt"""
    for s in range(model.num_slots()):
        model.set_sampling(do_sample=False, bad_ids=[img_tok], slot=s)
    for src, k, forks in layout:
        ids, px = _image_prompt(proc, k, size, model.config.vocab)
        model.prefill(ids, px, slot=src)
        for dst in forks:
            model.kv_fork(src, dst, ids.numel())


def _decode(model, active, steps, watch):
    """(tokens[step][i], logits[slot][step]) of the watched slots: logits AFTER each step"""
    toks, logits = [], {s: [] for s in watch}
    for _ in range(steps):
        model.decode_batch_launch(active)
        out = model.decode_batch_wait()
        toks.append([out[s] for s in watch])
        for s in watch:
            logits[s].append(model.get_logits_slot(s))
    return toks, logits


def _compare(run, ref, watch, first_logits, tol, tag):
    """tokens identical and per-step logits within `tol` — per slot up to the first step where the REFERENCE itself was within
    2 bf16 ulps of a tie (two correct pipelines may order such a pair either way; the contexts differ from there on)"""
    (t, lg), (rt, rlg) = run, ref
    worst, compared = 0.0, 0
    for i, s in enumerate(watch):
        prev = first_logits.get(s)
        for k in range(len(rt)):
            if t[k][i] != rt[k][i]:
                assert prev is not None, (tag, s, k)
                top2 = torch.topk(prev, 2)[0]
                assert float(top2[0] - top2[1]) <= 2.0 * float(top2[0].abs()) * ULP + 1e-6, (tag, s, k, t[k][i], rt[k][i])
                break
            r = rel_l2(lg[s][k], rlg[s][k])
            worst, compared = max(worst, r), compared + 1
            assert r < tol, (tag, s, k, r)
            prev = rlg[s][k]
    assert compared >= len(watch) * len(rt) // 2, (tag, compared)
    return worst


def _own_prompt(model, proc, size, slot):
    ids, px = _image_prompt(proc, 3, size, model.config.vocab)
    model.prefill(ids, px, slot=slot)


@pytest.mark.parametrize("name,size", [("detikzify-tiny", 96), ("detikzify-tiny-v2", 84)])
def test_grouped_prefix_attention_tracks_the_per_slot_walk(name, size):
    """64 decoding slots: image A (a 153-key prefix) has 40 forks (groups of 16, 16 and 8), image B (72 keys) 10, image C ONE fork
    (a singleton group); the three sources (slots 60-62) decode too and slot 63 is prefilled on its own — none of those four sees
    the prefix kernel.  12 greedy steps (the private part of every context grows behind the prefix), every key-split count,
    against prefix_mfma = 0."""
    model, proc = _load(name)
    try:
        img_tok = model.config.image_token_id
        A, B, C = list(range(0, 40)), list(range(40, 50)), [50]
        layout = [(60, 0, A), (61, 1, B), (62, 2, C)]
        active = A + B + C + [60, 61, 62, 63]
        watch = [0, 15, 16, 39, 40, 49, 50, 60, 62, 63]
        runs, first = {}, {}
        for prefix_on, splits in ((0, 4), (1, 1), (1, 2), (1, 3), (1, 4)):
            model.set_option("prefix_mfma", prefix_on)
            model.set_option("pfx_splits", splits)
            _setup(model, proc, size, layout, img_tok)
            _own_prompt(model, proc, size, 63)
            if not prefix_on:
                first = {s: model.get_logits_slot(s) for s in watch}       # the logits the first step's tokens are chosen from
            runs[(prefix_on, splits)] = _decode(model, active, 12, watch)
            assert model.stats()["last_batch_step_slots"] == 64
        ref = runs[(0, 4)]
        worst = 0.0
        for key, run in runs.items():
            if not key[0]:
                continue
            worst = max(worst, _compare(run, ref, watch, first, 1e-2, key))
            for s in (60, 62, 63):       # a source that decodes and a slot without a shared prefix never see the prefix kernel
                assert run[0] is not None and all(torch.equal(x, y) for x, y in zip(run[1][s], ref[1][s])), (key, s)
        # forks of one source hold the same context: the same tokens whichever group (and MFMA column) they sit in
        for key, (t, _) in runs.items():
            assert all(row[0] == row[1] == row[2] == row[3] for row in t) and all(row[4] == row[5] for row in t), key
        print(f"grouped prefix attention, {name}: 64 slots of 3 images (prefixes of 153 / 72 / 13-15 keys), 1-4 key splits: tokens as the per-slot walk, "
              f"logits worst rel-L2 {worst:.2e}")
    finally:
        del model
        gc.collect()


@pytest.mark.parametrize("name,size", [("detikzify-tiny", 96), ("detikzify-tiny-v2", 84)])
def test_a_slots_result_does_not_depend_on_its_company(name, size):
    """prefix_mfma = 1 (the default): fork 5 of image A decoded ALONE, then with the other forks of its image, then with all 64
    slots of three images — bit-identical logits and tokens at every step (its group is decided by its own (share_src, share_len);
    its MFMA column does not see the other columns; the step's tile count does not enter)."""
    model, proc = _load(name)
    try:
        img_tok = model.config.image_token_id
        A, B, C = list(range(0, 40)), list(range(40, 50)), [50]
        layout = [(60, 0, A), (61, 1, B), (62, 2, C)]
        model.set_option("prefix_mfma", 1)
        res = []
        for active in ([5], A, A + B + C + [60, 61, 62, 63]):
            _setup(model, proc, size, layout, img_tok)
            _own_prompt(model, proc, size, 63)
            res.append(_decode(model, active, 8, [5]))
        for t, lg in res[1:]:
            assert t == res[0][0]
            assert all(torch.equal(x, y) for x, y in zip(lg[5], res[0][1][5]))
    finally:
        del model
        gc.collect()


def test_more_prefixes_than_grid_rows_stay_on_the_matrix_cores():
    """20 images with two forks each in one step = 20 groups for a grid of DTK_PFX_GRID = 16 group rows: rows 0..3 take a second
    group (round 6; rounds 4-5 capped a step at 16 groups and let the rest walk their whole context per slot — so whether a slot's
    prefix went through the matrix cores depended on how many OTHER prefixes the step held: ADVICE r5).  Every slot's tokens as the
    per-slot path's, logits within 1e-2 of it; and a fork of the 20th image decoded ALONE (one group) is bit-identical to itself in
    the 40-slot step (group 19, the second trip of row 3)."""
    model, proc = _load("detikzify-tiny", slots=65)
    try:
        img_tok = model.config.image_token_id
        layout = [(40 + k, k, [2 * k, 2 * k + 1]) for k in range(20)]       # sources: slots 40..59 (they do not decode here)
        active = list(range(40))
        runs, first = {}, {}
        for prefix_on in (0, 1):
            model.set_option("prefix_mfma", prefix_on)
            _setup(model, proc, 96, layout, img_tok)
            if not prefix_on:
                first = {s: model.get_logits_slot(s) for s in active}
            runs[prefix_on] = _decode(model, active, 6, active)
        _compare(runs[1], runs[0], active, first, 1e-2, "20 groups")
        l0, l1 = runs[0][1], runs[1][1]
        # (groups 18 and 19 have prefixes of 162 / 113 keys: long enough for the two summation orders to differ in some logit of some step)
        assert any(not torch.equal(x, y) for s in range(36, 40) for x, y in zip(l1[s], l0[s])), "the forks beyond the 16th group did not take the grouped path"
        _setup(model, proc, 96, layout, img_tok)
        alone = _decode(model, [39], 6, [39])
        assert [t[39] for t in runs[1][0]] == [t[0] for t in alone[0]]
        assert all(torch.equal(x, y) for x, y in zip(alone[1][39], l1[39])), "slot 39's result depends on its company"
    finally:
        del model
        gc.collect()
