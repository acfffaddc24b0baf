"""Pins the CPU oracle against the golden vectors produced by tests/golden/make_golden.py (installed
HuggingFace Llama / SigLIP / logits processors, and the reference's own v2 model code).  CPU only."""
import numpy as np
import torch

from oracle import sampling
from oracle.llama import LlamaOracle
from oracle.model import DetikzifyOracle
from oracle.synth import make_weights
from oracle.vit import VitOracle
from tests.helpers import TINY_CFG, TINY_V2_CFG, rel_l2


def test_llama_fp32_matches_hf(golden_dir):
    g = np.load(golden_dir / "llama_tiny.npz")
    w = make_weights(TINY_CFG, 1234)
    llm = LlamaOracle(TINY_CFG, w, precision="fp32")
    h = llm.forward(torch.from_numpy(g["embeds"]))
    logits = llm.logits(h)
    assert rel_l2(logits, g["logits_embeds_fp32"]) < 2e-5          # same math, fp32 round-off only
    llm.reset()
    ids = torch.from_numpy(g["ids"][0])
    lo = llm.logits(llm.forward(llm.embed(ids))[-1])
    assert rel_l2(lo, g["logits_ids_fp32"]) < 2e-5


def test_llama_bf16_policy_close_to_hf_bf16(golden_dir):
    """the bf16 rounding policy tracks HF's own bf16 run (not bit-equal: HF eager/sdpa rounds the
    scores differently; documented in oracle/__init__.py)"""
    g = np.load(golden_dir / "llama_tiny.npz")
    w = make_weights(TINY_CFG, 1234)
    llm = LlamaOracle(TINY_CFG, w, precision="bf16")
    logits = llm.logits(llm.forward(torch.from_numpy(g["embeds"])))
    assert rel_l2(logits, g["logits_embeds_bf16"]) < 2e-2
    assert rel_l2(logits, g["logits_embeds_fp32"]) < 2e-2


def test_greedy_tokens_match_hf_generate(golden_dir):
    """HF generate() with bad_words_ids=[[1]], begin_suppress_tokens=[2] vs the oracle loop (fp32)"""
    g = np.load(golden_dir / "llama_tiny.npz")
    w = make_weights(TINY_CFG, 1234)
    o = DetikzifyOracle(TINY_CFG, w, precision="fp32")
    toks = o.generate(torch.from_numpy(g["ids"][0]), None, 24, bad=[1], begin=[2], eos=2)
    ref = g["greedy_fp32"].tolist()
    assert toks == ref[: len(toks)] and len(toks) >= min(24, len(ref))


def test_llama_gqa_llama3_rope_matches_hf(golden_dir):
    """the v2 decoder: GQA 4/2 + rope_type "llama3" vs HF LlamaForCausalLM, 90 positions with
    original_max_position 64 so every frequency band of the llama3 scaling is exercised"""
    from oracle.llama import llama3_inv_freq
    g = np.load(golden_dir / "llama_tiny_gqa.npz")
    c = TINY_V2_CFG
    inv = llama3_inv_freq(c["head_dim"], c["rope_theta"], c["rope_factor"], c["rope_low_freq_factor"],
                          c["rope_high_freq_factor"], c["rope_original_max_position"])
    assert np.allclose(inv.numpy(), g["inv_freq"], rtol=1e-6, atol=0)
    assert len({float(x) for x in (inv * (c["rope_theta"] ** (torch.arange(0, 128, 2).float() / 128)))}) > 3   # bands differ
    w = make_weights(c, 4321)
    llm = LlamaOracle(c, w, precision="fp32")
    logits = llm.logits(llm.forward(torch.from_numpy(g["embeds"])))
    assert rel_l2(logits, g["logits_embeds_fp32"]) < 2e-5
    llm16 = LlamaOracle(c, w, precision="bf16")
    l16 = llm16.logits(llm16.forward(torch.from_numpy(g["embeds"])))
    assert rel_l2(l16, g["logits_embeds_bf16"]) < 2e-2 and rel_l2(l16, g["logits_embeds_fp32"]) < 2e-2
    o = DetikzifyOracle(c, w, precision="fp32")
    toks = o.generate(torch.from_numpy(g["ids"][0]), None, 24, bad=[5], begin=[2], eos=2)
    ref = g["greedy_fp32"].tolist()
    assert toks == ref[: len(toks)] and len(toks) >= min(24, len(ref))


def test_vit_matches_hf_siglip(golden_dir):
    """timm-layout weights through the oracle == HF SiglipVisionModel through the name mapping"""
    g = np.load(golden_dir / "siglip_tiny.npz")
    w = make_weights(TINY_CFG, 1234, only_prefix="vision_model.")
    px = torch.from_numpy(g["pixels"])
    for tanh, tag in ((0, "erf"), (1, "tanh")):
        cfg = dict(TINY_CFG, vit_gelu_tanh=tanh)
        lh, pooled = VitOracle(cfg, w, precision="fp32").forward(px)
        assert rel_l2(lh, g[f"last_hidden_{tag}"]) < 2e-5
        assert rel_l2(pooled, g[f"pooled_{tag}"]) < 2e-5
        lh16, pooled16 = VitOracle(cfg, w, precision="bf16").forward(px)
        assert rel_l2(lh16, g[f"last_hidden_{tag}"]) < 2e-2
    # get_intermediate_layers(n=[last], norm=True) == forward_features
    v = VitOracle(TINY_CFG, w, precision="fp32")
    assert torch.equal(v.intermediate(px, TINY_CFG["vit_depth"] - 1), v.forward_features(px))


def test_logits_processors_match_hf(golden_dir):
    g = np.load(golden_dir / "processors.npz")
    logits = torch.from_numpy(g["logits"])
    for row, (T, k, p, first) in enumerate(g["cases"]):
        s = sampling.processed_scores(logits[row], temperature=float(T), top_k=int(k), top_p=float(p),
                                      bad=[1], begin=[2], first=bool(first))
        ref = torch.from_numpy(g[f"scores_{row}"])
        assert torch.equal(torch.isinf(s), torch.isinf(ref))
        keep = ~torch.isinf(ref)
        assert torch.allclose(s[keep], ref[keep], rtol=0, atol=0)


def test_deterministic_sampler_keeps_hf_nucleus(golden_dir):
    """the integer-mass sampler keeps exactly the HF top-k/top-p set (boundary ties aside) and
    draws reproducibly from it"""
    g = np.load(golden_dir / "processors.npz")
    logits = torch.from_numpy(g["logits"])
    for row, (T, k, p, first) in enumerate(g["cases"]):
        kw = dict(bad=[1], begin=[2], first=bool(first))
        z, q = sampling.integer_masses(logits[row], float(T), **kw)
        keep = sampling.kept_mask(z, q, int(k), float(p))
        ref_keep = ~torch.isinf(torch.from_numpy(g[f"scores_{row}"]))
        assert int((keep ^ ref_keep).sum()) <= 1
        toks = [sampling.draw(logits[row], float(T), int(k), float(p), seed=42, n=n, **kw)[0] for n in range(64)]
        assert all(bool(keep[t]) for t in toks)
        assert toks == [sampling.draw(logits[row], float(T), int(k), float(p), seed=42, n=n, **kw)[0] for n in range(64)]
        assert len(set(toks)) > 4


def test_synth_weights_are_bf16_representable():
    w = make_weights(TINY_CFG, 1234)
    for k, v in w.items():
        assert torch.equal(v, v.to(torch.bfloat16).float()), k
    assert abs(float(w["lm_head.weight"].std()) - 0.02) < 2e-3


def test_oracle_matches_the_reference_v2_model(golden_dir):
    """tests/golden/reference_v2_tiny.npz comes from the reference's OWN detikzify/model/modeling_detikzify.py
    (DetikzifyForConditionalGeneration: SigLIP tower -> 3-patch concat -> bias-free connector -> splice -> LLaMA with
    GQA + llama3 rope -> lm_head) run on the seeded synthetic weights in fp32: prefill logits at every position of the
    prompt's tail and 16 greedy steps through its KV cache.  The oracle on the same weights: logits within fp32
    round-off, tokens identical — the whole image-conditioned path, not just its HF building blocks."""
    from tests.golden.make_golden import REFERENCE_V2_SEED, REFERENCE_V2_STEPS
    g = np.load(golden_dir / "reference_v2_tiny.npz")
    ids, px = torch.from_numpy(g["ids"]), torch.from_numpy(g["pixels"])
    w = make_weights(TINY_V2_CFG, REFERENCE_V2_SEED)
    o = DetikzifyOracle(TINY_V2_CFG, w, precision="fp32")
    assert rel_l2(o.prefill(ids, px[0]), g["prefill_logits"][-1]) < 2e-5
    for cut in (13, 14):       # shorter prompts = earlier positions of the reference's prefill
        assert rel_l2(DetikzifyOracle(TINY_V2_CFG, w, precision="fp32").prefill(ids[:cut], px[0]), g["prefill_logits"][cut - 1]) < 2e-5
    toks, logits = DetikzifyOracle(TINY_V2_CFG, w, precision="fp32").generate(
        ids, px[0], REFERENCE_V2_STEPS, bad=[TINY_V2_CFG["image_token_id"]], begin=[2], return_logits=True)
    assert toks == g["tokens"].tolist()
    assert max(rel_l2(a, b) for a, b in zip(logits, g["step_logits"])) < 2e-5
    # the bf16 policy (what the device computes) stays within bf16 round-off of the reference
    o16 = DetikzifyOracle(TINY_V2_CFG, w, precision="bf16")
    assert rel_l2(o16.prefill(ids, px[0]), g["prefill_logits"][-1]) < 2e-2


def test_oracle_matches_the_reference_v1_model(golden_dir):
    """tests/golden/reference_v1_tiny.npz comes from the reference's OWN detikzify/model/v1/modeling_detikzify.py
    (DetikzifyForCausalLM: 3 consecutive patch features concatenated, mm_projector with bias, validated splice over the
    image-token run, LLaMA with linear rope scaling, lm_head; tower = HF SigLIP behind a timm-shaped shim): prefill
    logits, 16 greedy steps through its KV cache, and the two ValueErrors it raises for a bad image-token layout."""
    import pytest

    from tests.golden.make_golden import REFERENCE_V1_SEED
    g = np.load(golden_dir / "reference_v1_tiny.npz")
    ids, px = torch.from_numpy(g["ids"]), torch.from_numpy(g["pixels"])
    w = make_weights(TINY_CFG, REFERENCE_V1_SEED)
    tok = TINY_CFG["image_token_id"]
    for cut in (13, 14, 15):
        assert rel_l2(DetikzifyOracle(TINY_CFG, w, precision="fp32").prefill(ids[:cut], px[0]), g["prefill_logits"][cut - 1]) < 2e-5
    toks, logits = DetikzifyOracle(TINY_CFG, w, precision="fp32").generate(ids, px[0], len(g["tokens"]), bad=[tok], begin=[2],
                                                                           return_logits=True)
    assert toks == g["tokens"].tolist()
    assert max(rel_l2(a, b) for a, b in zip(logits, g["step_logits"])) < 2e-5
    assert rel_l2(DetikzifyOracle(TINY_CFG, w, precision="bf16").prefill(ids, px[0]), g["prefill_logits"][-1]) < 2e-2
    o = DetikzifyOracle(TINY_CFG, w, precision="fp32")
    with pytest.raises(ValueError) as e:
        o.prefill(torch.tensor([tok] * 11 + [7, 9]), px[0])
    assert str(e.value) == str(g["error_count"])
    with pytest.raises(ValueError) as e:
        o.prefill(torch.tensor([tok] * 6 + [7] + [tok] * 6), px[0])
    assert str(e.value) == str(g["error_gap"])


def test_real_size_tower_matches_hf_siglip():
    """the oracle's ViT at the BASELINE size (so400m/14 @ 384: 27 blocks, 729 tokens, D 1152, the 'map' pooling head) on
    seeded synthetic weights against the installed HF SiglipVisionModel through the timm -> HF name mapping, fp32: the
    toy-size golden (siglip_tiny.npz) says nothing about depth-27 accumulation or the 729-token attention.  ~30 s."""
    from transformers import SiglipVisionConfig, SiglipVisionModel

    from detikzify_amd.model.config import preset
    from tests.golden.make_golden import timm_to_hf_siglip
    cfg = preset("detikzify-ds-1.3b").oracle_dict()
    assert (cfg["vit_depth"], cfg["vit_dim"], cfg["vit_image"]) == (27, 1152, 384)
    w = make_weights(cfg, 1234, only_prefix="vision_model.")
    with torch.device("meta"):          # no random initialisation of 428 M parameters: the weights are assigned below
        hf = SiglipVisionModel(SiglipVisionConfig(
            hidden_size=cfg["vit_dim"], intermediate_size=cfg["vit_mlp"], num_hidden_layers=cfg["vit_depth"],
            num_attention_heads=cfg["vit_heads"], image_size=cfg["vit_image"], patch_size=cfg["vit_patch"],
            layer_norm_eps=cfg["vit_ln_eps"], hidden_act="gelu")).eval()
    sd = timm_to_hf_siglip(w, cfg)
    if not any(k.startswith("vision_model.") for k in hf.state_dict()):       # transformers >= 5 drops the prefix
        sd = {k[len("vision_model."):]: v for k, v in sd.items()}
    hf.load_state_dict({k: v.float().contiguous() for k, v in sd.items()}, strict=True, assign=True)
    for name, buf in list(hf.named_buffers()):                                 # position ids: the one non-persistent buffer
        if buf.is_meta:
            mod = hf.get_submodule(name.rsplit(".", 1)[0]) if "." in name else hf
            setattr(mod, name.rsplit(".", 1)[-1], torch.arange(buf.numel()).reshape(buf.shape))
    px = torch.randn(1, 3, cfg["vit_image"], cfg["vit_image"], generator=torch.Generator().manual_seed(3)).clamp(-1, 1)
    with torch.no_grad():
        want = hf(pixel_values=px)
    v = VitOracle(cfg, w, precision="fp32")
    feats = v.forward_features(px[0])
    assert feats.shape == (729, 1152)
    assert rel_l2(feats, want.last_hidden_state[0]) < 2e-5 and rel_l2(v.forward_head(feats), want.pooler_output[0]) < 2e-5


def test_real_width_decoder_matches_hf_llama():
    """the oracle's LLaMA at the ds-1.3b WIDTH (hidden 2048, 16 heads of 128, ffn 5504, vocab 32256, rope theta 1e5 with
    linear factor 4, rms eps 1e-6) but 2 layers, on seeded synthetic weights against the installed HF LlamaForCausalLM in
    fp32: 40 prompt positions and 3 cached decode steps.  The toy goldens (hidden 256) say nothing about K = 2048 / 5504
    accumulations or the real rope parameters.  ~25 s."""
    from transformers import LlamaConfig, LlamaForCausalLM

    from detikzify_amd.model.config import preset
    cfg = dict(preset("detikzify-ds-1.3b").oracle_dict(), layers=2)
    w = make_weights(cfg, 77, only_prefix="model.")
    w.update(make_weights(cfg, 77, only_prefix="lm_head"))
    hf = LlamaForCausalLM(LlamaConfig(
        hidden_size=cfg["hidden"], intermediate_size=cfg["ffn"], num_hidden_layers=2, num_attention_heads=cfg["heads"],
        num_key_value_heads=cfg["heads"], head_dim=cfg["head_dim"], vocab_size=cfg["vocab"], rms_norm_eps=cfg["rms_eps"],
        max_position_embeddings=cfg["max_positions"], rope_theta=cfg["rope_theta"],
        rope_scaling={"rope_type": "linear", "factor": cfg["rope_factor"]}, attention_bias=False,
        tie_word_embeddings=False, bos_token_id=1, eos_token_id=2, pad_token_id=0)).eval()
    sd = {k: v.float().contiguous() for k, v in w.items() if k in hf.state_dict()}
    assert set(sd) == set(hf.state_dict())
    hf.load_state_dict(sd, strict=True, assign=True)
    ids = torch.randint(3, cfg["vocab"], (1, 40), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = hf(input_ids=ids, use_cache=True)
        llm = LlamaOracle(cfg, w, precision="fp32")
        mine = llm.logits(llm.forward(llm.embed(ids[0])))
        assert rel_l2(mine, out.logits[0]) < 2e-5
        for tok in (17, 4096, 31999):
            out = hf(input_ids=torch.tensor([[tok]]), past_key_values=out.past_key_values, use_cache=True)
            assert rel_l2(llm.logits(llm.forward(llm.embed(torch.tensor([tok])))[-1]), out.logits[0, -1]) < 2e-5


def test_one_pass_teacher_forcing_is_the_stepwise_oracle():
    """DetikzifyOracle.extend (the GPU suite teacher-forces every watched slot in ONE causal pass since round 5) against the same
    tokens fed one step() at a time, from a restored snapshot: identical up to the order of the fp32 accumulation (GEMM rows vs
    GEMV) — fp32 policy to ~1e-6, bf16 policy to rounding flips of single logits, MXFP8-activation mode too; last_only = the
    last row; the cache ends at the same position either way."""
    import torch
    from oracle.model import DetikzifyOracle
    from oracle.synth import make_weights
    from tests.helpers import TINY_CFG, TINY_V2_CFG, rel_l2
    for cfg in (TINY_CFG, TINY_V2_CFG):
        w = make_weights(cfg, 1234)
        g = torch.Generator().manual_seed(11)
        ids = torch.randint(3, cfg["vocab"] - 1, (20,), generator=g)
        ids = ids[ids != cfg["image_token_id"]]
        toks = [5, 9, 100, 33, 7, 250, 8]
        for precision, bound in (("fp32", 2e-6), ("bf16", 5e-3)):
            for quant in (False, True):
                o = DetikzifyOracle(cfg, w, precision=precision)
                o.prefill(ids, None)
                snap = o.snapshot()
                o.llm.act_quant = quant
                steps = torch.stack([o.step(t) for t in toks])
                end = o.llm.pos
                o.restore(snap)
                rows = o.extend(toks)
                assert rows.shape == steps.shape and o.llm.pos == end == ids.numel() + len(toks)
                assert rel_l2(rows, steps) < bound, (precision, quant, rel_l2(rows, steps))
                o.restore(snap)
                assert rel_l2(o.extend(toks, last_only=True), rows[-1]) < bound          # (lm_head over one row instead of seven: GEMV vs GEMM order)
