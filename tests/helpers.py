"""Shared test fixtures: tiny config, seeded sketch image, scripted fake model (no GPU)."""
from __future__ import annotations

import random
from types import SimpleNamespace
from typing import List

import numpy as np
import torch
from PIL import Image, ImageDraw

from detikzify_amd.model.config import preset
from detikzify_amd.model.processing import DetikzifyImageProcessor, DetikzifyProcessor
from detikzify_amd.model.tokenizer import SyntheticTokenizer
from detikzify_amd.util.synthetic import sketch_image  # noqa: F401  (the tests' seeded sketch: same input as bench.py / smoke())

TINY = preset("detikzify-tiny")
TINY_CFG = TINY.kernel_dict()
TINY_V2 = preset("detikzify-tiny-v2")          # GQA 4/2, rope "llama3", bias-free connector, tanh GELU, dedicated image token
TINY_V2_CFG = TINY_V2.oracle_dict()


def fake_processor(vocab: int = 512, image_seq_len: int = 12, image_size: int = 84) -> DetikzifyProcessor:
    tok = SyntheticTokenizer(vocab, bos_token_id=1, eos_token_id=2, pad_token_id=0, model_max_length=160)
    return DetikzifyProcessor(image_processor=DetikzifyImageProcessor(size={"height": image_size, "width": image_size}),
                              tokenizer=tok, image_seq_len=image_seq_len, image_token=tok.convert_ids_to_tokens(1))


class FakeModel:
    """Scripted stand-in for the (HIP) model with the attribute surface DetikzifyGenerator touches.
    generate() emits a deterministic pseudo-random continuation that depends only on the prompt and
    the call index, honours streamer / stopping criteria / max_length like HF generate."""

    def __init__(self, seed: int = 0, vocab: int = 512):
        self.seed, self.vocab, self.calls = seed, vocab, 0
        cfg = SimpleNamespace(image_token_id=1, eos_token_id=2, pooling_mode="cos")
        cfg.text_config = cfg
        self.config = cfg
        self.device = torch.device("cpu")
        self.dtype = torch.bfloat16
        self.name_or_path = "fake"
        self.generation_config = SimpleNamespace(to_dict=lambda: {"max_length": 20})
        tok = SyntheticTokenizer(vocab, 1, 2, 0)
        self._newline = [i for i, t in enumerate(tok._id2tok) if "\n" in t and i > 2]
        self._plain = [i for i, t in enumerate(tok._id2tok) if "\n" not in t and i > 2]

    def generate(self, input_ids=None, streamer=None, stopping_criteria=None, max_length=20,
                 bad_words_ids=None, begin_suppress_tokens=None, pixel_values=None, **kw):
        self.calls += 1
        ids: List[int] = input_ids[0].tolist()
        mix = (sum((i + 1) * t for i, t in enumerate(ids)) * 2654435761 + self.seed * 97 + self.calls * 7919) & 0xFFFFFFFF
        rng = random.Random(mix)
        if streamer is not None:
            streamer.put(input_ids.cpu())
        out = list(ids)
        while len(out) < max_length:
            r = rng.random()
            if r < 0.06 and len(out) > len(ids):
                tok = 2
            elif r < 0.35:
                tok = rng.choice(self._newline)
            else:
                tok = rng.choice(self._plain)
            out.append(tok)
            if streamer is not None:
                streamer.put(torch.tensor([tok]))
            stop = tok == 2
            for crit in (stopping_criteria or []):
                stop = stop or bool(crit(torch.tensor([out]), None))
            if stop:
                break
        if streamer is not None:
            streamer.end()
        return torch.tensor([out])


def rel_l2(a, b) -> float:
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def peaked_lm_head(lm_head: torch.Tensor, beta: float = 2.0, seed: int = 0, clip: int = 8) -> torch.Tensor:
    """a second synthetic weight set with PEAKED logits: every lm_head row is multiplied by 2^e, e = round(beta * z) with z ~ N(0, 1)
    per row (clipped to +-clip).  A power of two is exact in bf16, so the result is still a bf16 tensor and the oracle and the
    device see the same values.  The row norms become log-normal: a few dozen rows dominate every step's logits and the top-1 /
    top-2 gap is tens of bf16 ulps instead of ~1 (uniform rows: the top of 32 k equal-variance values, P(gap < 1 ulp) ~ 10 %)."""
    g = torch.Generator().manual_seed(seed)
    e = torch.clamp(torch.round(beta * torch.randn(lm_head.shape[0], generator=g)), -clip, clip)
    return lm_head.float() * torch.exp2(e)[:, None]


def engines():
    """both engines of a model's batch slots — the native run loop (default) and the Python-driven one — for tests that pin a
    property of the SEQUENCES (tokens, prefix paths), which must not depend on who turns the crank"""
    from detikzify_amd.infer.batching import BatchEngine
    from detikzify_amd.infer.engine import NativeBatchEngine
    return [NativeBatchEngine, BatchEngine]


# ---- the parity envelope.  north_star's literal "1e-3 on encoder logits" cannot hold between two bf16
# pipelines of this depth (the bf16-policy oracle itself sits 1.2e-2 .. 5e-2 from fp32), so the asserted statement is: the device is
# no further from the fp32 oracle than ENVELOPE x the bf16-policy oracle is, plus a small absolute slack for the shallow cases where
# both errors are a few rounding flips.  Rounds 1-5 asserted 1.5 x + 2e-3 / 1e-3 and MEASURED 1.00-1.07 x (profiles/r05_pytest_gpu
# _summary.txt: "worst ratio to the envelope 0.64-0.69"): a kernel could have lost 45 % and stayed green.  Round 6
# asserts what is measured, with the headroom of one more rounding-flip random walk: 1.15 x + half the old slack.
ENVELOPE = 1.15
SLACK_LOGITS = 1e-3         # logits of a prefill / decode step (was 2e-3)
SLACK_SMALL = 5e-4          # ViT features, single blocks, shallow decoders (was 1e-3)
SLACK_MX = 2e-3             # the opt-in MXFP8 step against the oracle that quantises the same activations (was 4e-3)


def envelope_ratio(e_dev: float, e_orc: float, slack: float = SLACK_LOGITS) -> float:
    """< 1: inside the envelope"""
    return e_dev / (ENVELOPE * e_orc + slack)


BF16_ULP = 2.0 ** -7          # one bf16 ulp relative to the value's binade top (8 significant bits)
GAP_BINS = (0.0, 1.0, 2.0, 4.0, 8.0, 16.0, 32.0, float("inf"))


def top2_gap_ulps(logits, bad, begin, first) -> float:
    """top-1 minus top-2 of the processed scores (oracle/sampling.py::mask_scores), in bf16 ulps of the top logit"""
    from oracle import sampling
    top2 = torch.topk(sampling.mask_scores(logits, bad, begin, first), 2)[0]
    return float(top2[0] - top2[1]) / (float(top2[0].abs()) * BF16_ULP + 1e-30)


def gap_histogram(gaps) -> str:
    counts = [0] * (len(GAP_BINS) - 1)
    for g in gaps:
        for b in range(len(counts)):
            if GAP_BINS[b] <= g < GAP_BINS[b + 1]:
                counts[b] += 1
                break
    return " ".join(f"[{GAP_BINS[b]:g},{GAP_BINS[b + 1]:g}):{c}" for b, c in enumerate(counts))
