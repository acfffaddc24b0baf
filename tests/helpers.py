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
