"""
DeTikZify model glue restated on CPU (test infrastructure, see oracle/__init__.py).  Written against v1; the
v2 model (detikzify/model/modeling_detikzify.py:119-271) differs only in configuration here: GQA + rope "llama3"
in the decoder (oracle/llama.py), bias-free connector, HF SigLIP naming of the same ViT, a dedicated image token.

reference detikzify/model/v1/modeling_detikzify.py:
  get_vision_features   :132-137   feats[:, -n*c:].reshape(-1, n, D*c)  (3 consecutive patches)
  mm_projector          :108-114,163  nn.Linear(3D, d) WITH bias
  embedding splice      :158-189   the n consecutive image_token positions are replaced
  vision branch gate    :160       only when input_ids.shape[1] != 1 (prefill) and pixels given
  lm_head + .float()    :250-257
and the generate loop of HF GenerationMixin._sample as the reference configures it
(detikzify/infer/generate.py:209-227).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch

from . import sampling
from .llama import LlamaOracle
from .ops import linear
from .vit import VitOracle


class DetikzifyOracle:
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], precision: str = "bf16"):
        self.cfg, self.w, self.P = cfg, weights, precision
        self.vit = VitOracle(cfg, weights, precision)
        self.llm = LlamaOracle(cfg, weights, precision)
        n_patches = (cfg["vit_image"] // cfg["vit_patch"]) ** 2
        self.n_img = n_patches // cfg["concat_patches"]

    # -- vision ------------------------------------------------------------------------------
    def vision_features(self, pixels: torch.Tensor) -> torch.Tensor:
        """pixels [3,S,S] fp32 -> [n_img, concat*D]."""
        feats = self.vit.intermediate(pixels, self.cfg["vit_feature_layer"])
        c, n = self.cfg["concat_patches"], self.n_img
        return feats[-n * c:].reshape(n, feats.shape[-1] * c)

    def image_embeds(self, pixels: torch.Tensor, vit_feats: Optional[torch.Tensor] = None) -> torch.Tensor:
        # v1: nn.Linear(3D, d) with bias; v2 connector (modeling_detikzify.py:62-86): reshape(seq // 3, 3D) + bias-free Linear
        # vit_feats: the tower's output for `pixels` computed earlier by self.vit.intermediate (the tests keep it: one tower pass
        # instead of one per prefill)
        if vit_feats is None:
            feats = self.vision_features(pixels)
        else:
            c, n = self.cfg["concat_patches"], self.n_img
            feats = vit_feats[-n * c:].reshape(n, vit_feats.shape[-1] * c)
        return linear(feats, self.w["model.mm_projector.weight"], self.w.get("model.mm_projector.bias"), self.P)

    # -- decoder -----------------------------------------------------------------------------
    def input_embeds(self, ids: torch.Tensor, pixels: Optional[torch.Tensor], vit_feats: Optional[torch.Tensor] = None) -> torch.Tensor:
        x = self.llm.embed(ids)
        if pixels is not None and ids.numel() != 1:
            img = self.image_embeds(pixels, vit_feats)
            tok = self.cfg["image_token_id"]
            where = torch.where(ids == tok)[0]
            if where.numel() == 0:
                return x
            if where.numel() != self.n_img:
                raise ValueError("The number of image patch tokens should be the same as the number of image patches.")
            s = int(where[0])
            if not torch.equal(where, torch.arange(s, s + self.n_img)):
                raise ValueError("The image patch tokens should be consecutive.")
            x = torch.cat([x[:s], img, x[s + self.n_img:]], dim=0)
        return x

    def prefill(self, ids: torch.Tensor, pixels: Optional[torch.Tensor], vit_feats: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Fresh forward over the whole prompt; returns fp32 logits of the last position."""
        self.llm.reset()
        h = self.llm.forward(self.input_embeds(ids, pixels, vit_feats))
        return self.llm.logits(h[-1])

    def extend(self, tokens, last_only: bool = False) -> torch.Tensor:
        """Teacher-force `tokens` after the cached context in ONE pass: row i = the fp32 logits after tokens[i] — what len(tokens)
        calls of step() return, with the weights read once for all rows (the same causal forward the reference runs over a whole
        sequence, v1/modeling_detikzify.py:218-283; fp32 accumulation, every rounding point per row unchanged)."""
        h = self.llm.forward(self.llm.embed(torch.as_tensor(tokens, dtype=torch.long).reshape(-1)))
        return self.llm.logits(h[-1]) if last_only else self.llm.logits(h)

    def snapshot(self):
        """(keys, values, position) of the decoder cache: restore() returns to it, so several continuations of one prefix share its prefill"""
        return list(self.llm.k), list(self.llm.v), self.llm.pos

    def restore(self, snap) -> None:
        self.llm.k, self.llm.v, self.llm.pos = list(snap[0]), list(snap[1]), snap[2]

    def step(self, token: int) -> torch.Tensor:
        h = self.llm.forward(self.llm.embed(torch.tensor([token])))
        return self.llm.logits(h[-1])

    def generate(self, ids: torch.Tensor, pixels, max_new_tokens: int, do_sample=False,
                 temperature=1.0, top_k=0, top_p=1.0, seed=0, bad=(), begin=(), always=(),
                 eos: Optional[int] = None, return_logits=False):
        """HF _sample loop: returns the list of new tokens (and per-step logits)."""
        logits = self.prefill(ids, pixels)
        out: List[int] = []
        all_logits = []
        for n in range(max_new_tokens):
            if return_logits:
                all_logits.append(logits.clone())
            if do_sample:
                tok, _ = sampling.draw(logits, temperature, top_k, top_p, seed, n, bad, begin, n == 0, always)
            else:
                tok = sampling.greedy(logits, bad, begin, n == 0, always)
            out.append(tok)
            if eos is not None and tok == eos:
                break
            if n + 1 < max_new_tokens:
                logits = self.step(tok)
        return (out, all_logits) if return_logits else out
