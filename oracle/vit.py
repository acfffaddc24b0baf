"""
timm VisionTransformer `vit_so400m_patch14_siglip_384` restated on CPU (test infrastructure,
see oracle/__init__.py).  timm ~=1.0.11 is an un-vendored dependency of the reference
(pyproject.toml:46-48) and is not installed here; the call pattern comes from the reference:
  create_model(...)                                   v1/modeling_detikzify.py:94
  get_intermediate_layers(pixels, n=[layer], norm=True)   :71-72, :134
  forward_features + forward_head (SelfSim pooling)       :63-69
and the published timm algorithm (timm/models/vision_transformer.py, timm/layers/attention_pool.py):
  PatchEmbed: Conv2d(3, D, p, stride p) -> flatten -> [N, D];  x = x + pos_embed (no class token)
  Block: x = x + proj(attn(norm1(x)));  x = x + fc2(gelu(fc1(norm2(x))))
         fused qkv Linear(D, 3D): rows [0:D]=q, [D:2D]=k, [2D:3D]=v, head h = rows h*hd..(h+1)*hd
  norm: final LayerNorm (eps 1e-6), applied by get_intermediate_layers(norm=True)
  AttentionPoolLatent ('map'): q = q(latent), k,v = kv(x) (rows [0:D]=k, [D:2D]=v),
         x = proj(sdpa(q,k,v)); x = x + mlp(norm(x)); token 0.
GELU flavour: timm's default act_layer is erf-GELU; original SigLIP uses the tanh approximation.
It is a config switch (`vit_gelu_tanh`) until a timm checkpoint is reachable (parity unpinned).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

from .llama import attention
from .ops import linear, rb


def layernorm(x, w, b, eps, precision="bf16"):
    return rb(F.layer_norm(x, (x.shape[-1],), w, b, eps), precision)


def gelu(x, tanh: bool, precision="bf16"):
    return rb(F.gelu(x, approximate="tanh" if tanh else "none"), precision)


def im2col(pixels: torch.Tensor, patch: int) -> torch.Tensor:
    """[3,S,S] -> [N, 3*p*p] with column order (c, kh, kw) = flattening of the conv weight."""
    c, s, _ = pixels.shape
    n = s // patch                      # Conv2d(stride=patch) drops the trailing s % patch pixels
    pixels = pixels[:, : n * patch, : n * patch]
    x = pixels.reshape(c, n, patch, n, patch).permute(1, 3, 0, 2, 4)  # py, px, c, kh, kw
    return x.reshape(n * n, c * patch * patch)


class VitOracle:
    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], precision="bf16",
                 prefix="vision_model."):
        self.cfg, self.w, self.P, self.pre = cfg, weights, precision, prefix
        self.D, self.depth, self.H = cfg["vit_dim"], cfg["vit_depth"], cfg["vit_heads"]
        self.hd = self.D // self.H
        self.eps = cfg["vit_ln_eps"]
        self.tanh = bool(cfg["vit_gelu_tanh"])

    def _g(self, name):
        return self.w[self.pre + name]

    def _heads(self, x):  # [N, D] -> [H, N, hd]
        return x.view(x.shape[0], self.H, self.hd).transpose(0, 1)

    def embed(self, pixels: torch.Tensor) -> torch.Tensor:
        P = self.P
        cols = rb(im2col(pixels, self.cfg["vit_patch"]), P)  # to_input_dtype: pixels -> bf16
        wpe = self._g("patch_embed.proj.weight").reshape(self.D, -1)
        x = linear(cols, wpe, self._g("patch_embed.proj.bias"), P)
        return rb(x + self._g("pos_embed").reshape(-1, self.D), P)

    def block(self, x, i):
        P, p = self.P, f"blocks.{i}."
        h = layernorm(x, self._g(p + "norm1.weight"), self._g(p + "norm1.bias"), self.eps, P)
        qkv = linear(h, self._g(p + "attn.qkv.weight"), self._g(p + "attn.qkv.bias"), P)
        q, k, v = (self._heads(t) for t in qkv.split(self.D, dim=-1))
        a = attention(q, k, v, 1.0 / math.sqrt(self.hd), None, P)
        a = a.transpose(0, 1).reshape(-1, self.D)
        x = rb(x + linear(a, self._g(p + "attn.proj.weight"), self._g(p + "attn.proj.bias"), P), P)
        h = layernorm(x, self._g(p + "norm2.weight"), self._g(p + "norm2.bias"), self.eps, P)
        h = gelu(linear(h, self._g(p + "mlp.fc1.weight"), self._g(p + "mlp.fc1.bias"), P), self.tanh, P)
        return rb(x + linear(h, self._g(p + "mlp.fc2.weight"), self._g(p + "mlp.fc2.bias"), P), P)

    def final_norm(self, x):
        return layernorm(x, self._g("norm.weight"), self._g("norm.bias"), self.eps, self.P)

    def intermediate(self, pixels, layer: int, return_all=False):
        """get_intermediate_layers(pixels, n=[layer], norm=True)[0] -> [N, D]."""
        x = self.embed(pixels)
        outs = {}
        for i in range(layer + 1):
            x = self.block(x, i)
            outs[i] = x
        feats = self.final_norm(x)
        return (feats, outs, x) if return_all else feats

    def forward_features(self, pixels):
        x = self.embed(pixels)
        for i in range(self.depth):
            x = self.block(x, i)
        return self.final_norm(x)

    def forward_head(self, last_hidden):
        """AttentionPoolLatent with latent_len 1, pool_type 'token' -> [D]."""
        P = self.P
        lat = self._g("attn_pool.latent").reshape(1, self.D)
        q = self._heads(linear(lat, self._g("attn_pool.q.weight"), self._g("attn_pool.q.bias"), P))
        kv = linear(last_hidden, self._g("attn_pool.kv.weight"), self._g("attn_pool.kv.bias"), P)
        k, v = (self._heads(t) for t in kv.split(self.D, dim=-1))
        a = attention(q, k, v, 1.0 / math.sqrt(self.hd), None, P).transpose(0, 1).reshape(1, self.D)
        x = linear(a, self._g("attn_pool.proj.weight"), self._g("attn_pool.proj.bias"), P)
        h = layernorm(x, self._g("attn_pool.norm.weight"), self._g("attn_pool.norm.bias"), self.eps, P)
        h = gelu(linear(h, self._g("attn_pool.mlp.fc1.weight"), self._g("attn_pool.mlp.fc1.bias"), P), self.tanh, P)
        x = rb(x + linear(h, self._g("attn_pool.mlp.fc2.weight"), self._g("attn_pool.mlp.fc2.bias"), P), P)
        return x[0]

    def forward(self, pixels):
        """DetikzifyVisionModel.forward: (last_hidden_state [N,D], pooler_output [D])."""
        lh = self.forward_features(pixels)
        return lh, self.forward_head(lh)
