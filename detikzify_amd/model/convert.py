"""
Checkpoint-name conversion for the v2 models.

A v2 checkpoint is the state dict of DetikzifyForConditionalGeneration (reference
detikzify/model/modeling_detikzify.py:119-135,274-285):
    model.vision_model.[vision_model.]…      HF SiglipVisionModel (separate q/k/v projections, MAP `head`)
    model.connector.modality_projection.proj.weight      Linear(3*D -> d, bias=False)   (:62-70)
    model.text_model.…                        HF LlamaModel
    lm_head.weight
The device registry (csrc/dtk_api.hip plan()) uses the v1 names: HF LlamaForCausalLM names for the decoder,
`model.mm_projector.*`, and timm VisionTransformer names for the tower (fused qkv, `attn_pool`).  The two ViTs
are the same network (tests/golden/siglip_tiny.npz pins the oracle's timm-named restatement against HF's
SiglipVisionModel), so conversion is renaming plus two row-concatenations:
    q_proj,k_proj,v_proj -> attn.qkv            head.attention.in_proj -> attn_pool.q / attn_pool.kv
"""
from __future__ import annotations

import re
from typing import Dict, Iterator, List, Tuple

import torch

_VIT_DIRECT = {
    "embeddings.patch_embedding.weight": "patch_embed.proj.weight",
    "embeddings.patch_embedding.bias": "patch_embed.proj.bias",
    "embeddings.position_embedding.weight": "pos_embed",
    "post_layernorm.weight": "norm.weight",
    "post_layernorm.bias": "norm.bias",
    "head.probe": "attn_pool.latent",
    "head.attention.out_proj.weight": "attn_pool.proj.weight",
    "head.attention.out_proj.bias": "attn_pool.proj.bias",
    "head.layernorm.weight": "attn_pool.norm.weight",
    "head.layernorm.bias": "attn_pool.norm.bias",
    "head.mlp.fc1.weight": "attn_pool.mlp.fc1.weight",
    "head.mlp.fc1.bias": "attn_pool.mlp.fc1.bias",
    "head.mlp.fc2.weight": "attn_pool.mlp.fc2.weight",
    "head.mlp.fc2.bias": "attn_pool.mlp.fc2.bias",
}
_VIT_LAYER = {
    "layer_norm1": "norm1", "layer_norm2": "norm2", "self_attn.out_proj": "attn.proj",
    "mlp.fc1": "mlp.fc1", "mlp.fc2": "mlp.fc2",
}
_LAYER_RE = re.compile(r"^encoder\.layers\.(\d+)\.(.+)\.(weight|bias)$")


def is_v2_key(key: str) -> bool:
    return key.startswith(("model.text_model.", "model.connector.", "model.vision_model."))


class V2Converter:
    """Streams (key, tensor) pairs of a v2 state dict and yields (registry name, tensor) pairs as soon as they
    are complete; the q/k/v (and in_proj) groups are buffered until all their parts have been seen."""

    def __init__(self):
        self._qkv: Dict[Tuple[int, str], Dict[str, torch.Tensor]] = {}

    def feed(self, key: str, t: torch.Tensor) -> List[Tuple[str, torch.Tensor]]:
        if key == "lm_head.weight":
            return [(key, t)]
        if key.startswith("model.text_model."):
            return [("model." + key[len("model.text_model."):], t)]
        if key == "model.connector.modality_projection.proj.weight":
            return [("model.mm_projector.weight", t)]
        if key.startswith("model.vision_model."):
            k = key[len("model.vision_model."):]
            if k.startswith("vision_model."):          # transformers < 5 nests SiglipVisionTransformer once more
                k = k[len("vision_model."):]
            return self._vit(k, t)
        return []

    def _vit(self, k: str, t: torch.Tensor) -> List[Tuple[str, torch.Tensor]]:
        vp = "vision_model."
        if k in _VIT_DIRECT:
            return [(vp + _VIT_DIRECT[k], t)]
        if k in ("head.attention.in_proj_weight", "head.attention.in_proj_bias"):
            kind = "weight" if k.endswith("weight") else "bias"
            D = t.shape[0] // 3
            return [(vp + f"attn_pool.q.{kind}", t[:D]), (vp + f"attn_pool.kv.{kind}", t[D:])]
        m = _LAYER_RE.match(k)
        if not m:
            return []
        i, mod, kind = int(m.group(1)), m.group(2), m.group(3)
        if mod in _VIT_LAYER:
            return [(vp + f"blocks.{i}.{_VIT_LAYER[mod]}.{kind}", t)]
        if mod in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj"):
            grp = self._qkv.setdefault((i, kind), {})
            grp[mod[-6]] = t                                # 'q' | 'k' | 'v'
            if len(grp) == 3:
                del self._qkv[(i, kind)]
                return [(vp + f"blocks.{i}.attn.qkv.{kind}", torch.cat([grp["q"], grp["k"], grp["v"]], 0))]
        return []

    def finish(self):
        if self._qkv:
            raise KeyError(f"incomplete q/k/v groups in the vision tower: {sorted(self._qkv)[:4]}")


def registry_to_v2(name: str, t: torch.Tensor, vit_dim: int) -> Iterator[Tuple[str, torch.Tensor]]:
    """Inverse direction (fixtures, export): registry name -> v2 checkpoint key(s)."""
    vp, hv = "vision_model.", "model.vision_model.vision_model."
    inv_direct = {v: k for k, v in _VIT_DIRECT.items()}
    inv_layer = {v: k for k, v in _VIT_LAYER.items()}
    if name == "lm_head.weight":
        yield name, t
    elif name == "model.mm_projector.weight":
        yield "model.connector.modality_projection.proj.weight", t
    elif name.startswith(vp):
        k = name[len(vp):]
        if k in inv_direct:
            yield hv + inv_direct[k], (t.reshape(-1, vit_dim) if k == "pos_embed" else t)
            return
        m = re.match(r"^attn_pool\.(q|kv)\.(weight|bias)$", k)
        if m:
            yield f"__inproj__.{m.group(1)}.{m.group(2)}", t      # caller concatenates q + kv
            return
        m = re.match(r"^blocks\.(\d+)\.(.+)\.(weight|bias)$", k)
        i, mod, kind = int(m.group(1)), m.group(2), m.group(3)
        if mod == "attn.qkv":
            for part, piece in zip("qkv", t.split(vit_dim, 0)):
                yield hv + f"encoder.layers.{i}.self_attn.{part}_proj.{kind}", piece
        else:
            yield hv + f"encoder.layers.{i}.{inv_layer[mod]}.{kind}", t
    elif name.startswith("model."):
        yield "model.text_model." + name[len("model."):], t
