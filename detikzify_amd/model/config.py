"""
Model configuration.  Mirrors the fields the reference reads from the checkpoint's config.json
(DetikzifyConfig(LlamaConfig), reference detikzify/model/v1/configuration_detikzify.py:3-13, plus
the vision fields written by initialize_vision_modules, v1/modeling_detikzify.py:98-106).
`text_config` is `self`: DetikzifyGenerator.generate reads model.config.text_config.eos_token_id
(infer/generate.py:221), which a flat v1 config does not have upstream.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass, field
from pathlib import Path
from typing import Any, Dict


@dataclass
class DetikzifyConfig:
    # LLaMA text decoder (HF LlamaConfig names in comments)
    hidden: int = 4096               # hidden_size
    layers: int = 32                 # num_hidden_layers
    heads: int = 32                  # num_attention_heads (== num_key_value_heads for v1)
    head_dim: int = 128
    ffn: int = 11008                 # intermediate_size
    vocab: int = 32256               # vocab_size
    max_positions: int = 2048        # tokenizer.model_max_length (v1/__init__.py:28)
    rms_eps: float = 1e-6            # rms_norm_eps
    rope_theta: float = 100000.0
    rope_factor: float = 4.0         # rope_scaling {"type": "linear", "factor": 4}
    bos_token_id: int = 32013
    eos_token_id: int = 32014
    pad_token_id: int = 32018
    # vision tower: timm vit_so400m_patch14_siglip_384.webli (v1/__init__.py:24)
    vit_dim: int = 1152
    vit_depth: int = 27
    vit_heads: int = 16
    vit_mlp: int = 4304
    vit_patch: int = 14
    vit_image: int = 384
    vit_feature_layer: int = 26      # feature_layer=-1 -> clip(...) % depth (modeling_detikzify.py:104)
    vit_ln_eps: float = 1e-6
    vit_gelu_tanh: int = 0
    concat_patches: int = 3
    patch_token_id: int = 32013      # == BOS (v1/__init__.py:49)
    attn_splits: int = 8
    batch_slots: int = 0             # KV slots for batched decode of independent rollouts (0 = none)
    weight_format: str = "bf16"      # "bf16" | "fp8" (e4m3 decoder Linear weights, per-row 2^e scales)
    model_type: str = "detikzify"
    name_or_path: str = ""
    vision_tower: str = "vit_so400m_patch14_siglip_384.webli"

    # ---- compat properties of the reference config -----------------------------------------
    @property
    def image_token_id(self) -> int:
        return self.patch_token_id

    @property
    def pooling_mode(self) -> str:
        return "cos"

    @property
    def text_config(self) -> "DetikzifyConfig":
        return self

    @property
    def num_patches(self) -> int:
        return (self.vit_image // self.vit_patch) ** 2 // self.concat_patches

    @property
    def mm_hidden_size(self) -> int:
        return self.vit_dim * self.concat_patches

    @property
    def hidden_size(self) -> int:
        return self.hidden

    @property
    def vocab_size(self) -> int:
        return self.vocab

    def to_dict(self) -> Dict[str, Any]:
        d = asdict(self)
        d["image_token_id"] = self.image_token_id
        return d

    def kernel_dict(self) -> Dict[str, Any]:
        """the fields of include/dtk.h's dtk_config (also what the test oracle consumes)"""
        keys = ["hidden", "layers", "heads", "head_dim", "ffn", "vocab", "max_positions", "rms_eps",
                "rope_theta", "rope_factor", "vit_dim", "vit_depth", "vit_heads", "vit_mlp",
                "vit_patch", "vit_image", "vit_feature_layer", "vit_ln_eps", "vit_gelu_tanh",
                "concat_patches", "attn_splits"]
        d = {k: getattr(self, k) for k in keys}
        d["image_token_id"] = self.image_token_id
        return d

    @classmethod
    def from_hf_json(cls, path: str) -> "DetikzifyConfig":
        """Read a checkpoint's config.json (HF LlamaConfig names)."""
        j = json.loads(Path(path).read_text())
        rs = j.get("rope_scaling") or {}
        c = cls(
            hidden=j["hidden_size"], layers=j["num_hidden_layers"], heads=j["num_attention_heads"],
            head_dim=j.get("head_dim") or j["hidden_size"] // j["num_attention_heads"],
            ffn=j["intermediate_size"], vocab=j["vocab_size"],
            rms_eps=j.get("rms_norm_eps", 1e-6), rope_theta=j.get("rope_theta", 10000.0),
            rope_factor=float(rs.get("factor", 1.0)) if rs.get("type", rs.get("rope_type", "linear")) == "linear" else 1.0,
            bos_token_id=j.get("bos_token_id", 1), eos_token_id=j.get("eos_token_id", 2),
            pad_token_id=j.get("pad_token_id") or 0,
            patch_token_id=j.get("patch_token_id", j.get("bos_token_id", 1)),
            concat_patches=j.get("concat_patches", 3),
            vit_feature_layer=j.get("feature_layer", 26),
            max_positions=j.get("model_max_length", 2048),
        )
        # optional vision-tower description (written by our own fixtures; real v1 checkpoints use the defaults)
        for src, dst in (("vit_dim", "vit_dim"), ("vit_depth", "vit_depth"), ("vit_heads", "vit_heads"), ("vit_mlp", "vit_mlp"),
                         ("vit_patch", "vit_patch"), ("vit_image", "vit_image"), ("vit_gelu_tanh", "vit_gelu_tanh"),
                         ("attn_splits", "attn_splits")):
            if src in j:
                setattr(c, dst, j[src])
        if j.get("num_key_value_heads", c.heads) != c.heads:
            raise NotImplementedError("GQA checkpoints (v2 models) are not supported by this build")
        return c


def _tiny() -> DetikzifyConfig:
    # exercises every kernel path of the real models at toy size: hd 128 decoder heads, ViT head
    # dim 72, mlp % 32 == 16, patch K (588) padded to 592, image 90 % 14 = 6 trailing pixels dropped like 384 % 14, N=36 patches
    return DetikzifyConfig(hidden=256, layers=2, heads=2, ffn=688, vocab=512, max_positions=160,
                           rms_eps=1e-6, rope_theta=100000.0, rope_factor=4.0,
                           bos_token_id=1, eos_token_id=2, pad_token_id=0, patch_token_id=1,
                           vit_dim=144, vit_depth=2, vit_heads=2, vit_mlp=304, vit_patch=14,
                           vit_image=90, vit_feature_layer=1, attn_splits=4,
                           name_or_path="detikzify-tiny")


PRESETS = {
    # dimensions from the upstream model cards — real checkpoints override them
    # through from_hf_json; these presets exist for synthetic-weight runs.
    "detikzify-tiny": _tiny,
    "detikzify-ds-1.3b": lambda: DetikzifyConfig(hidden=2048, layers=24, heads=16, ffn=5504, vocab=32256,
                                                 name_or_path="nllg/detikzify-ds-1.3b"),
    "detikzify-ds-7b": lambda: DetikzifyConfig(name_or_path="nllg/detikzify-ds-7b"),
    "detikzify-cl-7b": lambda: DetikzifyConfig(vocab=32024, rms_eps=1e-5, rope_theta=1e6, rope_factor=1.0,
                                               bos_token_id=1, eos_token_id=2, pad_token_id=32016,
                                               patch_token_id=1, name_or_path="nllg/detikzify-cl-7b"),
}


def preset(name: str) -> DetikzifyConfig:
    key = name.split("/")[-1]
    if key not in PRESETS:
        raise KeyError(f"unknown preset {name!r}; known: {sorted(PRESETS)}")
    return PRESETS[key]()
