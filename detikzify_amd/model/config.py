"""
Model configuration.  Mirrors the fields the reference reads from the checkpoint's config.json
(DetikzifyConfig(LlamaConfig), reference detikzify/model/v1/configuration_detikzify.py:3-13, plus
the vision fields written by initialize_vision_modules, v1/modeling_detikzify.py:98-106).
`text_config` is `self`: DetikzifyGenerator.generate reads model.config.text_config.eos_token_id
(infer/generate.py:221), which a flat v1 config does not have upstream.
"""
from __future__ import annotations

import json
from dataclasses import asdict, dataclass
from pathlib import Path
from typing import Any, Dict


@dataclass
class DetikzifyConfig:
    # LLaMA text decoder (HF LlamaConfig names in comments)
    hidden: int = 4096               # hidden_size
    layers: int = 32                 # num_hidden_layers
    heads: int = 32                  # num_attention_heads
    kv_heads: int = 0                # num_key_value_heads; 0 = heads (MHA, every v1 model); v2 LLaMA-3.1: 8 (GQA)
    head_dim: int = 128
    ffn: int = 11008                 # intermediate_size
    vocab: int = 32256               # vocab_size
    max_positions: int = 2048        # tokenizer.model_max_length (v1/__init__.py:28)
    rms_eps: float = 1e-6            # rms_norm_eps
    rope_theta: float = 100000.0
    rope_factor: float = 4.0         # rope_scaling {"type": "linear", "factor": 4}; llama3: factor 8
    rope_type: str = "linear"        # "linear" (v1: DeepSeek / CodeLlama, factor 1 = none) | "llama3" (v2: LLaMA-3.1)
    rope_low_freq_factor: float = 1.0
    rope_high_freq_factor: float = 4.0
    rope_original_max_position: int = 8192
    bos_token_id: int = 32013
    eos_token_id: Any = 32014        # int, or a list of ids (HF allows several)
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
    patch_token_id: int = 32013      # == BOS (v1/__init__.py:49); v2: image_token_id 128005 (configuration_detikzify.py:89)
    proj_bias: bool = True           # v1 mm_projector nn.Linear(3D, d) with bias; v2 connector bias=False (modeling_detikzify.py:67)
    arch: str = "v1"                 # "v1" (timm tower + LlamaModel subclass) | "v2" (HF SigLIP + Idefics3-style merger)
    attn_splits: int = 0             # split-K factor of the decode attention; 0 = auto (16 for one sequence, 8 for the batched step)
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
        # v1 config carries pooling_mode "cos"; a v2 config has none, so ImageSim.from_detikzify falls back to "emd"
        # (reference evaluate/imagesim.py:63)
        return "cos" if self.arch == "v1" else "emd"

    @property
    def num_kv_heads(self) -> int:
        return self.kv_heads or self.heads

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

    def oracle_dict(self) -> Dict[str, Any]:
        """kernel_dict + the fields only the host (rope tables) and the test oracle need"""
        d = self.kernel_dict()
        d.update(kv_heads=self.num_kv_heads, proj_bias=self.proj_bias, rope_type=self.rope_type,
                 rope_low_freq_factor=self.rope_low_freq_factor, rope_high_freq_factor=self.rope_high_freq_factor,
                 rope_original_max_position=self.rope_original_max_position)
        return d

    @classmethod
    def from_hf_json(cls, path: str) -> "DetikzifyConfig":
        """Read a checkpoint's config.json: flat HF LlamaConfig names (v1) or the composite v2 layout with
        text_config / vision_config (reference configuration_detikzify.py:83-120)."""
        j = json.loads(Path(path).read_text())
        v2 = "text_config" in j
        t = j["text_config"] if v2 else j
        rs = t.get("rope_scaling") or t.get("rope_parameters") or {}
        rtype = rs.get("rope_type", rs.get("type", "linear" if rs else "default"))
        if rtype not in ("linear", "llama3", "default"):
            raise NotImplementedError(f"rope scaling {rtype!r} is not supported by this build")
        c = cls(
            hidden=t["hidden_size"], layers=t["num_hidden_layers"], heads=t["num_attention_heads"],
            head_dim=t.get("head_dim") or t["hidden_size"] // t["num_attention_heads"],
            ffn=t["intermediate_size"], vocab=t["vocab_size"],
            rms_eps=t.get("rms_norm_eps", 1e-5 if v2 else 1e-6), rope_theta=t.get("rope_theta", rs.get("rope_theta", 10000.0)),
            rope_factor=float(rs.get("factor", 1.0)) if rtype in ("linear", "llama3") else 1.0,
            rope_type="llama3" if rtype == "llama3" else "linear",
            bos_token_id=t.get("bos_token_id", 1), eos_token_id=t.get("eos_token_id", 2),
            pad_token_id=next((v for v in (j.get("pad_token_id"), t.get("pad_token_id")) if v is not None),
                              128004 if v2 else 0),        # v2 default: configuration_detikzify.py:89
            patch_token_id=j.get("image_token_id", 128005) if v2 else j.get("patch_token_id", j.get("bos_token_id", 1)),
            concat_patches=j.get("concat_factor", 3) if v2 else j.get("concat_patches", 3),
            vit_feature_layer=j.get("feature_layer", 26),
            max_positions=j.get("model_max_length", 2048),
            proj_bias=not v2, arch="v2" if v2 else "v1",
        )
        kvh = t.get("num_key_value_heads", c.heads)
        c.kv_heads = 0 if kvh == c.heads else int(kvh)
        if isinstance(c.eos_token_id, list) and len(c.eos_token_id) == 1:
            c.eos_token_id = c.eos_token_id[0]      # several ids stay a list: generate() stops on any of them
        if rtype == "llama3":
            c.rope_low_freq_factor = float(rs.get("low_freq_factor", 1.0))
            c.rope_high_freq_factor = float(rs.get("high_freq_factor", 4.0))
            c.rope_original_max_position = int(rs.get("original_max_position_embeddings", 8192))
        if v2:   # HF SigLIP vision_config (configuration_detikzify.py:28-57): the whole tower, post_layernorm output
            vc = j.get("vision_config") or {}
            c.vit_dim = vc.get("hidden_size", 1152); c.vit_mlp = vc.get("intermediate_size", 4304)
            c.vit_depth = vc.get("num_hidden_layers", 27); c.vit_heads = vc.get("num_attention_heads", 16)
            c.vit_image = vc.get("image_size", 420); c.vit_patch = vc.get("patch_size", 14)
            c.vit_ln_eps = vc.get("layer_norm_eps", 1e-6)
            c.vit_gelu_tanh = 1 if vc.get("hidden_act", "gelu_pytorch_tanh") == "gelu_pytorch_tanh" else 0
            c.vit_feature_layer = c.vit_depth - 1
            c.vision_tower = "siglip"
        # optional vision-tower description (written by our own fixtures; real v1 checkpoints use the defaults)
        for key in ("vit_dim", "vit_depth", "vit_heads", "vit_mlp", "vit_patch", "vit_image", "vit_gelu_tanh", "attn_splits"):
            if key in j:
                setattr(c, key, j[key])
        return c


def v2_8b(name: str) -> DetikzifyConfig:
    """nllg/detikzify-v2-8b / v2.5-8b: HF SigLIP so400m/14 at 420 px (900 patches -> 300 tokens) + LLaMA-3.1-8B
    (dimensions from the upstream model cards; a real checkpoint's config.json overrides them)."""
    return DetikzifyConfig(hidden=4096, layers=32, heads=32, kv_heads=8, ffn=14336, vocab=128256, rms_eps=1e-5,
                           rope_theta=500000.0, rope_factor=8.0, rope_type="llama3", bos_token_id=128000, eos_token_id=128001,
                           pad_token_id=128004, patch_token_id=128005, vit_image=420, vit_gelu_tanh=1, vit_feature_layer=26,
                           proj_bias=False, arch="v2", vision_tower="siglip", name_or_path=name)


def _tiny() -> DetikzifyConfig:
    # exercises every kernel path of the real models at toy size: hd 128 decoder heads, ViT head
    # dim 72, mlp % 32 == 16, patch K (588) padded to 592, image 90 % 14 = 6 trailing pixels dropped like 384 % 14, N=36 patches
    return DetikzifyConfig(hidden=256, layers=2, heads=2, ffn=688, vocab=512, max_positions=160,
                           rms_eps=1e-6, rope_theta=100000.0, rope_factor=4.0,
                           bos_token_id=1, eos_token_id=2, pad_token_id=0, patch_token_id=1,
                           vit_dim=144, vit_depth=2, vit_heads=2, vit_mlp=304, vit_patch=14,
                           vit_image=90, vit_feature_layer=1, attn_splits=4,
                           name_or_path="detikzify-tiny")


def _tiny_v2() -> DetikzifyConfig:
    # the v2 differences at toy size: GQA 4 query / 2 kv heads, rope "llama3", bias-free connector, tanh GELU,
    # a dedicated image token, 7x7 = 49 -> 48 patches... (98 px / 14 = 7 -> 49 patches is not divisible by 3: use 84 px -> 36)
    return DetikzifyConfig(hidden=512, layers=2, heads=4, kv_heads=2, ffn=688, vocab=640, max_positions=192,
                           rms_eps=1e-5, rope_theta=500000.0, rope_factor=8.0, rope_type="llama3",
                           rope_original_max_position=64, bos_token_id=1, eos_token_id=2, pad_token_id=0, patch_token_id=5,
                           vit_dim=144, vit_depth=2, vit_heads=2, vit_mlp=304, vit_patch=14, vit_image=84,
                           vit_feature_layer=1, vit_gelu_tanh=1, attn_splits=4, proj_bias=False, arch="v2",
                           vision_tower="siglip", name_or_path="detikzify-tiny-v2")


PRESETS = {
    # dimensions from the upstream model cards — real checkpoints override them
    # through from_hf_json; these presets exist for synthetic-weight runs.
    "detikzify-tiny": _tiny,
    "detikzify-tiny-v2": _tiny_v2,
    "detikzify-v2-8b": lambda: v2_8b("nllg/detikzify-v2-8b"),
    "detikzify-v2.5-8b": lambda: v2_8b("nllg/detikzify-v2.5-8b"),
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
