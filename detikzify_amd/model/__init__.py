"""
load(): the producer side of the drop-in boundary (reference detikzify/model/__init__.py:28-61 and
detikzify/model/v1/__init__.py:24-56).  Returns the (model, processor) pair that
DetikzifyPipeline / DetikzifyGenerator / ImageSim.from_detikzify consume.

Two sources of weights:
  * a local checkpoint directory in HF layout (config.json + *.safetensors [+ tokenizer files],
    vision tower tensors under the "vision_model." prefix in timm naming);
  * `synthetic=<seed>`: seeded synthetic weights at a preset's shapes — the only option offline
    (no checkpoints or tokenizer files exist in this environment).
"""
from __future__ import annotations

import os
from pathlib import Path
from typing import Optional, Tuple

from .config import DetikzifyConfig, PRESETS, preset
from .modeling import DetikzifyForCausalLM, DetikzifyVisionModel, GenerationConfig
from .processing import BatchFeature, DetikzifyImageProcessor, DetikzifyProcessor
from .tokenizer import SyntheticTokenizer, load_tokenizer

# names the reference resolves to its v1 loader (detikzify/model/v1/__init__.py:10-15)
v1_models = ["nllg/detikzify-ds-1.3b", "nllg/detikzify-ds-7b", "nllg/detikzify-tl-1.1b", "nllg/detikzify-cl-7b"]
# the current upstream default family (README.md:21-30): HF SigLIP-420 + LLaMA-3.1-8B (GQA), loaded through the
# same entry point (detikzify/model/__init__.py:44-61)
v2_models = ["nllg/detikzify-v2-8b", "nllg/detikzify-v2.5-8b"]


def _default_device() -> int:
    return int(os.environ.get("LOCAL_RANK", "0"))


# from_pretrained arguments the reference's callers pass (examples/infer.py:33-37, examples/eval.py:110-115,
# webui/webui.py:76-81) that have no meaning on this path: placement is one GPU per context, attention is always the fused
# kernel, there is no remote code and no lazy loading
_IGNORED_FROM_PRETRAINED = {"attn_implementation", "low_cpu_mem_usage", "trust_remote_code", "use_safetensors", "revision",
                            "cache_dir", "token", "local_files_only", "use_cache", "offload_folder", "offload_state_dict"}


def _check_dtype(torch_dtype) -> None:
    if torch_dtype is None:
        return
    name = str(torch_dtype).replace("torch.", "")
    if name not in ("bfloat16", "auto"):
        raise NotImplementedError(f"torch_dtype={torch_dtype}: this build stores weights and activations in bfloat16 "
                                  "(the dtype the reference's examples load with); other dtypes are not implemented")


def load(model_name_or_path: str, modality_projector: Optional[str] = None, is_v1: bool = False,
         synthetic: Optional[int] = None, device_map=None, torch_dtype=None, max_positions: Optional[int] = None,
         batch_slots: int = 0, weight_format: str = "bf16", synthetic_tokenizer: bool = False, vit_gelu_tanh: Optional[int] = None,
         **from_pretrained_kwargs) -> Tuple[DetikzifyForCausalLM, DetikzifyProcessor]:
    """(model, processor).  `device_map` may be an int GPU index (the reference passes
    device_map=RANK, examples/eval.py:112); torch_dtype must be bf16 / "auto" / None.

    A checkpoint directory must carry its tokenizer files, as the reference's loader requires
    (v1/__init__.py:26-34 fails hard otherwise); `synthetic_tokenizer=True` (or `"synthetic_tokenizer": true` in the
    directory's config.json, written by our weight-only test fixtures) opts into the byte-level stand-in.
    `vit_gelu_tanh` (0 | 1) overrides the vision tower's GELU flavour (a v1 config.json does not record it: `_announce_gelu`;
    tests/real_checkpoint.py runs both to see which one a real checkpoint wants)."""
    unknown = set(from_pretrained_kwargs) - _IGNORED_FROM_PRETRAINED
    if unknown:
        raise TypeError(f"load() got arguments it does not implement: {sorted(unknown)}")
    _check_dtype(torch_dtype)
    if isinstance(device_map, str) and device_map not in ("auto", "cuda", "cuda:0"):
        raise NotImplementedError(f"device_map={device_map!r}: pass a GPU index (one full replica per GPU, as examples/eval.py:112)")
    dev = device_map if isinstance(device_map, int) else _default_device()
    path = Path(model_name_or_path)
    if path.is_dir() and (path / "config.json").exists() and synthetic is None:
        import json
        cfg = DetikzifyConfig.from_hf_json(str(path / "config.json"))
        cfg.name_or_path = str(path)
        _require_supported(cfg)
        if max_positions:
            cfg.max_positions = max_positions
        if vit_gelu_tanh is not None:
            cfg.vit_gelu_tanh = int(bool(vit_gelu_tanh))
        cfg.batch_slots, cfg.weight_format = batch_slots, weight_format
        if synthetic_tokenizer or json.loads((path / "config.json").read_text()).get("synthetic_tokenizer"):
            tokenizer = _synthetic_tokenizer(cfg)
        else:
            tokenizer = load_tokenizer(str(path), cfg.max_positions, cfg.arch)       # raises when the files are missing / broken
            if cfg.arch == "v1":
                cfg.patch_token_id = tokenizer.bos_token_id      # v1/__init__.py:49
        model = DetikzifyForCausalLM(cfg, dev)
        _load_safetensors_dir(model, path)
        if modality_projector:
            _load_projector(model, modality_projector)
        gc_file = path / "generation_config.json"
        if gc_file.exists():                                     # what from_pretrained reads into model.generation_config
            model.generation_config.update_from_dict(json.loads(gc_file.read_text()))
        _announce_gelu(cfg, path)
    else:
        cfg = preset(model_name_or_path)
        _require_supported(cfg)
        if max_positions:
            cfg.max_positions = max_positions
        if vit_gelu_tanh is not None:
            cfg.vit_gelu_tanh = int(bool(vit_gelu_tanh))
        cfg.batch_slots, cfg.weight_format = batch_slots, weight_format
        if synthetic is None:
            raise FileNotFoundError(
                f"{model_name_or_path!r} is not a local checkpoint directory and there is no network; "
                "pass synthetic=<seed> for seeded synthetic weights at this preset's shapes")
        tokenizer = _synthetic_tokenizer(cfg)
        model = DetikzifyForCausalLM(cfg, dev)
        model.fill_synthetic(int(synthetic))
    model.generation_config.pad_token_id = tokenizer.pad_token_id     # v1/__init__.py:41
    image_processor = _checkpoint_image_processor(path, cfg)
    processor = DetikzifyProcessor(
        image_processor=image_processor, tokenizer=tokenizer, image_seq_len=cfg.num_patches,
        image_token=tokenizer.convert_ids_to_tokens(cfg.patch_token_id))
    return model, processor


def _synthetic_tokenizer(cfg: DetikzifyConfig) -> SyntheticTokenizer:
    eos = cfg.eos_token_id[0] if isinstance(cfg.eos_token_id, (list, tuple)) else cfg.eos_token_id
    return SyntheticTokenizer(cfg.vocab, bos_token_id=cfg.bos_token_id, eos_token_id=eos, pad_token_id=cfg.pad_token_id,
                              model_max_length=cfg.max_positions, image_token_id=cfg.patch_token_id)


def _require_supported(cfg: DetikzifyConfig) -> None:
    """Shapes the kernels are written for; everything else fails here with the reason, not deep inside dtk_create."""
    if cfg.head_dim != 128:
        raise NotImplementedError(
            f"{cfg.name_or_path or 'this checkpoint'}: decoder head_dim {cfg.head_dim} — the decode / prefill kernels are built "
            "for head_dim 128 (ds-1.3b, ds-7b, cl-7b, v2-8b); nllg/detikzify-tl-1.1b (TinyLlama, head_dim 64) is not supported yet")
    if cfg.concat_patches < 1 or ((cfg.vit_image // cfg.vit_patch) ** 2) % cfg.concat_patches:
        raise ValueError(f"concat_patches={cfg.concat_patches} does not divide the tower's {(cfg.vit_image // cfg.vit_patch) ** 2} patches")


def _announce_gelu(cfg: DetikzifyConfig, path: Path) -> None:
    """The v1 tower is timm's vit_so400m_patch14_siglip_384; a v1 config.json does not record the tower's activation, and
    timm is not available offline to settle it (open item).  Say which one is used, every time."""
    if cfg.arch != "v1":
        return
    import json
    import warnings
    if "vit_gelu_tanh" in json.loads((path / "config.json").read_text()) or getattr(cfg, "_gelu_stated", False):
        return
    warnings.warn(f"{path}: config.json does not state the vision tower's GELU flavour; using "
                  f"{'tanh-approximated' if cfg.vit_gelu_tanh else 'exact (erf)'} GELU, timm's default for "
                  f"{cfg.vision_tower}.  Set \"vit_gelu_tanh\": 0|1 in config.json to silence or override this.", stacklevel=3)


def _checkpoint_image_processor(path: Path, cfg: DetikzifyConfig) -> DetikzifyImageProcessor:
    """The image processor a checkpoint was trained with: v2 directories carry preprocessor_config.json (saved by the
    processor, reference model/__init__.py:44 loads it through AutoProcessor), v1 checkpoints keep the processor's dict in
    config.json under `vision_config` (v1/modeling_detikzify.py:112).  Without either: the tower's published data config
    (bicubic, 1/255, mean = std = 0.5) at the tower's resolution."""
    import json
    keys = ("size", "resample", "do_resize", "do_rescale", "rescale_factor", "do_normalize", "image_mean", "image_std")
    saved = {}
    if path.is_dir():
        if (path / "preprocessor_config.json").exists():
            saved = json.loads((path / "preprocessor_config.json").read_text())
        elif cfg.arch == "v1" and (path / "config.json").exists():
            saved = json.loads((path / "config.json").read_text()).get("vision_config") or {}
    kw = {k: saved[k] for k in keys if saved.get(k) is not None}
    size = kw.get("size")
    if not (isinstance(size, dict) and size.get("height") == cfg.vit_image and size.get("width") == cfg.vit_image):
        if size is not None:
            import warnings
            warnings.warn(f"image processor size {size} of {path} does not match the tower's {cfg.vit_image} px; using the tower's")
        kw["size"] = {"height": cfg.vit_image, "width": cfg.vit_image}
    return DetikzifyImageProcessor(**kw)


def _load_safetensors_dir(model: DetikzifyForCausalLM, path: Path):
    """HF-layout checkpoint -> device.  The v1 checkpoints carry the decoder + mm_projector; the
    timm tower ships separately (pretrained=True at v1/modeling_detikzify.py:94): accept it either
    inside the same files under "vision_model." / "model.vision_model.model.0." or as
    vision_tower.safetensors with bare timm names."""
    from safetensors import safe_open
    from .convert import V2Converter, is_v2_key
    known = set(model.tensor_names())
    seen = set()
    conv = V2Converter()
    files = sorted(path.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no *.safetensors under {path}")
    for f in files:
        bare_timm = f.name == "vision_tower.safetensors"
        with safe_open(str(f), framework="pt") as sf:
            for k in sf.keys():
                if not bare_timm and (is_v2_key(k) or (model.config.arch == "v2" and k == "lm_head.weight")):
                    pairs = conv.feed(k, sf.get_tensor(k))         # v2 checkpoint names (model/convert.py)
                else:
                    name = k
                    if bare_timm:       # timm names; the open_clip file timm downloads for the SigLIP towers says visual.trunk.*
                        name = "vision_model." + (k[len("visual.trunk."):] if k.startswith("visual.trunk.") else k)
                    for pre in ("model.vision_model.model.0.", "model.vision_model."):
                        if k.startswith(pre):
                            name = "vision_model." + k[len(pre):]
                    pairs = [(name, None)]
                for name, t in pairs:
                    if name in known:
                        model.load_tensor(name, sf.get_tensor(k) if t is None else t)
                        seen.add(name)
    conv.finish()
    # a v2 checkpoint may ship without the (unused: SelfSim is "emd" there) SigLIP pooling head
    optional = "vision_model.attn_pool." if model.config.arch == "v2" else "\0"
    missing = [k for k in known if k not in seen and not k.startswith("rope.") and not k.startswith(optional)]
    if missing:
        raise KeyError(f"checkpoint {path} lacks {len(missing)} tensors, e.g. {missing[:4]}")
    model._install_rope_tables()
    model._weights_ready = True


def _load_projector(model: DetikzifyForCausalLM, filename: str):
    """modality_projector file (v1/modeling_detikzify.py:116-121): keys end in .weight / .bias."""
    from safetensors.torch import load_file
    for k, v in load_file(filename).items():
        model.load_tensor("model.mm_projector." + k.split(".")[-1], v)


__all__ = ["load", "DetikzifyConfig", "DetikzifyForCausalLM", "DetikzifyVisionModel", "DetikzifyProcessor",
           "DetikzifyImageProcessor", "BatchFeature", "SyntheticTokenizer", "GenerationConfig", "PRESETS",
           "preset", "v1_models", "v2_models"]
