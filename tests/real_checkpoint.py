"""
Real-checkpoint parity, ready for the first box that has weights (test infrastructure, nothing here ships).

    DTK_REAL_CKPT=/path/to/nllg-detikzify-ds-7b  scripts/real_parity.sh          (or: pytest tests/test_gpu_real_checkpoint.py -m gpu -s)

`real_parity(path)` loads an HF-layout DeTikZify checkpoint directory (v1: reference detikzify/model/v1/__init__.py:24-56 —
LlamaForCausalLM keys + model.mm_projector.*, the timm tower under model.vision_model.model.0.* / vision_model.* or beside it as
vision_tower.safetensors; v2: detikzify/model/modeling_detikzify.py:119-135 — model.text_model.* / model.connector.* /
model.vision_model.*) twice:

  * on the device through the product loader (detikzify_amd.model.load);
  * on the CPU through the INSTALLED third-party classes, fed straight from the safetensors files — transformers' LlamaForCausalLM
    (the class the reference subclasses, v1/modeling_detikzify.py:203) and, for v2, SiglipVisionModel; for v1 `timm` itself when
    it is importable (the reference's tower, v1/modeling_detikzify.py:94), else the oracle's timm-named restatement (flagged) —
    with the connector and the embedding splice restated here from v1/modeling_detikzify.py:132-137,158-189 and
    modeling_detikzify.py:62-86.  Nothing on the CPU side goes through the product's name conversion or kernels.

and compares: vision features, prefill logits (last position), `n_tokens` greedy tokens (teacher-forced with the device's tokens,
near-tie rule of tests/test_gpu_parity.py).  For v1 it also settles the open question of SURVEY 8c — is the timm tower's GELU the
exact (erf) or the tanh-approximated one: with timm importable the tower itself says; without it the only evidence is the
checkpoint's own behaviour, so both flavours are run on the device and the mean log-probability the decoder gives its own greedy
tokens is reported for each (the flavour the weights were trained with is the more confident one) — printed as a proxy, not a verdict.
"""
from __future__ import annotations

import json
import math
import time
from pathlib import Path
from typing import Dict, Optional

import torch

ULP = 2.0 ** -7


def read_checkpoint(path: Path) -> Dict[str, torch.Tensor]:
    from safetensors import safe_open
    out: Dict[str, torch.Tensor] = {}
    for f in sorted(path.glob("*.safetensors")):
        with safe_open(str(f), framework="pt") as sf:
            for k in sf.keys():
                out[("tower::" if f.name == "vision_tower.safetensors" else "") + k] = sf.get_tensor(k)
    return out


def _hf_decoder(text_cfg: dict, ck: Dict[str, torch.Tensor], prefix: str, dtype=torch.bfloat16):
    """installed transformers LlamaForCausalLM over the checkpoint's own tensors (no product code in between)"""
    from transformers import LlamaConfig, LlamaForCausalLM
    from transformers.models.llama.modeling_llama import LlamaRotaryEmbedding
    hc = LlamaConfig.from_dict({k: v for k, v in text_cfg.items() if k not in ("model_type", "architectures", "transformers_version")})
    hc.tie_word_embeddings = False
    with torch.device("meta"):
        hf = LlamaForCausalLM(hc)
    sd = {}
    for k in hf.state_dict():
        src = "lm_head.weight" if k == "lm_head.weight" else prefix + k[len("model."):]
        sd[k] = ck[src].to(dtype)
    hf.load_state_dict(sd, strict=True, assign=True)
    hf.model.rotary_emb = LlamaRotaryEmbedding(hc)
    return hf.eval()


def _tower_tensors(ck: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """the v1 tower's tensors under bare timm names, wherever the checkpoint keeps them"""
    out = {}
    for k, v in ck.items():
        for pre in ("tower::visual.trunk.", "tower::", "model.vision_model.model.0.", "model.vision_model.", "vision_model."):
            if k.startswith(pre):
                if pre == "tower::" and k.startswith(("tower::text.", "tower::logit_")):
                    break
                out[k[len(pre):]] = v
                break
    return out


def _cpu_features_v1(cfg: dict, ck, px: torch.Tensor, gelu_tanh: int):
    """[N, D] fp32 features of get_intermediate_layers(n=[feature_layer], norm=True) (v1/modeling_detikzify.py:63-72) and who computed them"""
    tower = {k: v.float() for k, v in _tower_tensors(ck).items()}
    try:
        import timm  # noqa: F401
        have_timm = True
    except Exception:  # noqa: BLE001
        have_timm = False
    if have_timm and cfg["vit_dim"] == 1152:
        import timm
        m = timm.create_model("vit_so400m_patch14_siglip_384", pretrained=False, num_classes=0)
        m.load_state_dict(tower, strict=False)
        m.eval()
        with torch.no_grad():
            feats = m.get_intermediate_layers(px[None].float(), n=[cfg["vit_feature_layer"]], norm=True)[0][0]
        act = type(m.blocks[0].mlp.act).__name__ + (f"(approximate={getattr(m.blocks[0].mlp.act, 'approximate', 'none')})")
        return feats, f"timm {timm.__version__}", act
    from oracle.vit import VitOracle
    c = dict(cfg, vit_gelu_tanh=gelu_tanh)
    w = {"vision_model." + k: v for k, v in tower.items()}
    return VitOracle(c, w, "fp32").intermediate(px.float(), cfg["vit_feature_layer"]), "oracle/vit.py restatement (timm not installed)", None


def _cpu_features_v2(cfgj: dict, ck, px: torch.Tensor):
    from transformers import SiglipVisionConfig, SiglipVisionModel
    vc = SiglipVisionConfig.from_dict({k: v for k, v in cfgj["vision_config"].items() if k not in ("model_type", "architectures", "transformers_version")})
    m = SiglipVisionModel(vc)           # (not on the meta device: its position_ids buffer is not in the state dict)
    sd = {}
    for k in m.state_dict():
        bare = k[len("vision_model."):] if k.startswith("vision_model.") else k
        for cand in ("model.vision_model." + k, "model.vision_model." + bare, "model.vision_model.vision_model." + bare):
            if cand in ck:
                sd[k] = ck[cand].float()
                break
        else:
            if "head." in k:        # a v2 checkpoint may ship without the (unused) pooling head
                sd[k] = torch.zeros(m.state_dict()[k].shape)
            else:
                raise KeyError(f"no checkpoint tensor for the HF SigLIP parameter {k}")
    m.load_state_dict(sd, strict=True)
    m.eval()
    import transformers
    with torch.no_grad():
        out = m(pixel_values=px[None].float())
    return out.last_hidden_state[0], f"transformers {transformers.__version__} SiglipVisionModel", vc.hidden_act


def rel_l2(a, b) -> float:
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / (b.norm() + 1e-30))


def real_parity(path: str, n_tokens: int = 32, image=None, device: int = 0, settle_gelu: bool = True) -> Dict[str, object]:
    from detikzify_amd.model import load
    from detikzify_amd.model.config import DetikzifyConfig
    t0 = time.perf_counter()
    p = Path(path)
    cfgj = json.loads((p / "config.json").read_text())
    c = DetikzifyConfig.from_hf_json(str(p / "config.json"))
    cfg = c.oracle_dict()
    arch = c.arch
    model, proc = load(str(p), device_map=device, max_positions=min(c.max_positions, 1024))
    if image is None:
        from PIL import Image
        sketch = Path(__file__).resolve().parents[1] / "examples" / "sketch.png"
        image = Image.open(sketch).convert("RGB")
    enc = proc(images=image, return_tensors="pt")
    ids, px = enc.input_ids[0], enc.pixel_values
    img_tok, eos = model.config.image_token_id, model.config.text_config.eos_token_id
    eos_ids = list(eos) if isinstance(eos, (list, tuple)) else [eos]
    ck = read_checkpoint(p)

    # ---- vision tower
    dev_feats, _ = model.vit_encode(px, want_pooled=False)
    if arch == "v2":
        cpu_feats, tower_by, act = _cpu_features_v2(cfgj, ck, px[0])
    else:
        cpu_feats, tower_by, act = _cpu_features_v1(cfg, ck, px[0], c.vit_gelu_tanh)
    r_feats = rel_l2(dev_feats[0].float(), cpu_feats)

    # ---- connector + splice on the CPU (v1/modeling_detikzify.py:132-137,158-189; modeling_detikzify.py:62-86), decoder by HF
    n, cc = c.num_patches, c.concat_patches
    feats = cpu_feats[-n * cc:].reshape(n, cpu_feats.shape[-1] * cc).to(torch.bfloat16)
    if arch == "v2":
        w_proj, b_proj, text_cfg, prefix = ck["model.connector.modality_projection.proj.weight"], None, cfgj["text_config"], "model.text_model."
    else:
        w_proj, b_proj, text_cfg, prefix = ck["model.mm_projector.weight"], ck.get("model.mm_projector.bias"), cfgj, "model."
    img_emb = torch.nn.functional.linear(feats, w_proj.to(torch.bfloat16), None if b_proj is None else b_proj.to(torch.bfloat16))
    hf = _hf_decoder(text_cfg, ck, prefix)
    with torch.no_grad():
        emb = hf.model.embed_tokens(ids[None])[0]
        where = torch.where(ids == img_tok)[0]
        s = int(where[0])
        emb = torch.cat([emb[:s], img_emb, emb[s + n:]], 0)
        res = hf(inputs_embeds=emb[None], use_cache=True)
    cpu_logits = res.logits[0, -1].float()
    dev_logits = model.prefill(ids, px, return_logits=True)
    r_logits = rel_l2(dev_logits, cpu_logits)

    # ---- greedy tokens: the device decodes, the CPU is teacher-forced with its tokens
    out = model.generate(input_ids=ids[None], pixel_values=px, do_sample=False, max_new_tokens=n_tokens, bad_words_ids=[[img_tok]],
                         begin_suppress_tokens=eos_ids)
    toks = out[0, ids.numel():].tolist()
    kv, logits, same, near, logp = res.past_key_values, cpu_logits, 0, 0, 0.0
    with torch.no_grad():
        for i, t in enumerate(toks):
            sc = logits.clone()
            sc[img_tok] = float("-inf")
            if i == 0:
                for e in eos_ids:
                    sc[e] = float("-inf")
            mine = int(sc.argmax())
            logp += float(torch.log_softmax(sc, -1)[t])
            if mine == t:
                same += 1
            else:
                top2 = torch.topk(sc, 2)[0]
                near += int(float(top2[0] - top2[1]) <= 2 * float(top2[0].abs()) * ULP + 1e-6)
            if i + 1 < len(toks):
                r = hf(input_ids=torch.tensor([[t]]), past_key_values=kv, use_cache=True)
                kv, logits = r.past_key_values, r.logits[0, -1].float()
    report = {"path": str(p), "arch": arch, "tower_reference": tower_by, "tower_activation": act,
              "device_gelu": "tanh" if c.vit_gelu_tanh else "erf", "feats_rel_l2": r_feats, "prefill_logits_rel_l2": r_logits,
              "greedy_identical": same, "greedy_near_tie_flips": near, "greedy_tokens": len(toks),
              "cpu_mean_logprob_of_device_tokens": logp / max(1, len(toks)), "text": proc.decode(toks, skip_special_tokens=True)[:120]}
    del hf, res, kv

    # ---- v1: which GELU does the tower want?
    if arch == "v1" and settle_gelu:
        if act is not None:                 # timm itself answered
            report["gelu_verdict"] = f"timm's tower uses {act}"
        else:
            conf = {}
            for flavour in (0, 1):
                m2, _ = load(str(p), device_map=device, max_positions=min(c.max_positions, 1024), vit_gelu_tanh=flavour)
                lg = m2.prefill(ids, px, return_logits=True)
                m2.set_sampling(do_sample=False, bad_ids=[img_tok], begin_suppress_ids=eos_ids)
                lp = 0.0
                for i in range(min(16, n_tokens)):
                    m2.decode_launch()
                    t = m2.decode_wait()
                    sc = lg.clone()
                    sc[img_tok] = float("-inf")
                    lp += float(torch.log_softmax(sc, -1)[t])
                    lg = m2.get_logits()
                conf["tanh" if flavour else "erf"] = lp / min(16, n_tokens)
                del m2
            report["gelu_proxy_mean_logprob_of_own_greedy_tokens"] = conf
            report["gelu_verdict"] = ("timm not installed: proxy only — the decoder is more confident after the "
                                      f"{'tanh' if conf['tanh'] > conf['erf'] else 'erf'} tower ({conf})")
    report["seconds"] = round(time.perf_counter() - t0, 1)
    return report


def one_line(r: Dict[str, object]) -> str:
    return (f"REAL-CHECKPOINT PARITY {r['path']} [{r['arch']}, tower reference: {r['tower_reference']}, device GELU {r['device_gelu']}]: "
            f"vision features rel-L2 {r['feats_rel_l2']:.2e}; prefill logits vs HF-on-CPU (bf16) rel-L2 {r['prefill_logits_rel_l2']:.2e}; "
            f"greedy {r['greedy_identical']}/{r['greedy_tokens']} identical ({r['greedy_near_tie_flips']} of the rest at near-ties); "
            f"CPU mean log-prob of the device's tokens {r['cpu_mean_logprob_of_device_tokens']:.3f}; "
            f"{r.get('gelu_verdict', 'GELU: stated by the checkpoint (v2 vision_config.hidden_act)')}; {r['seconds']} s")
