"""
LLaMA decoder restated on CPU (test infrastructure, see oracle/__init__.py).

Follows transformers/models/llama/modeling_llama.py (installed 5.15.0; the reference pins
4.52.4 and subclasses LlamaModel at detikzify/model/v1/modeling_detikzify.py:75,203):
  LlamaRMSNorm.forward            :62-67    fp32 stats, cast, THEN multiply by weight
  LlamaRotaryEmbedding / apply    :108-160  rotate-half, cos/sin cast to the activation dtype
  LlamaMLP.forward                :174-176  down(silu(gate(x)) * up(x))
  LlamaAttention.forward          :230-262  q/k/v/o Linear without bias, KV cache append
  LlamaDecoderLayer.forward       :297-324  pre-norm residual blocks
Attention uses the fused (SDPA/flash) rounding: fp32 scores/probabilities, one bf16 rounding
of the head output (reference default attn_implementation: sdpa or flash_attention_2,
examples/infer.py:36).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional

import torch

from .ops import linear, rb


def rmsnorm(x, w, eps, precision="bf16"):
    var = x.pow(2).mean(-1, keepdim=True)
    n = rb(x * torch.rsqrt(var + eps), precision)
    return rb(w * n, precision)


def rope_tables(head_dim: int, theta: float, factor: float, max_pos: int, precision="bf16"):
    """cos/sin [max_pos, head_dim/2]; 'linear' scaling divides inv_freq by factor."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    if factor and factor != 1.0:
        inv_freq = inv_freq / factor
    pos = torch.arange(max_pos, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :]
    return rb(freqs.cos(), precision), rb(freqs.sin(), precision)


def llama3_inv_freq(head_dim: int, theta: float, factor: float = 8.0, low_freq_factor: float = 1.0,
                    high_freq_factor: float = 4.0, original_max_position: int = 8192) -> torch.Tensor:
    """rope_type "llama3" (LLaMA-3.1, the v2 text model): transformers/modeling_rope_utils.py
    _compute_llama3_parameters — wavelengths above original/low_freq_factor are divided by `factor`,
    those below original/high_freq_factor are kept, the band in between is interpolated."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, dtype=torch.int64).float() / head_dim))
    low_freq_wavelen = original_max_position / low_freq_factor
    high_freq_wavelen = original_max_position / high_freq_factor
    wavelen = 2 * math.pi / inv_freq
    inv_llama = torch.where(wavelen > low_freq_wavelen, inv_freq / factor, inv_freq)
    smooth = (original_max_position / wavelen - low_freq_factor) / (high_freq_factor - low_freq_factor)
    smoothed = (1 - smooth) * inv_llama / factor + smooth * inv_llama
    is_medium = ~(wavelen < high_freq_wavelen) * ~(wavelen > low_freq_wavelen)
    return torch.where(is_medium, smoothed, inv_llama)


def rope_tables_from_inv_freq(inv_freq: torch.Tensor, max_pos: int, precision="bf16"):
    pos = torch.arange(max_pos, dtype=torch.float32)
    freqs = pos[:, None] * inv_freq[None, :].float()
    return rb(freqs.cos(), precision), rb(freqs.sin(), precision)


def apply_rope(x, cos, sin, precision="bf16"):
    """x [H, T, hd]; cos/sin [T, hd/2].  q*cos + rotate_half(q)*sin, each product a bf16 tensor."""
    h = x.shape[-1] // 2
    x1, x2 = x[..., :h], x[..., h:]
    o1 = rb(rb(x1 * cos, precision) + rb(-x2 * sin, precision), precision)
    o2 = rb(rb(x2 * cos, precision) + rb(x1 * sin, precision), precision)
    return torch.cat([o1, o2], dim=-1)


def mx_exponent(amax: torch.Tensor) -> torch.Tensor:
    """E8M0 exponent of a group with largest magnitude `amax` (fp32): the smallest e with amax * 2^-e <= 448, read off the bits of
    amax = m * 2^E as e = E - 8 (+ 1 if m > 1.75); 0 for an all-zero group; clamped to -127 .. 126.  The integer arithmetic of
    detikzify_amd/csrc/mx_quant.h::mx_exp, restated."""
    bits = amax.contiguous().view(torch.int32) & 0x7fffffff
    e = (bits >> 23) - 127 - 8 + ((bits & 0x7fffff) > 0x600000).to(torch.int32)
    e = torch.where(bits == 0, torch.zeros_like(e), e)
    return e.clamp(-127, 126)


def mx_quantise(x: torch.Tensor, group: int):
    """x [..., K] (bf16-representable fp32) -> (e4m3 codes as uint8 [..., K], E8M0 bytes [..., K / group]): OCP MXFP8 with the scale
    rule above.  x * 2^-e is exact; the e4m3 rounding is round-to-nearest-even (torch.float8_e4m3fn == v_cvt_pk_fp8_f32)."""
    K = x.shape[-1]
    g = x.float().reshape(*x.shape[:-1], K // group, group)
    e = mx_exponent(g.abs().amax(-1))
    q = torch.ldexp(g, -e[..., None]).to(torch.float8_e4m3fn)
    return q.view(torch.uint8).reshape(x.shape), (e + 127).to(torch.uint8)


def mx_fake_quant(x: torch.Tensor, group: int) -> torch.Tensor:
    """The values the fp8 matrix cores see for x: quantise to MXFP8 in groups of `group` consecutive elements, de-quantise."""
    K = x.shape[-1]
    g = x.float().reshape(*x.shape[:-1], K // group, group)
    e = mx_exponent(g.abs().amax(-1))[..., None]
    q = torch.ldexp(g, -e).to(torch.float8_e4m3fn).float()
    return torch.ldexp(q, e).reshape(x.shape)


def attention(q, k, v, scale, causal_offset: Optional[int] = None, precision="bf16"):
    """q [H,Tq,hd], k/v [H,Tk,hd] (or [KVH,Tk,hd] with H % KVH == 0: GQA, HF repeat_kv).
    causal_offset = absolute position of query 0 (None = full)."""
    if k.shape[0] != q.shape[0]:
        rep = q.shape[0] // k.shape[0]
        k, v = k.repeat_interleave(rep, dim=0), v.repeat_interleave(rep, dim=0)
    s = (q @ k.transpose(-1, -2)) * scale
    if causal_offset is not None:
        Tq, Tk = q.shape[1], k.shape[1]
        qpos = torch.arange(Tq)[:, None] + causal_offset
        kpos = torch.arange(Tk)[None, :]
        s = s.masked_fill(kpos > qpos, float("-inf"))
    p = torch.softmax(s, dim=-1)
    return rb(p @ v, precision)


class LlamaOracle:
    """Weights: dict of HF state-dict names -> fp32 tensors holding bf16-representable values."""

    def __init__(self, cfg: dict, weights: Dict[str, torch.Tensor], precision: str = "bf16"):
        self.cfg, self.w, self.precision = cfg, weights, precision
        # act_quant = True: the four Linear inputs of a layer and the lm_head input are MXFP8 (groups of 32; 16 for the SwiGLU
        # output) — what the batched step of an fp8 model computes on the fp8 matrix cores (csrc/kernels_batch_mx.hip).  The
        # tests switch it on for the batched decode steps only: prefill and the single-sequence step keep bf16 activations.
        self.act_quant = False
        self.d, self.L, self.H = cfg["hidden"], cfg["layers"], cfg["heads"]
        self.hd = cfg["head_dim"]
        self.KVH = cfg.get("kv_heads") or self.H
        self.scale = 1.0 / math.sqrt(self.hd)
        if "rope.cos" in weights:
            self.cos, self.sin = weights["rope.cos"], weights["rope.sin"]
        elif cfg.get("rope_type") == "llama3":
            self.cos, self.sin = rope_tables_from_inv_freq(
                llama3_inv_freq(self.hd, cfg["rope_theta"], cfg.get("rope_factor", 8.0), cfg.get("rope_low_freq_factor", 1.0),
                                cfg.get("rope_high_freq_factor", 4.0), cfg.get("rope_original_max_position", 8192)),
                cfg["max_positions"], precision)
        else:
            self.cos, self.sin = rope_tables(self.hd, cfg["rope_theta"], cfg["rope_factor"],
                                             cfg["max_positions"], precision)
        self.reset()

    def reset(self):
        self.k: List[Optional[torch.Tensor]] = [None] * self.L
        self.v: List[Optional[torch.Tensor]] = [None] * self.L
        self.pos = 0

    def truncate(self, n: int):
        for i in range(self.L):
            if self.k[i] is not None:
                self.k[i], self.v[i] = self.k[i][:, :n], self.v[i][:, :n]
        self.pos = n

    def embed(self, ids: torch.Tensor) -> torch.Tensor:
        return self.w["model.embed_tokens.weight"][ids]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [T, d] input embeddings at positions pos..pos+T-1 -> hidden states [T, d]."""
        P = self.precision
        T = x.shape[0]
        start = self.pos
        cos, sin = self.cos[start:start + T], self.sin[start:start + T]
        for i in range(self.L):
            p = f"model.layers.{i}."
            h = rmsnorm(x, self.w[p + "input_layernorm.weight"], self.cfg["rms_eps"], P)
            if self.act_quant:
                h = mx_fake_quant(h, 32)
            q = linear(h, self.w[p + "self_attn.q_proj.weight"], None, P)
            k = linear(h, self.w[p + "self_attn.k_proj.weight"], None, P)
            v = linear(h, self.w[p + "self_attn.v_proj.weight"], None, P)
            q = q.view(T, self.H, self.hd).transpose(0, 1)
            k = k.view(T, self.KVH, self.hd).transpose(0, 1)
            v = v.view(T, self.KVH, self.hd).transpose(0, 1)
            q, k = apply_rope(q, cos, sin, P), apply_rope(k, cos, sin, P)
            self.k[i] = k if self.k[i] is None else torch.cat([self.k[i], k], dim=1)
            self.v[i] = v if self.v[i] is None else torch.cat([self.v[i], v], dim=1)
            a = attention(q, self.k[i], self.v[i], self.scale, causal_offset=start, precision=P)
            a = a.transpose(0, 1).reshape(T, self.d)
            if self.act_quant:
                a = mx_fake_quant(a, 32)
            x = rb(x + linear(a, self.w[p + "self_attn.o_proj.weight"], None, P), P)
            h = rmsnorm(x, self.w[p + "post_attention_layernorm.weight"], self.cfg["rms_eps"], P)
            if self.act_quant:
                h = mx_fake_quant(h, 32)
            g = linear(h, self.w[p + "mlp.gate_proj.weight"], None, P)
            u = linear(h, self.w[p + "mlp.up_proj.weight"], None, P)
            act = rb(rb(torch.nn.functional.silu(g), P) * u, P)
            if self.act_quant:
                act = mx_fake_quant(act, 16)
            x = rb(x + linear(act, self.w[p + "mlp.down_proj.weight"], None, P), P)
        self.pos = start + T
        return x

    def logits(self, h_last: torch.Tensor) -> torch.Tensor:
        """final norm + lm_head + .float() (reference v1/modeling_detikzify.py:250-257)."""
        h = rmsnorm(h_last, self.w["model.norm.weight"], self.cfg["rms_eps"], self.precision)
        if self.act_quant:
            h = mx_fake_quant(h, 32)
        return linear(h, self.w["lm_head.weight"], None, self.precision)
