"""
Deterministic synthetic weights (test infrastructure, see oracle/__init__.py).

No real checkpoints exist offline, so parity runs use seeded synthetic weights
at the real shapes.  This is the numpy twin of `dtk_fill_synthetic` (csrc/kernels_batched.hip,
k_fill_synth): value(i) = offset + scale*sqrt(3)*u(i), u in [-1,1) from a 32-bit integer hash
of (seed, tensor index, i), rounded to bf16 — bit-identical on CPU and GPU (checked by
tests/test_gpu_parity.py::test_synth_weights_bit_exact).  The tensor order below must match
plan() in csrc/dtk_api.hip (the tensor's index is its hash tag).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from .ops import bits_to_f32

WS, BS, NS = 0.02, 0.01, 0.1   # weight / bias / norm-weight scales (norm weights are 1 + NS*u)


def tensor_specs(cfg: dict) -> List[Tuple[str, Tuple[int, ...], float, float]]:
    d, L, ff, V, T = cfg["hidden"], cfg["layers"], cfg["ffn"], cfg["vocab"], cfg["max_positions"]
    D, depth, mlp, p = cfg["vit_dim"], cfg["vit_depth"], cfg["vit_mlp"], cfg["vit_patch"]
    N = (cfg["vit_image"] // p) ** 2
    cc = cfg["concat_patches"]
    kvd = (cfg.get("kv_heads") or cfg["heads"]) * cfg["head_dim"]
    s: List[Tuple[str, Tuple[int, ...], float, float]] = []
    s.append(("model.embed_tokens.weight", (V, d), WS, 0.0))
    for i in range(L):
        q = f"model.layers.{i}."
        s += [
            (q + "input_layernorm.weight", (d,), NS, 1.0),
            (q + "self_attn.q_proj.weight", (d, d), WS, 0.0),
            (q + "self_attn.k_proj.weight", (kvd, d), WS, 0.0),
            (q + "self_attn.v_proj.weight", (kvd, d), WS, 0.0),
            (q + "self_attn.o_proj.weight", (d, d), WS, 0.0),
            (q + "post_attention_layernorm.weight", (d,), NS, 1.0),
            (q + "mlp.gate_proj.weight", (ff, d), WS, 0.0),
            (q + "mlp.up_proj.weight", (ff, d), WS, 0.0),
            (q + "mlp.down_proj.weight", (d, ff), WS, 0.0),
        ]
    s += [
        ("model.norm.weight", (d,), NS, 1.0),
        ("lm_head.weight", (V, d), WS, 0.0),
        ("model.mm_projector.weight", (d, cc * D), WS, 0.0),
    ]
    if cfg.get("proj_bias", True):   # the v2 connector is bias-free: the tensor does not exist (nor its tag)
        s.append(("model.mm_projector.bias", (d,), BS, 0.0))
    s += [
        ("rope.cos", (T, 64), 0.0, 0.0),   # not synthesised (computed), keeps the index aligned
        ("rope.sin", (T, 64), 0.0, 0.0),
    ]
    v = "vision_model."
    s += [
        (v + "patch_embed.proj.weight", (D, 3, p, p), WS, 0.0),
        (v + "patch_embed.proj.bias", (D,), BS, 0.0),
        (v + "pos_embed", (1, N, D), WS, 0.0),
    ]
    for i in range(depth):
        b = v + f"blocks.{i}."
        s += [
            (b + "norm1.weight", (D,), NS, 1.0), (b + "norm1.bias", (D,), BS, 0.0),
            (b + "attn.qkv.weight", (3 * D, D), WS, 0.0), (b + "attn.qkv.bias", (3 * D,), BS, 0.0),
            (b + "attn.proj.weight", (D, D), WS, 0.0), (b + "attn.proj.bias", (D,), BS, 0.0),
            (b + "norm2.weight", (D,), NS, 1.0), (b + "norm2.bias", (D,), BS, 0.0),
            (b + "mlp.fc1.weight", (mlp, D), WS, 0.0), (b + "mlp.fc1.bias", (mlp,), BS, 0.0),
            (b + "mlp.fc2.weight", (D, mlp), WS, 0.0), (b + "mlp.fc2.bias", (D,), BS, 0.0),
        ]
    s += [
        (v + "norm.weight", (D,), NS, 1.0), (v + "norm.bias", (D,), BS, 0.0),
        (v + "attn_pool.latent", (1, 1, D), WS, 0.0),
        (v + "attn_pool.q.weight", (D, D), WS, 0.0), (v + "attn_pool.q.bias", (D,), BS, 0.0),
        (v + "attn_pool.kv.weight", (2 * D, D), WS, 0.0), (v + "attn_pool.kv.bias", (2 * D,), BS, 0.0),
        (v + "attn_pool.proj.weight", (D, D), WS, 0.0), (v + "attn_pool.proj.bias", (D,), BS, 0.0),
        (v + "attn_pool.norm.weight", (D,), NS, 1.0), (v + "attn_pool.norm.bias", (D,), BS, 0.0),
        (v + "attn_pool.mlp.fc1.weight", (mlp, D), WS, 0.0), (v + "attn_pool.mlp.fc1.bias", (mlp,), BS, 0.0),
        (v + "attn_pool.mlp.fc2.weight", (D, mlp), WS, 0.0), (v + "attn_pool.mlp.fc2.bias", (D,), BS, 0.0),
    ]
    return s


def synth_hash(seed: int, tag: int, idx: np.ndarray) -> np.ndarray:
    """uint32 hash, twin of synth_hash() in csrc/kernels_batched.hip (wrap-around arithmetic)."""
    lo, hi = np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        x = idx.astype(np.uint32) * np.uint32(0x9E3779B1) + np.uint32(tag) * np.uint32(0x85EBCA77) + lo
        x ^= x >> np.uint32(16); x *= np.uint32(0x7FEB352D)
        x ^= x >> np.uint32(15); x *= np.uint32(0x846CA68B)
        x ^= x >> np.uint32(16)
        x += hi * np.uint32(0xC2B2AE3D)
        x ^= x >> np.uint32(15); x *= np.uint32(0x2C1B3C6D)
        x ^= x >> np.uint32(12)
    return x


def synth_bits(seed: int, tag: int, n: int, scale: float, offset: float) -> np.ndarray:
    """bf16 bit patterns (uint16) of the n synthetic values of tensor `tag`."""
    h = synth_hash(seed, tag, np.arange(n, dtype=np.uint64))
    u = ((h >> np.uint32(8)).astype(np.int64) - 8388608).astype(np.float32) * np.float32(1.0 / 8388608.0)
    s = np.float32(scale) * np.float32(1.7320508)
    # device: fmaf(u, s, offset) — one rounding; float64 holds u*s exactly (24x24 bits)
    val = (u.astype(np.float64) * np.float64(s) + np.float64(np.float32(offset))).astype(np.float32)
    b = val.view(np.uint32).astype(np.uint64)
    b = b + np.uint64(0x7FFF) + ((b >> np.uint64(16)) & np.uint64(1))   # RNE to bf16
    return (b >> np.uint64(16)).astype(np.uint16)


def make_weights(cfg: dict, seed: int, only_prefix: str = None) -> Dict[str, torch.Tensor]:
    """name -> fp32 tensor of bf16-representable values, checkpoint-shaped."""
    out = {}
    for tag, (name, shape, scale, offset) in enumerate(tensor_specs(cfg)):
        if name.startswith("rope."):
            continue
        if only_prefix is not None and not name.startswith(only_prefix):
            continue
        n = int(np.prod(shape))
        out[name] = bits_to_f32(synth_bits(seed, tag, n, scale, offset)).reshape(shape)
    return out
