"""
The (model, processor) pair's model half: an object with the attribute surface DeTikZify's
inference code touches, over the C ABI of libdtk_hip.so.

Replaces (reference, all Python): DetikzifyForCausalLM / DetikzifyModel / DetikzifyVisionModel
(detikzify/model/v1/modeling_detikzify.py:49-305) and the HF GenerationMixin.generate loop they
inherit (called at detikzify/infer/generate.py:218-227).  Surface kept:
  model.generate(input_ids, bad_words_ids, begin_suppress_tokens, pixel_values, streamer,
                 stopping_criteria, temperature, top_p, top_k, max_length, do_sample, ...)
  model.device / .dtype / .name_or_path / .generation_config.to_dict() / .config.{image_token_id,
  text_config.eos_token_id, pooling_mode} / model.model.vision_model(pixel_values=...)
There is no CPU fallback: constructing the model without the HIP library or a GPU raises.
"""
from __future__ import annotations

import math
import threading
import ctypes as C
import hashlib
from types import SimpleNamespace
from typing import Any, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

from .. import _lib
from .config import DetikzifyConfig


class GenerationConfig:
    """transformers.GenerationConfig stand-in: only `.to_dict()` and attribute reads are used
    (infer/generate.py:211; v1/__init__.py:41)."""

    def __init__(self, **kw):
        self.max_length = 20           # HF default; the pipeline overrides it (generate.py:383)
        self.max_new_tokens = None
        self.do_sample = False
        self.temperature = 1.0
        self.top_p = 1.0
        self.top_k = 50
        self.eos_token_id = None
        self.pad_token_id = None
        self.bos_token_id = None
        self.__dict__.update(kw)

    def to_dict(self) -> Dict[str, Any]:
        return dict(self.__dict__)

    def update_from_dict(self, values: Dict[str, Any]) -> None:
        """generation_config.json of a checkpoint (what from_pretrained loads into model.generation_config); bookkeeping
        keys ("_from_model_config", "transformers_version") are dropped"""
        self.__dict__.update({k: v for k, v in values.items() if not k.startswith("_") and k != "transformers_version"})


class VisionOutput(SimpleNamespace):
    """BaseModelOutputWithPoolingAndNoAttention stand-in (last_hidden_state, pooler_output)."""


class DetikzifyVisionModel:
    """model.model.vision_model: timm ViT forward_features + forward_head on the GPU
    (reference v1/modeling_detikzify.py:63-69)."""

    def __init__(self, owner: "DetikzifyForCausalLM"):
        self._owner = owner
        self._pool_lock = threading.Lock()
        self._pool_queue: List[Any] = []        # [pixels, done event, result | exception] of threads waiting for a pooled output
        self._pool_leader = False

    def __call__(self, pixel_values: torch.Tensor, **_) -> VisionOutput:
        return self.forward(pixel_values)

    def forward(self, pixel_values: torch.Tensor) -> VisionOutput:
        feats, pooled = self._owner.vit_encode(pixel_values, want_pooled=True)
        return VisionOutput(last_hidden_state=feats, pooler_output=pooled)

    def pooled_only(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pooler_output without copying the 729 x 1152 patch features back (SelfSim "cos": the reward's hot call).

        Calls that arrive from several threads at once (the trees of a parallel search scoring their rollouts) are combined:
        the first caller becomes the leader and encodes whatever has queued up — up to DTK_VIT_BATCH images per pass over the
        tower, 2.7 ms per image instead of 4.1 — until the queue is empty; the others sleep until their result is in.  Per image
        the result is bit-identical to a call of its own (dtk_vit_encode's batch rows are independent)."""
        px = pixel_values.detach().to("cpu", torch.float32)
        if px.dim() == 3:
            px = px[None]
        if px.shape[0] != 1:
            return self._owner.vit_encode(px, want_pooled=True, want_feats=False)[1]
        item = [px, threading.Event(), None]
        with self._pool_lock:
            self._pool_queue.append(item)
            lead = not self._pool_leader
            if lead:
                self._pool_leader = True
        if lead:
            while True:
                with self._pool_lock:
                    batch, self._pool_queue = self._pool_queue[:_lib.DTK_VIT_BATCH], self._pool_queue[_lib.DTK_VIT_BATCH:]
                    if not batch:
                        self._pool_leader = False
                        break
                try:
                    out = self._owner.vit_encode(torch.cat([it[0] for it in batch]), want_pooled=True, want_feats=False)[1]
                    for k, it in enumerate(batch):
                        it[2] = out[k:k + 1].clone()
                except BaseException as e:  # noqa: BLE001  (every waiter gets the error; the leader re-raises its own below)
                    for it in batch:
                        it[2] = e
                for it in batch:
                    it[1].set()
        item[1].wait()
        if isinstance(item[2], BaseException):
            raise item[2]
        return item[2]

    def get_intermediate_layers(self, pixel_values: torch.Tensor, *_, **__):
        feats, _ = self._owner.vit_encode(pixel_values, want_pooled=False)
        return [feats]


# HF generate() arguments this path does not implement, each with the value(s) at which HF's _sample does exactly what
# this loop does.  Passing one of them at such a value is harmless (the reference's callers and HF's own defaults do);
# any other value, and any name that is neither here nor a parameter of generate(), raises.
_NEUTRAL_GENERATE_KWARGS: Dict[str, tuple] = {
    "attention_mask": "any", "use_cache": "any", "pad_token_id": "any", "bos_token_id": "any", "synced_gpus": "any",
    "return_dict_in_generate": (None, False), "output_scores": (None, False), "output_logits": (None, False),
    "output_attentions": (None, False), "output_hidden_states": (None, False),
    "num_beams": (None, 1), "num_beam_groups": (None, 1), "num_return_sequences": (None, 1),
    "repetition_penalty": (None, 1.0), "encoder_repetition_penalty": (None, 1.0), "length_penalty": (None, 1.0),
    "diversity_penalty": (None, 0.0), "no_repeat_ngram_size": (None, 0), "encoder_no_repeat_ngram_size": (None, 0),
    "min_length": (None, 0), "min_new_tokens": (None, 0), "typical_p": (None, 1.0), "min_p": (None, 0.0),
    "epsilon_cutoff": (None, 0.0), "eta_cutoff": (None, 0.0), "penalty_alpha": (None, 0.0), "early_stopping": (None, False),
    "renormalize_logits": (None, False), "remove_invalid_values": (None, False), "guidance_scale": (None, 1.0),
    "forced_bos_token_id": (None,), "forced_eos_token_id": (None,), "exponential_decay_length_penalty": (None,),
    "sequence_bias": (None,), "logits_processor": (None, [], ()), "prefix_allowed_tokens_fn": (None,),
    "assistant_model": (None,), "negative_prompt_ids": (None,), "negative_prompt_attention_mask": (None,),
    "generation_config": (None,), "stop_strings": (None,), "max_time": (None,), "cache_implementation": (None,),
    "past_key_values": (None,), "inputs_embeds": (None,), "tokenizer": "any",
}


def _is_neutral(value: Any, neutral: Any) -> bool:
    if neutral is None:
        return value is None
    if isinstance(neutral, (list, tuple)):          # "no extra processors"
        return isinstance(value, (list, tuple)) and len(value) == 0
    return isinstance(value, (bool, int, float)) and value == neutral


def _reject_unsupported_generate_kwargs(kw: Dict[str, Any]) -> None:
    for name, value in kw.items():
        allowed = _NEUTRAL_GENERATE_KWARGS.get(name)
        if allowed is None:
            raise TypeError(f"generate() got an argument this decoder does not implement: {name!r}")
        if allowed != "any" and not any(_is_neutral(value, a) for a in allowed):
            raise NotImplementedError(
                f"generate({name}={value!r}) is not supported: this path implements greedy / temperature / top-k / top-p "
                f"sampling of one sequence (the calls DetikzifyGenerator.generate makes); only {name} in {allowed} is accepted")


def _flat_ids(values) -> List[int]:
    """[id, [id, id], ...] -> flat list of ints (an eos_token_id may be a list in HF configs)"""
    out: List[int] = []
    for v in values or ():
        if isinstance(v, (list, tuple)):
            out.extend(int(x) for x in v)
        elif v is not None:
            out.append(int(v))
    return out


def _bf16_tensor_from_bits(bits: np.ndarray) -> torch.Tensor:
    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


class DetikzifyForCausalLM:
    def __init__(self, config: DetikzifyConfig, device_index: int = 0):
        self.config = config
        self.lib = _lib.load_library()
        self.hip_device = int(device_index)
        kd = config.kernel_dict()
        cc = _lib.DtkConfig(**kd)
        cc.reserved[0] = int(getattr(config, "batch_slots", 0) or 0)
        cc.reserved[1] = 1 if getattr(config, "weight_format", "bf16") == "fp8" else 0
        cc.reserved[2] = int(getattr(config, "kv_heads", 0) or 0)                       # GQA (v2)
        cc.reserved[3] = 0 if getattr(config, "proj_bias", True) else _lib.DTK_ARCH_PROJ_NO_BIAS
        ctx = C.c_void_p()
        rc = self.lib.dtk_create(C.byref(cc), self.hip_device, C.byref(ctx))
        if rc != 0:
            msg = self.lib.dtk_last_error(None)
            raise _lib.DtkError(f"dtk_create failed ({rc}): {msg.decode() if msg else ''} — "
                                "detikzify_amd needs an MI355X-class GPU; there is no CPU path")
        self._ctx = ctx
        self.generation_config = GenerationConfig(
            eos_token_id=config.eos_token_id, pad_token_id=config.pad_token_id,
            bos_token_id=config.bos_token_id)
        self.name_or_path = config.name_or_path
        self.model = SimpleNamespace(vision_model=DetikzifyVisionModel(self))
        self.reuse_prefix = False     # output-identical KV/image reuse across rollouts
        self.batch_engine = None      # set by infer.batching.BatchEngine: generate() then decodes in a slot
        # ViT passes (SelfSim reward) run on their own HIP stream and overlap with the decode steps of other sequences; they
        # share activation buffers with the image branch of a prefill, so those two are serialised by this lock (a reward
        # never takes the batch engine's lock: the trees that are decoding keep stepping)
        self._vit_lock = threading.RLock()
        self._single_busy = threading.Lock()     # held by a generate() that decodes on the context's single sequence
        self._weights_ready = False

    # ---- HF-shaped attributes ---------------------------------------------------------------
    @property
    def device(self) -> torch.device:
        # ids / pixels cross the C ABI as HOST buffers, so tensors handed to this model live on
        # the CPU; the GPU is self.hip_device.
        return torch.device("cpu")

    @property
    def dtype(self) -> torch.dtype:
        return torch.bfloat16

    def eval(self):
        return self

    def get_model(self):
        return self.model

    def __del__(self):
        try:
            if getattr(self, "_ctx", None):
                self.lib.dtk_destroy(self._ctx)
                self._ctx = None
        except Exception:
            pass

    def _check(self, rc, what):
        _lib.check(self.lib, self._ctx, rc, what)

    # ---- weights ------------------------------------------------------------------------------
    def tensor_names(self) -> List[str]:
        n = self.lib.dtk_num_tensors(self._ctx)
        return [self.lib.dtk_tensor_name(self._ctx, i).decode() for i in range(n)]

    def load_tensor(self, name: str, t: torch.Tensor):
        t = t.detach().cpu().contiguous()
        if t.dtype == torch.bfloat16:
            arr, dt = t.view(torch.int16).numpy(), _lib.DTK_BF16
        elif t.dtype == torch.float16:
            arr, dt = t.view(torch.int16).numpy(), _lib.DTK_F16
        else:
            arr, dt = t.float().numpy(), _lib.DTK_F32
        shape = (C.c_int64 * t.dim())(*t.shape) if t.dim() else (C.c_int64 * 1)(1)
        self._check(self.lib.dtk_load_tensor(self._ctx, name.encode(), arr.ctypes.data_as(C.c_void_p), dt,
                                             shape, max(t.dim(), 1)), f"dtk_load_tensor({name})")

    def load_state_dict(self, state: Dict[str, torch.Tensor], strict: bool = True):
        known = set(self.tensor_names())
        missing = sorted(k for k in known if k not in state and not k.startswith("rope."))
        unexpected = sorted(k for k in state if k not in known)
        if strict and (missing or unexpected):
            raise KeyError(f"missing: {missing[:5]}... unexpected: {unexpected[:5]}...")
        for k, v in state.items():
            if k in known:
                self.load_tensor(k, v)
        self._install_rope_tables()
        self._weights_ready = True
        return SimpleNamespace(missing_keys=missing, unexpected_keys=unexpected)

    def read_tensor(self, name: str) -> torch.Tensor:
        """stored bf16 tensor -> flat torch.bfloat16 (tests / CPU baseline)"""
        n = self.lib.dtk_tensor_numel(self._ctx, name.encode())
        if n < 0:
            raise KeyError(name)
        buf = np.empty(n, dtype=np.uint16)
        self._check(self.lib.dtk_read_tensor(self._ctx, name.encode(), buf.ctypes.data_as(C.c_void_p), n),
                    f"dtk_read_tensor({name})")
        return _bf16_tensor_from_bits(buf)

    def fill_synthetic(self, seed: int = 1234):
        self._check(self.lib.dtk_fill_synthetic(self._ctx, C.c_uint64(seed)), "dtk_fill_synthetic")
        self._install_rope_tables()
        self._weights_ready = True

    def _install_rope_tables(self):
        """cos/sin exactly as HF LlamaRotaryEmbedding computes them (modeling_llama.py:108-140):
        fp32 inv_freq (linear scaling: / factor; "llama3": modeling_rope_utils._compute_llama3_parameters),
        fp32 pos*inv_freq, cos/sin cast to bf16."""
        c = self.config
        inv = 1.0 / (c.rope_theta ** (torch.arange(0, c.head_dim, 2, dtype=torch.int64).float() / c.head_dim))
        if getattr(c, "rope_type", "linear") == "llama3":
            low_wl = c.rope_original_max_position / c.rope_low_freq_factor
            high_wl = c.rope_original_max_position / c.rope_high_freq_factor
            wavelen = 2 * math.pi / inv
            scaled = torch.where(wavelen > low_wl, inv / c.rope_factor, inv)
            smooth = (c.rope_original_max_position / wavelen - c.rope_low_freq_factor) / (c.rope_high_freq_factor - c.rope_low_freq_factor)
            mid = (1 - smooth) * scaled / c.rope_factor + smooth * scaled
            inv = torch.where((wavelen <= low_wl) & (wavelen >= high_wl), mid, scaled)
        elif c.rope_factor and c.rope_factor != 1.0:
            inv = inv / c.rope_factor
        freqs = torch.arange(c.max_positions, dtype=torch.float32)[:, None] * inv[None, :]
        self.load_tensor("rope.cos", freqs.cos().to(torch.bfloat16))
        self.load_tensor("rope.sin", freqs.sin().to(torch.bfloat16))

    # ---- vision tower -------------------------------------------------------------------------
    def vit_encode(self, pixel_values: torch.Tensor, want_pooled: bool = True, want_feats: bool = True):
        px = pixel_values.detach().to("cpu", torch.float32).contiguous()
        if px.dim() == 3:
            px = px[None]
        B = px.shape[0]
        c = self.config
        n = (c.vit_image // c.vit_patch) ** 2
        feats = np.empty((B, n, c.vit_dim), dtype=np.uint16) if want_feats else None
        pooled = np.empty((B, c.vit_dim), dtype=np.uint16)
        with self._vit_lock:
            self._check(self.lib.dtk_vit_encode(
                self._ctx, px.numpy().ctypes.data_as(C.c_void_p), B, feats.ctypes.data_as(C.c_void_p) if want_feats else None,
                pooled.ctypes.data_as(C.c_void_p) if want_pooled else None), "dtk_vit_encode")
        f = _bf16_tensor_from_bits(feats.reshape(-1)).view(B, n, c.vit_dim) if want_feats else None
        p = _bf16_tensor_from_bits(pooled.reshape(-1)).view(B, c.vit_dim) if want_pooled else None
        return f, p

    # ---- decoder ------------------------------------------------------------------------------
    def prefill(self, input_ids: torch.Tensor, pixel_values: Optional[torch.Tensor] = None,
                return_logits: bool = False, reuse: Optional[bool] = None, slot: Optional[int] = None) -> Optional[torch.Tensor]:
        ids = input_ids.detach().to("cpu", torch.int64).reshape(-1).contiguous()
        T = ids.numel()
        px_ptr, key = None, 0
        if pixel_values is not None:
            px = pixel_values.detach().to("cpu", torch.float32).contiguous()
            if px.dim() == 4:
                if px.shape[0] != 1:
                    raise ValueError("batch size 1 only")
                px = px[0]
            self._px_keepalive = px
            px_ptr = px.numpy().ctypes.data_as(C.c_void_p)
            key = self.image_key(pixel_values)
        reuse = self.reuse_prefix if reuse is None else reuse
        flags = (_lib.DTK_PREFILL_REUSE_PREFIX | _lib.DTK_PREFILL_REUSE_IMAGE) if reuse else 0
        logits = np.empty(self.config.vocab, dtype=np.float32) if return_logits else None
        lp = logits.ctypes.data_as(C.c_void_p) if return_logits else None
        with self._vit_lock:    # the image branch shares the ViT buffers with vit_encode (which runs on its own stream)
            if slot is None:
                self._check(self.lib.dtk_prefill(self._ctx, ids.numpy().ctypes.data_as(C.c_void_p), T, px_ptr,
                                                 C.c_uint64(key), flags, lp), "dtk_prefill")
            else:
                self._check(self.lib.dtk_prefill_slot(self._ctx, int(slot), ids.numpy().ctypes.data_as(C.c_void_p), T, px_ptr,
                                                      C.c_uint64(key), flags, lp), "dtk_prefill_slot")
        return torch.from_numpy(logits) if return_logits else None

    def set_sampling(self, do_sample=False, temperature=1.0, top_p=1.0, top_k=0, seed=0,
                     bad_ids: Iterable[int] = (), begin_suppress_ids: Iterable[int] = (),
                     always_suppress_ids: Iterable[int] = (), slot: Optional[int] = None):
        s = _lib.DtkSampling()
        s.do_sample, s.temperature, s.top_p, s.top_k = int(bool(do_sample)), float(temperature), float(top_p), int(top_k or 0)
        s.seed = int(seed) & ((1 << 64) - 1)
        for field, cnt, vals in (("bad_ids", "n_bad", bad_ids), ("begin_suppress_ids", "n_begin_suppress", begin_suppress_ids),
                                 ("always_suppress_ids", "n_always_suppress", always_suppress_ids)):
            vals = [int(v) for v in vals]
            if len(vals) > 8:
                raise ValueError("at most 8 ids per suppression list")
            setattr(s, cnt, len(vals))
            arr = getattr(s, field)
            for i, v in enumerate(vals):
                arr[i] = v
        if slot is None:
            self._check(self.lib.dtk_set_sampling(self._ctx, C.byref(s)), "dtk_set_sampling")
        else:
            self._check(self.lib.dtk_set_sampling_slot(self._ctx, int(slot), C.byref(s)), "dtk_set_sampling_slot")

    # ---- batched decode (independent rollouts share one pass over the weights) -----------------------
    def num_slots(self) -> int:
        return int(self.lib.dtk_num_slots(self._ctx))

    def max_decode_slots(self) -> int:
        """slots 0..n-1 that may take part in a decode step: 4 (contexts with <= 5 slots: multi-vector kernels), 16, 32 or 64
        (one, two, four MFMA column tiles), never more than num_slots()"""
        return int(self.lib.dtk_max_decode_slots(self._ctx))

    def decode_batch_launch(self, active_slots: Iterable[int]):
        arr = (C.c_int32 * _lib.DTK_MAX_BATCH)()
        for j in active_slots:
            arr[int(j)] = 1
        self._check(self.lib.dtk_decode_batch_launch(self._ctx, arr), "dtk_decode_batch_launch")

    def decode_batch_wait(self) -> List[int]:
        out = (C.c_int64 * _lib.DTK_MAX_BATCH)()
        self._check(self.lib.dtk_decode_batch_wait(self._ctx, out), "dtk_decode_batch_wait")
        return [int(v) for v in out]

    def kv_fork(self, src_slot: int, dst_slot: int, n_tokens: int):
        self._check(self.lib.dtk_kv_fork(self._ctx, int(src_slot), int(dst_slot), int(n_tokens)), "dtk_kv_fork")

    _image_keys: Dict[Tuple[int, int, int], Tuple[Any, int]] = {}     # (storage address, elements, tensor version) -> (tensor, key)
    _image_keys_lock = threading.Lock()     # every tree thread of a parallel search comes through image_key()

    @classmethod
    def image_key(cls, pixel_values: torch.Tensor) -> int:
        """content hash of the pixels (the C side keys cached image prefixes by it).  Hashing 1.8 MB costs ~2 ms under the GIL and
        the 64 trees of a parallel search all present the SAME tensor object: memoised per tensor (the entry keeps the tensor
        alive, so its address cannot be reused by another one; the version counter catches in-place edits)"""
        # (inference-mode tensors keep no version counter: an in-place edit of one between two calls would go unnoticed —
        # processor outputs are never edited)
        ident = (pixel_values.data_ptr(), pixel_values.numel(), -1 if pixel_values.is_inference() else pixel_values._version)
        with cls._image_keys_lock:
            hit = cls._image_keys.get(ident)
        if hit is not None and hit[0] is pixel_values:
            return hit[1]
        px = pixel_values.detach().to("cpu", torch.float32).contiguous()
        key = int.from_bytes(hashlib.blake2b(px.numpy().tobytes(), digest_size=8).digest(), "little")    # (outside the lock: 2 ms)
        with cls._image_keys_lock:
            while len(cls._image_keys) >= 64:       # oldest entry out (dicts keep insertion order); at most 64 pixel tensors stay alive
                cls._image_keys.pop(next(iter(cls._image_keys)))
            cls._image_keys[ident] = (pixel_values, key)
        return key

    def slot_lcp(self, slot: int, ids: torch.Tensor, key: int = 0) -> int:
        ids = ids.detach().to("cpu", torch.int64).reshape(-1).contiguous()
        out = C.c_int(0)
        self._check(self.lib.dtk_slot_lcp(self._ctx, int(slot), ids.numpy().ctypes.data_as(C.c_void_p), ids.numel(), C.c_uint64(key),
                                          C.byref(out)), "dtk_slot_lcp")
        return int(out.value)

    def best_lcp_slot(self, slots: Iterable[int], ids: torch.Tensor, key: int = 0) -> Optional[Tuple[int, int]]:
        """(slot, lcp) of the slot among `slots` whose cache shares the longest prefix with ids (lowest index on ties), None if
        none shares a token"""
        ids = ids.detach().to("cpu", torch.int64).reshape(-1).contiguous()
        ptr, n, out, best = ids.numpy().ctypes.data_as(C.c_void_p), ids.numel(), C.c_int(0), None
        for s_ in slots:
            self._check(self.lib.dtk_slot_lcp(self._ctx, int(s_), ptr, n, C.c_uint64(key), C.byref(out)), "dtk_slot_lcp")
            if out.value > (best[1] if best else 0):
                best = (int(s_), int(out.value))
        return best

    def cached_ids(self, slot: int, n_max: int = 4096) -> List[int]:
        out = (C.c_int64 * n_max)()
        n = self.lib.dtk_slot_cached_ids(self._ctx, int(slot), out, n_max)
        return [int(out[i]) for i in range(max(0, n))]

    def resume_slot(self, slot: int, ids: torch.Tensor, key: int = 0):
        ids = ids.detach().to("cpu", torch.int64).reshape(-1).contiguous()
        self._check(self.lib.dtk_resume_slot(self._ctx, int(slot), ids.numpy().ctypes.data_as(C.c_void_p), ids.numel(), C.c_uint64(key)),
                    "dtk_resume_slot")

    def get_logits_slot(self, slot: int) -> torch.Tensor:
        out = np.empty(self.config.vocab, dtype=np.float32)
        self._check(self.lib.dtk_get_logits_slot(self._ctx, int(slot), out.ctypes.data_as(C.c_void_p)), "dtk_get_logits_slot")
        return torch.from_numpy(out)

    def context_len_slot(self, slot: int) -> int:
        return int(self.lib.dtk_context_len_slot(self._ctx, int(slot)))

    def decode_launch(self):
        self._check(self.lib.dtk_decode_launch(self._ctx), "dtk_decode_launch")

    def decode_wait(self) -> int:
        tok = C.c_int64()
        self._check(self.lib.dtk_decode_wait(self._ctx, C.byref(tok)), "dtk_decode_wait")
        return int(tok.value)

    def get_logits(self) -> torch.Tensor:
        out = np.empty(self.config.vocab, dtype=np.float32)
        self._check(self.lib.dtk_get_logits(self._ctx, out.ctypes.data_as(C.c_void_p)), "dtk_get_logits")
        return torch.from_numpy(out)

    def context_len(self) -> int:
        return int(self.lib.dtk_context_len(self._ctx))

    def set_graph_mode(self, mode: int):
        self._check(self.lib.dtk_set_graph_mode(self._ctx, int(mode)), "dtk_set_graph_mode")

    def set_option(self, name: str, value: int):
        self._check(self.lib.dtk_set_option(self._ctx, name.encode(), int(value)), f"dtk_set_option({name})")

    def synchronize(self):
        self._check(self.lib.dtk_synchronize(self._ctx), "dtk_synchronize")

    def stats(self) -> Dict[str, Any]:
        st = _lib.DtkStats()
        self._check(self.lib.dtk_get_stats(self._ctx, C.byref(st)), "dtk_get_stats")
        return {k: getattr(st, k) for k, _ in st._fields_}

    # ---- generation ---------------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None,
                 bad_words_ids: Optional[List[List[int]]] = None,
                 begin_suppress_tokens: Optional[List[int]] = None,
                 suppress_tokens: Optional[List[int]] = None,
                 streamer=None, stopping_criteria=None, do_sample: Optional[bool] = None,
                 temperature: Optional[float] = None, top_p: Optional[float] = None,
                 top_k: Optional[int] = None, max_length: Optional[int] = None,
                 max_new_tokens: Optional[int] = None, eos_token_id=None, seed: Optional[int] = None,
                 inputs: Optional[torch.Tensor] = None, sequence_owner: Optional[int] = None, **hf_kwargs) -> torch.Tensor:
        """One sequence of HF GenerationMixin.generate/_sample semantics (generation/utils.py
        :2783-2950): streamer.put(prompt) once, then per token: processors -> argmax|draw ->
        append -> streamer.put(token) -> stopping criteria (max length, EOS, user criteria);
        streamer.end().  Returns (1, T') int64 on the host.

        Any other HF generation argument is accepted only at the value that leaves `_sample` unchanged
        (`_NEUTRAL_GENERATE_KWARGS`); everything else — beams, penalties, several return sequences, constraints,
        unknown names — raises instead of being dropped: a drop-in must not silently decode something else."""
        if not self._weights_ready:
            raise _lib.DtkError("no weights loaded (load_state_dict / fill_synthetic first)")
        _reject_unsupported_generate_kwargs(hf_kwargs)
        if input_ids is None:
            input_ids = inputs
        ids = input_ids.detach().to("cpu", torch.int64)
        if ids.dim() == 1:
            ids = ids[None]
        if ids.shape[0] != 1:
            raise ValueError("batch size 1 only (the reference generates one sequence per call)")
        gc = self.generation_config
        do_sample = gc.do_sample if do_sample is None else do_sample
        temperature = gc.temperature if temperature is None else temperature
        top_p = gc.top_p if top_p is None else top_p
        top_k = gc.top_k if top_k is None else top_k
        T = ids.shape[1]
        if max_new_tokens is not None:
            max_length = T + int(max_new_tokens)
        elif max_length is None:
            max_length = gc.max_length
        max_length = min(int(max_length), self.config.max_positions)
        eos = eos_token_id if eos_token_id is not None else gc.eos_token_id
        eos_set = set(_flat_ids([eos]))
        begin_suppress_tokens = _flat_ids(begin_suppress_tokens)
        suppress_tokens = _flat_ids(suppress_tokens)
        bad = []
        for w in (bad_words_ids or []):
            if len(w) != 1:
                raise NotImplementedError("multi-token bad words are not used by DeTikZify")
            bad.append(int(w[0]))
        if seed is None:  # reproducible under torch.manual_seed / transformers.set_seed, like HF
            seed = int(torch.randint(0, 2 ** 62, (), dtype=torch.int64).item()) if do_sample else 0

        if streamer is not None:
            streamer.put(ids.cpu())
        criteria = list(stopping_criteria) if stopping_criteria is not None else []
        n_new_max = max_length - T
        buf = torch.empty((1, max(max_length, T)), dtype=torch.int64)
        buf[0, :T] = ids[0]
        cur = T
        # per-token host work (runs under the GPU's next step): append, stream, stopping criteria.  Kept light: 32 rollouts
        # share one GIL.  The token tensor the HF protocols expect is only built for callers that need it.
        put_token = getattr(streamer, "put_token", None) if streamer is not None else None
        from ..util.generation import ExplicitAbort
        light = [c for c in criteria if type(c) is ExplicitAbort]        # polled flag: ignores its arguments
        heavy = [c for c in criteria if type(c) is not ExplicitAbort]
        new_tokens: List[int] = []

        def emit(tok: int) -> bool:
            nonlocal cur
            new_tokens.append(tok)
            cur += 1
            if put_token is not None:
                put_token(tok)
            elif streamer is not None:
                streamer.put(torch.tensor([tok], dtype=torch.int64))
            stop = tok in eos_set or cur >= max_length
            for c in light:
                stop = stop or c.should_stop
            if heavy:
                buf[0, cur - 1] = tok
                for crit in heavy:
                    r = crit(buf[:, :cur], None)
                    stop = stop or bool(r.all() if isinstance(r, torch.Tensor) else r)
            return stop

        put_tokens = getattr(streamer, "put_tokens", None) if streamer is not None else None

        def emit_many(toks: List[int]) -> bool:
            """the tokens of consecutive steps (a multi-step engine run): the same effects, in order, as emit() per token.  The
            polled abort flag is looked at once per call — a consumer that quits stops the sequence at most one run later."""
            nonlocal cur
            if heavy or (streamer is not None and put_token is None):
                for tok in toks:            # arbitrary criteria / foreign streamers keep the per-token protocol
                    if emit(tok):
                        return True
                return False
            n, stop = 0, False
            for tok in toks:
                n += 1
                if tok in eos_set or cur + n >= max_length:
                    stop = True
                    break
            chunk = toks[:n]
            new_tokens.extend(chunk)
            cur += n
            if put_tokens is not None:
                put_tokens(chunk)
            elif put_token is not None:
                for tok in chunk:
                    put_token(tok)
            for c in light:
                stop = stop or c.should_stop
            return stop

        emit.many, emit.budget, emit.stop_ids = emit_many, (lambda: max_length - cur), eos_set
        emit.aborted = lambda: any(c.should_stop for c in light)
        engine = self.batch_engine
        if n_new_max > 0 and engine is not None:
            # batched mode: this sequence decodes in a KV slot together with the other threads' sequences (infer/engine.py: the
            # native run loop; infer/batching.py: the Python-driven one); one pass over the weights serves all of them.  The
            # sequence's own end (EOS, length budget) goes with it: the native loop stops the slot there.  Tokens come back in
            # bursts (one per source line) unless something here needs to see every token as it is made.
            per_token = bool(heavy) or (streamer is not None and (put_token is None or bool(getattr(streamer, "per_token", put_tokens is None))))
            with engine.sequence(ids[0], pixel_values, dict(
                    do_sample=do_sample, temperature=temperature, top_p=top_p, top_k=top_k, seed=seed, bad_ids=bad,
                    begin_suppress_ids=begin_suppress_tokens or (), always_suppress_ids=suppress_tokens or ()),
                    owner=sequence_owner, max_new_tokens=n_new_max, stop_ids=eos_set, per_token=per_token) as seq:
                seq.run(emit)       # emit.many() per burst in this thread (native engine) / emit() per token by the driving thread
        elif n_new_max > 0:
            # the context has ONE un-slotted sequence: a second generate() on it from another thread would interleave its
            # prefill / decode steps with ours and both would return garbage — refuse loudly (the reference never does
            # this either; concurrent rollouts go through a BatchEngine)
            if not self._single_busy.acquire(blocking=False):
                raise _lib.DtkError("concurrent generate() calls on one model: decode them as a batch "
                                    "(detikzify_amd.infer.batching.BatchEngine / simulate_parallel)")
            try:
                self.set_sampling(do_sample, temperature, top_p, top_k, seed, bad,
                                  begin_suppress_tokens or (), suppress_tokens or ())
                self.prefill(ids[0], pixel_values)
                launched = received = 0
                stop = False
                ahead = 2  # one step always in flight while the host handles the previous token
                while launched < min(ahead, n_new_max):
                    self.decode_launch(); launched += 1
                while received < launched:
                    tok = self.decode_wait(); received += 1
                    stop = emit(tok)
                    if stop:
                        break
                    if launched < n_new_max:
                        self.decode_launch(); launched += 1
            finally:
                self._single_busy.release()
        if streamer is not None:
            streamer.end()
        if new_tokens:
            buf[0, T:T + len(new_tokens)] = torch.tensor(new_tokens, dtype=torch.int64)
        return buf[:, :cur].clone()
