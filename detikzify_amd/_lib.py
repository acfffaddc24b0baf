"""ctypes binding of include/dtk.h.  No CPU fallback: a missing library is a hard error."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "lib" / "libdtk_hip.so"

DTK_ABI_VERSION = 6          # include/dtk.h DTK_ABI_VERSION
DTK_VIT_BATCH = 8            # include/dtk.h: images per pass of dtk_vit_encode
DTK_F32, DTK_BF16, DTK_F16 = 0, 1, 2
DTK_ARCH_PROJ_NO_BIAS = 1   # include/dtk.h: dtk_config.reserved[3] flag
DTK_PREFILL_REUSE_PREFIX, DTK_PREFILL_REUSE_IMAGE = 1, 2
DTK_MAX_INFLIGHT = 4
DTK_MAX_BATCH = 64          # entries of the active / tokens_out arrays of dtk_decode_batch_*
DTK_EPI_BIAS, DTK_EPI_GELU, DTK_EPI_RESIDUAL, DTK_GEMM_NAIVE = 1, 2, 4, 256
DTK_GEMM_WT = 512
DTK_GEMM_SL = 1024
DTK_GEMM_KSLICES_SHIFT = 12     # dtk_op_gemm: flags |= S << 12 selects the sliced-K family (include/dtk.h)


class DtkConfig(C.Structure):
    _fields_ = [
        ("hidden", C.c_int32), ("layers", C.c_int32), ("heads", C.c_int32), ("head_dim", C.c_int32),
        ("ffn", C.c_int32), ("vocab", C.c_int32), ("max_positions", C.c_int32),
        ("rms_eps", C.c_float), ("rope_theta", C.c_float), ("rope_factor", C.c_float),
        ("vit_dim", C.c_int32), ("vit_depth", C.c_int32), ("vit_heads", C.c_int32),
        ("vit_mlp", C.c_int32), ("vit_patch", C.c_int32), ("vit_image", C.c_int32),
        ("vit_feature_layer", C.c_int32), ("vit_ln_eps", C.c_float), ("vit_gelu_tanh", C.c_int32),
        ("concat_patches", C.c_int32), ("image_token_id", C.c_int32), ("attn_splits", C.c_int32),
        ("reserved", C.c_int32 * 7),
    ]


class DtkSampling(C.Structure):
    _fields_ = [
        ("do_sample", C.c_int32), ("temperature", C.c_float), ("top_p", C.c_float),
        ("top_k", C.c_int32), ("seed", C.c_uint64),
        ("n_bad", C.c_int32), ("bad_ids", C.c_int32 * 8),
        ("n_begin_suppress", C.c_int32), ("begin_suppress_ids", C.c_int32 * 8),
        ("n_always_suppress", C.c_int32), ("always_suppress_ids", C.c_int32 * 8),
    ]


class DtkStats(C.Structure):
    _fields_ = [
        ("weight_bytes_per_token", C.c_uint64), ("kv_bytes_per_ctx_token", C.c_uint64),
        ("decode_steps", C.c_uint64), ("prefill_tokens", C.c_uint64), ("vit_images", C.c_uint64),
        ("last_prefill_ms", C.c_double), ("last_vit_ms", C.c_double),
        ("probe_kernel_ms_sum", C.c_double), ("probe_kernel_launches", C.c_uint64),
        ("probe_kernel_bytes", C.c_uint64), ("probe_event_pair_ms", C.c_double),
        ("last_batch_step_slots", C.c_uint32), ("device_errors", C.c_uint32),
        ("last_batch_step_fp8_mfma", C.c_uint32), ("reserved0", C.c_uint32),
    ]


class DtkJoin(C.Structure):
    """dtk_join: one sequence handed to the native run loop (dtk_engine_join)"""
    _fields_ = [
        ("slot", C.c_int32), ("n_ids", C.c_int32), ("ids", C.c_void_p), ("pixels", C.c_void_p), ("image_key", C.c_uint64),
        ("try_resume", C.c_int32), ("n_candidates", C.c_int32), ("candidates", C.c_int32 * DTK_MAX_BATCH),
        ("prefix_len", C.c_int32), ("prefix_src", C.c_int32), ("prefix_src_whole", C.c_int32), ("prefix_encode", C.c_int32),
        ("prefix_in_place", C.c_int32), ("full_flags", C.c_int32), ("sampling", DtkSampling),
        ("max_new_tokens", C.c_int32), ("n_stop", C.c_int32), ("stop_ids", C.c_int64 * 8),
        ("flush_mode", C.c_int32), ("flush_max", C.c_int32), ("slot_out", C.c_int32), ("how_out", C.c_int32),
        ("error_out", C.c_char * 240),
    ]


class DtkEngineStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("steps", "tokens_out", "joins", "resumed", "steps_below_half_occupancy", "host_bound_steps",
                                          "reader_wakeups", "wasted_slot_steps")] + \
               [(n, C.c_double) for n in ("wait_s", "launch_s", "join_s", "idle_s", "drain_s", "first_launch_t", "last_collect_t")]


_I32P, _I64P = C.POINTER(C.c_int32), C.POINTER(C.c_int64)


class DtkEngineOps(C.Structure):
    """dtk_engine_ops: the device under the native run loop (the CPU tests script one in Python)"""
    LAUNCH = C.CFUNCTYPE(C.c_int, C.c_void_p, _I32P)
    WAIT = C.CFUNCTYPE(C.c_int, C.c_void_p, _I64P)
    PREFILL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _I64P, C.c_int, C.c_void_p, C.c_uint64, C.c_int)
    SAMPLING = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(DtkSampling))
    FORK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int)
    LCP = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _I64P, C.c_int, C.c_uint64, C.POINTER(C.c_int))
    RESUME = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, _I64P, C.c_int, C.c_uint64)
    CTXLEN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)
    LASTERR = C.CFUNCTYPE(C.c_void_p, C.c_void_p)     # (const char*: the callee owns the text)
    _fields_ = [
        ("dev", C.c_void_p), ("launch", LAUNCH), ("wait", WAIT), ("prefill_slot", PREFILL), ("set_sampling_slot", SAMPLING),
        ("kv_fork", FORK), ("slot_lcp", LCP), ("resume_slot", RESUME), ("context_len_slot", CTXLEN), ("last_error", LASTERR),
        ("max_positions", C.c_int32), ("decode_slots", C.c_int32),
    ]


DTK_JOIN_FULL, DTK_JOIN_FORK_TAIL, DTK_JOIN_FORK_WHOLE, DTK_JOIN_RESUMED, DTK_JOIN_IN_PLACE = range(5)
DTK_SEQ_RUNNING, DTK_SEQ_FINISHED, DTK_SEQ_LEFT = 1, 2, 3

# every symbol include/dtk.h declares: (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "dtk_abi_version": (C.c_int, []),
    "dtk_abi_struct_size": (C.c_int, [C.c_int]),
    "dtk_last_error": (C.c_char_p, [_P]),
    "dtk_create": (C.c_int, [C.POINTER(DtkConfig), C.c_int, C.POINTER(_P)]),
    "dtk_destroy": (None, [_P]),
    "dtk_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64), C.c_int]),
    "dtk_read_tensor": (C.c_int, [_P, C.c_char_p, _P, C.c_int64]),
    "dtk_fill_synthetic": (C.c_int, [_P, C.c_uint64]),
    "dtk_num_tensors": (C.c_int, [_P]),
    "dtk_tensor_name": (C.c_char_p, [_P, C.c_int]),
    "dtk_tensor_numel": (C.c_int64, [_P, C.c_char_p]),
    "dtk_vit_encode": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "dtk_prefill": (C.c_int, [_P, _P, C.c_int, _P, C.c_uint64, C.c_int, _P]),
    "dtk_set_sampling": (C.c_int, [_P, C.POINTER(DtkSampling)]),
    "dtk_decode_launch": (C.c_int, [_P]),
    "dtk_decode_wait": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "dtk_decode": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "dtk_get_logits": (C.c_int, [_P, _P]),
    "dtk_context_len": (C.c_int, [_P]),
    "dtk_set_graph_mode": (C.c_int, [_P, C.c_int]),
    "dtk_synchronize": (C.c_int, [_P]),
    "dtk_get_stats": (C.c_int, [_P, C.POINTER(DtkStats)]),
    "dtk_num_slots": (C.c_int, [_P]),
    "dtk_max_decode_slots": (C.c_int, [_P]),
    "dtk_max_positions": (C.c_int, [_P]),
    "dtk_engine_create": (C.c_int, [_P, C.POINTER(_P)]),
    "dtk_engine_create_ops": (C.c_int, [C.POINTER(DtkEngineOps), C.POINTER(_P)]),
    "dtk_engine_destroy": (None, [_P]),
    "dtk_engine_last_error": (C.c_char_p, [_P]),
    "dtk_engine_set_flush_tokens": (C.c_int, [_P, C.POINTER(C.c_int64), C.c_int]),
    "dtk_engine_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "dtk_engine_expect": (C.c_int, [_P, C.c_int, C.c_int]),
    "dtk_engine_join": (C.c_int, [_P, C.POINTER(DtkJoin)]),
    "dtk_engine_submit": (C.c_int, [_P, C.POINTER(DtkJoin), C.POINTER(C.c_uint64)]),
    "dtk_engine_await": (C.c_int, [_P, C.c_uint64]),
    "dtk_engine_read": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int]),
    "dtk_engine_leave": (C.c_int, [_P, C.c_int]),
    "dtk_engine_get_stats": (C.c_int, [_P, C.POINTER(DtkEngineStats)]),
    "dtk_prefill_slot": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_uint64, C.c_int, _P]),
    "dtk_set_sampling_slot": (C.c_int, [_P, C.c_int, C.POINTER(DtkSampling)]),
    "dtk_decode_batch_launch": (C.c_int, [_P, C.POINTER(C.c_int32)]),
    "dtk_decode_batch_wait": (C.c_int, [_P, C.POINTER(C.c_int64)]),
    "dtk_kv_fork": (C.c_int, [_P, C.c_int, C.c_int, C.c_int]),
    "dtk_get_logits_slot": (C.c_int, [_P, C.c_int, _P]),
    "dtk_context_len_slot": (C.c_int, [_P, C.c_int]),
    "dtk_slot_lcp": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_uint64, C.POINTER(C.c_int)]),
    "dtk_slot_cached_ids": (C.c_int, [_P, C.c_int, _P, C.c_int]),
    "dtk_resume_slot": (C.c_int, [_P, C.c_int, _P, C.c_int, C.c_uint64]),
    "dtk_bench_gemv": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "dtk_set_gemv_variant": (C.c_int, [_P, C.c_int, C.c_int]),
    "dtk_set_option": (C.c_int, [_P, C.c_char_p, C.c_int]),
    "dtk_op_gemm": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dtk_op_gemv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, _P]),
    "dtk_op_gemv_mv": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    "dtk_mx_layout": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, _P]),
    "dtk_op_gemv_mx": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P, _P]),
    "dtk_op_attention": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "dtk_op_layernorm": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_float, _P]),
    "dtk_op_sample": (C.c_int, [_P, _P, C.c_int, C.c_int, C.POINTER(C.c_int64), _P]),
}

_lib = None


class DtkError(RuntimeError):
    pass


def load_library() -> C.CDLL:
    """Load libdtk_hip.so once.  If torch is importable it is imported FIRST so that exactly one
    HIP runtime (torch's bundled libamdhip64.so.7) is in the process — the library's NEEDED
    libamdhip64.so.7 then resolves to the already-loaded copy and torch.distributed (RCCL) can
    share the process.  DTK_HIP_RUNTIME=system skips that and uses /opt/rocm via RUNPATH."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise DtkError(
            f"{LIB_PATH} is missing: build it with ./build.sh (hipcc --offload-arch=gfx950). "
            "detikzify_amd has no CPU fallback.")
    if os.environ.get("DTK_HIP_RUNTIME", "torch") != "system":
        try:
            import torch  # noqa: F401  (side effect: loads torch/lib/libamdhip64.so)
        except Exception:  # pragma: no cover - torch-less deployments use the system runtime
            pass
    lib = C.CDLL(str(LIB_PATH), mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError here = ABI mismatch, loud by design
        fn.restype, fn.argtypes = res, args
    if lib.dtk_abi_version() != DTK_ABI_VERSION:
        raise DtkError(f"ABI version {lib.dtk_abi_version()} != {DTK_ABI_VERSION}")
    for which, (struct, field) in enumerate(((DtkConfig, None), (DtkSampling, None), (DtkStats, None), (DtkSampling, "seed"),
                                             (DtkConfig, "reserved"), (DtkStats, "probe_event_pair_ms"))):
        ours = C.sizeof(struct) if field is None else getattr(struct, field).offset
        if lib.dtk_abi_struct_size(which) != ours:     # a drifted struct would corrupt every call: refuse to run
            raise DtkError(f"struct layout mismatch with include/dtk.h: {struct.__name__}{'.' + field if field else ''} "
                           f"is {ours} here, {lib.dtk_abi_struct_size(which)} in the library")
    _lib = lib
    return lib


def check(lib, ctx, rc: int, what: str):
    if rc != 0:
        msg = lib.dtk_last_error(ctx)
        text = msg.decode(errors="replace") if msg else ""
        if rc == -1 and ("image patch tokens" in text):
            raise ValueError(text)          # same exception type/message as the reference
        raise DtkError(f"{what} failed ({rc}): {text}")
