"""ctypes binding of libowc_hip.so (C ABI: include/owc.h).

The product path has no CPU fallback: if the shared library is missing or a call fails the
error is raised, never swallowed.  ``oracle/`` is test infrastructure and is never imported here.
"""

from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

# torch ships its own libamdhip64: it must be the HIP runtime of the process, so import torch BEFORE
# dlopen-ing libowc_hip.so (whose DT_NEEDED libamdhip64.so.7 then binds to the already loaded runtime).
# Loading the extension first would put a second HIP runtime in the process (hipSetDevice fails).
import torch  # noqa: F401

_LIB_PATH = Path(__file__).resolve().parent / "libowc_hip.so"
_lock = threading.Lock()
_lib: C.CDLL | None = None
_ctx: dict[int, C.c_void_p] = {}
_timing_ok = False

ABI_VERSION = 19

EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_RESIDUAL, EPI_SWIGLU, EPI_F32 = range(6)

vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t


class OwcError(RuntimeError):
    """A libowc_hip.so call returned a negative status (or the extension is missing)."""


class VitLayer(C.Structure):
    _fields_ = [(n, vp) for n in (
        "ln1_w", "ln1_b", "qkv_w", "qkv_b", "proj_w", "proj_b",
        "ln2_w", "ln2_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b")]


class VitWeights(C.Structure):
    _fields_ = [
        ("depth", C.c_int32), ("embed_dim", C.c_int32), ("num_heads", C.c_int32),
        ("mlp_hidden", C.c_int32), ("patch_k", C.c_int32), ("out_dim", C.c_int32),
        ("merge_unit", C.c_int32), ("ln_eps", f32),
        ("patch_w", vp), ("layers", C.POINTER(VitLayer)),
        ("merger_ln_w", vp), ("merger_ln_b", vp), ("merger_fc1_w", vp), ("merger_fc1_b", vp),
        ("merger_fc2_w", vp), ("merger_fc2_b", vp),
        ("rope_cos", vp), ("rope_sin", vp), ("rope_positions", C.c_int32),
        ("variant", C.c_int32), ("fullatt_mask", C.c_uint64),
    ]


class ClipWeights(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("embed_dim", C.c_int32), ("num_heads", C.c_int32), ("mlp_hidden", C.c_int32),
        ("patch_k", C.c_int32), ("tokens", C.c_int32), ("out_dim", C.c_int32), ("ln_eps", f32),
        ("patch_w", vp), ("pos_cls", vp), ("pre_ln_w", vp), ("pre_ln_b", vp), ("layers", C.POINTER(VitLayer)),
        ("proj1_w", vp), ("proj1_b", vp), ("proj2_w", vp), ("proj2_b", vp),
    ]


class LlmLayer(C.Structure):
    _fields_ = [(n, vp) for n in ("ln1_w", "qkv_w", "qkv_b", "o_w", "ln2_w", "gateup_w", "down_w",
                                  "qkv_s", "o_s", "gateup_s", "down_s")]


class LlmWeights(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("d_model", C.c_int32), ("n_q_heads", C.c_int32),
        ("n_kv_heads", C.c_int32), ("head_dim", C.c_int32), ("d_ff", C.c_int32), ("vocab", C.c_int32),
        ("mrope_sec0", C.c_int32), ("mrope_sec1", C.c_int32), ("rms_eps", f32),
        ("embed", vp), ("layers", C.POINTER(LlmLayer)), ("final_norm_w", vp), ("lm_head_w", vp),
        ("rope_cos", vp), ("rope_sin", vp), ("rope_positions", C.c_int32), ("weight_dtype", C.c_int32),
    ]


WEIGHTS_BF16, WEIGHTS_FP8 = 0, 1
PREFILL_LAST_TOKENS, PREFILL_SCORE_ROWS = 0, 1   # owc_llm_prefill score_mode


class KvCache(C.Structure):
    _fields_ = [("k", vp), ("v", vp), ("n_slots", C.c_int32), ("s_max", C.c_int32)]


class Sampling(C.Structure):
    """owc_sampling: temperature / top-k / top-p + the Philox key and the per-sequence stream ids (include/owc.h)."""
    _fields_ = [("temperature", f32), ("top_k", C.c_int32), ("top_p", f32), ("seed", C.c_uint64), ("stream_id", vp), ("step_offset", vp)]


class BertLayer(C.Structure):
    _fields_ = [(n, vp) for n in (
        "qkv_w", "qkv_b", "o_w", "o_b", "ln1_w", "ln1_b", "fc1_w", "fc1_b", "fc2_w", "fc2_b",
        "ln2_w", "ln2_b")]


class BertWeights(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("hidden", C.c_int32), ("n_heads", C.c_int32), ("inter", C.c_int32),
        ("vocab", C.c_int32), ("max_pos", C.c_int32), ("ln_eps", f32),
        ("word_emb", vp), ("pos_emb", vp), ("type_emb", vp), ("emb_ln_w", vp), ("emb_ln_b", vp),
        ("layers", C.POINTER(BertLayer)),
        ("rel_bias", vp), ("rel_span", C.c_int32), ("pos_offset", C.c_int32),    # MPNet (include/owc.h); NULL / 0 / 0 for BERT
    ]


# name -> (restype, argtypes); every symbol include/owc.h declares
SIGNATURES: dict[str, tuple] = {
    "owc_abi_version": (i32, []),
    "owc_has_timing_knobs": (i32, []),
    "owc_tuning_set": (i32, [C.c_char_p, i32]),
    "owc_init": (i32, [i32, C.POINTER(vp)]),
    "owc_destroy": (i32, [vp]),
    "owc_last_error": (C.c_char_p, [vp]),
    "owc_gemm_bf16": (i32, [vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "owc_gemm_f32": (i32, [vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "owc_layernorm_bf16": (i32, [vp, vp, i64, vp, vp, vp, i64, i32, i32, f32, vp]),
    "owc_rmsnorm_bf16": (i32, [vp, vp, i64, vp, vp, i64, i32, i32, f32, vp, vp]),
    "owc_rope_table": (i32, [vp, vp, vp, i32, i32, i32, f32, i32, vp]),
    "owc_vision_rope": (i32, [vp, vp, i64, vp, vp, vp, i32, i32, i32, vp]),
    "owc_mrope_kv_write": (i32, [vp, vp, i64, vp, i64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "owc_decode_attention": (i32, [vp, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, i32, i32, f32, vp]),
    "owc_attention_bf16": (i32, [vp, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, i64, i64, vp, vp, vp, vp, vp,
                                 i32, i32, i32, i32, i32, i32, f32, vp]),
    "owc_quantize_rows_fp8": (i32, [vp, vp, i64, vp, i64, vp, i32, i32, vp]),
    "owc_rmsnorm_quant_fp8": (i32, [vp, vp, i64, vp, vp, i64, vp, i32, i32, f32, vp]),
    "owc_gemm_fp8": (i32, [vp, vp, i64, vp, vp, i64, vp, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]),
    "owc_embed_tokens": (i32, [vp, vp, vp, vp, vp, vp, i32, i32, vp]),
    "owc_argmax_bf16": (i32, [vp, vp, i64, i32, i32, vp, vp]),
    "owc_sample_bf16": (i32, [vp, vp, i64, i32, i32, C.POINTER(Sampling), vp, i32, vp, vp]),
    "owc_beam_candidates": (i32, [vp, vp, i64, i32, i32, i32, vp, vp, vp, vp]),
    "owc_token_logprob_bf16": (i32, [vp, vp, i64, vp, i32, i32, vp, vp]),
    "owc_patchify_u8": (i32, [vp, vp, vp, i64, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp]),
    "owc_vit_workspace_bytes": (sz, [C.POINTER(VitWeights), i32]),
    "owc_vit_forward": (i32, [vp, C.POINTER(VitWeights), vp, i64, vp, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "owc_vit25_forward": (i32, [vp, C.POINTER(VitWeights), vp, i64, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, sz, vp]),
    "owc_clip_workspace_bytes": (sz, [C.POINTER(ClipWeights), i32]),
    "owc_clip_forward": (i32, [vp, C.POINTER(ClipWeights), vp, i64, i32, vp, vp, sz, vp]),
    "owc_clip_patchify_u8": (i32, [vp, vp, vp, i64, i32, i32, i32, C.POINTER(f32), C.POINTER(f32), vp]),
    "owc_llm_workspace_bytes": (sz, [C.POINTER(LlmWeights), i32, i32]),
    "owc_llm_prefill": (i32, [vp, C.POINTER(LlmWeights), C.POINTER(KvCache), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp,
                              i32, i32, i32, i32, i32, i32, i32, C.POINTER(Sampling), i32, vp, vp, vp, sz, vp]),
    "owc_llm_decode_step": (i32, [vp, C.POINTER(LlmWeights), C.POINTER(KvCache), vp, vp, vp, vp, vp, vp, vp, vp, vp,
                                  vp, vp, i32, i32, vp, i32, i32, i32, i32, vp, vp, C.POINTER(Sampling), vp, vp, sz, vp]),
    "owc_decode_update": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp]),
    "owc_llm_set_repetition_penalty": (i32, [vp, f32, vp, i32]),
    "owc_seen_mark": (i32, [vp, vp, vp, i32, i32, vp, i32, vp]),
    "owc_argmax_penalized_bf16": (i32, [vp, vp, i64, i32, i32, vp, i32, vp, f32, vp, vp]),
    "owc_decode_compact": (i32, [vp, vp, i32] + [vp] * 17),
    "owc_bert_workspace_bytes": (sz, [C.POINTER(BertWeights), i32, i32]),
    "owc_bert_embed": (i32, [vp, C.POINTER(BertWeights), vp, vp, i32, i32, vp, vp, sz, vp]),
    "owc_bert_packed_workspace_bytes": (sz, [C.POINTER(BertWeights), i32]),
    "owc_bert_embed_packed": (i32, [vp, C.POINTER(BertWeights), vp, vp, vp, i32, i32, i32, vp, vp, sz, vp]),
    "owc_cosine_topk": (i32, [vp, vp, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp]),
    "owc_paired_dot": (i32, [vp, vp, vp, i32, i32, vp, vp]),
    "owc_gemm_profile_enable": (i32, [vp, i32]),
    "owc_gemm_profile_read": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "owc_profile_read": (i32, [vp, i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "owc_profile_shapes": (i32, [vp, i32, C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
}

PROF_KINDS = ("gemm_bf16", "gemm_fp8", "attn_vision", "attn_prefill", "scorer_gemm", "cosine_topk", "attn_decode")   # enum owc_prof_kind


def lib_path() -> Path:
    return _LIB_PATH


def use_timing_library() -> None:
    """tools/ only: load libowc_hip_timing.so (the -DOWC_TIMING_KNOBS build, `python -m lmms_owc_amd.build --timing`) instead of
    the product library.  Must be called before the first `load()`; nothing in the package calls it."""
    global _LIB_PATH, _timing_ok
    if _lib is not None:
        raise OwcError("use_timing_library() must come before the first load()")
    _LIB_PATH = _LIB_PATH.with_name("libowc_hip_timing.so")
    _timing_ok = True


def load() -> C.CDLL:
    """Load libowc_hip.so; raises (loudly) if it has not been built or lacks a symbol."""
    global _lib
    with _lock:
        if _lib is None:
            if not _LIB_PATH.exists():
                raise OwcError(
                    f"{_LIB_PATH} is missing: run `python -m lmms_owc_amd.build` "
                    "(the HIP extension is mandatory, there is no fallback path)"
                )
            lib = C.CDLL(str(_LIB_PATH))
            for name, (res, args) in SIGNATURES.items():
                fn = getattr(lib, name)  # AttributeError if the symbol is not exported
                fn.restype = res
                fn.argtypes = args
            if lib.owc_abi_version() != ABI_VERSION:
                raise OwcError("libowc_hip.so ABI version mismatch: rebuild the extension")
            if lib.owc_has_timing_knobs() and not _timing_ok:
                raise OwcError(f"{_LIB_PATH} was built with -DOWC_TIMING_KNOBS: that build is for tools/ only, rebuild the product library")
            _lib = lib
        return _lib


def ctx(device: int = 0) -> C.c_void_p:
    """The process's library context (needs a GPU).  One process per GPU: a second device in the same process is refused
    (kernel attributes, tuning knobs and the profile recording are process-wide in the library)."""
    lib = load()
    with _lock:
        if _ctx and device not in _ctx:
            raise OwcError(f"this process already drives cuda:{next(iter(_ctx))}; libowc_hip is one context per process "
                           f"(one process per GPU) - cuda:{device} needs its own process")
        if device not in _ctx:
            h = C.c_void_p()
            rc = lib.owc_init(device, C.byref(h))
            if rc != 0:
                raise OwcError(f"owc_init(device={device}) failed with status {rc}")
            _ctx[device] = h
        return _ctx[device]


def check(rc: int, device: int = 0) -> None:
    if rc != 0:
        msg = load().owc_last_error(_ctx.get(device))
        raise OwcError(f"libowc_hip status {rc}: {msg.decode() if msg else '?'}")


_h2d_pending: list = []   # (event, host tensor) of copies the stream may not have executed yet


def h2d(a, device, dtype=None) -> "torch.Tensor":
    """Host array (numpy, or a CPU tensor) -> device tensor on the current stream, without a host synchronisation and without
    depending on what the runtime does with pageable memory.

    `torch.from_numpy(tmp).to(device, non_blocking=True)` on a temporary is only safe if the runtime has read the source before the
    call returns.  ROCm 7.2 does (the call stages the bytes: its host time grows with the size like a memcpy, tools/bench_h2d.py), but
    nothing here should hinge on that: the source tensor - and through it the numpy buffer - is kept referenced until an event
    recorded behind the copy has passed, so its bytes can be neither freed nor recycled earlier.  (Going through torch's pinned-host
    cache instead costs 0.3-0.4 ms per MB on this platform - writes into pinned memory are slow - and took the label-cosine leg from
    670 k to 215 k labels/s.)"""
    import numpy as np

    t = a if isinstance(a, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(a, dtype=dtype))
    if t.numel() == 0:
        return torch.empty(t.shape, dtype=t.dtype, device=device)
    t = t.contiguous()
    out = t.to(device, non_blocking=True)
    with _lock:
        while _h2d_pending and _h2d_pending[0][0].query():
            _h2d_pending.pop(0)
        ev = torch.cuda.Event()
        ev.record()
        _h2d_pending.append((ev, t))
    return out


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()
