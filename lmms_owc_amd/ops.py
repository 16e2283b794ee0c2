"""Torch-tensor front end of the op-level C ABI (include/owc.h).

Tensors are only device-memory handles here: every function passes raw pointers and sizes to
libowc_hip.so on torch's current HIP stream.  No torch arithmetic happens in this module.
"""

from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import EPI_F32, EPI_NONE, EPI_SWIGLU, ptr

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise _lib.OwcError("libowc_hip needs device tensors (no CPU fallback on the product path)")
    return t.device.index or 0


def _call(name: str, dev: int, *args) -> None:
    rc = getattr(_lib.load(), name)(_lib.ctx(dev), *args, _lib.stream_ptr())
    _lib.check(rc, dev)


def gemm_bf16(a, w, bias=None, *, epilogue=EPI_NONE, residual=None, out=None):
    """``out[M,N] = a[M,K] @ w[N,K].T (+bias)`` with a fused epilogue (nn.Linear semantics)."""
    assert a.dtype == BF16 and w.dtype == BF16 and a.dim() == 2 and w.dim() == 2
    assert a.shape[1] == w.shape[1] and a.stride(1) == 1 and w.stride(1) == 1
    m, k = a.shape
    n = w.shape[0]
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    if out is None:
        out = torch.empty((m, n_out), dtype=F32 if epilogue == EPI_F32 else BF16, device=a.device)
    assert out.stride(1) == 1 and out.shape == (m, n_out)
    _call("owc_gemm_bf16", _dev(a), a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), ptr(bias),
          ptr(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(),
          out.stride(0), m, n, k, epilogue)
    return out


def quantize_rows_fp8(x, out=None, scale=None):
    """bf16 [rows, cols] -> (uint8 e4m3fn codes [rows, cols], float32 scales [rows]): q = rne(x / s), s = max|x| / 448."""
    assert x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    rows, cols = x.shape
    if out is None:
        out = torch.empty((rows, cols), dtype=torch.uint8, device=x.device)
    if scale is None:
        scale = torch.empty(rows, dtype=F32, device=x.device)
    _call("owc_quantize_rows_fp8", _dev(x), x.data_ptr(), x.stride(0), out.data_ptr(), out.stride(0), scale.data_ptr(), rows, cols)
    return out, scale


def rmsnorm_quant_fp8(x, w, eps):
    """rmsnorm(x) quantised per token to e4m3fn without materialising the bf16 rows: (codes uint8 [rows, d], scales f32 [rows])."""
    assert x.dtype == BF16 and w.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    rows, d = x.shape
    q = torch.empty((rows, d), dtype=torch.uint8, device=x.device)
    s = torch.empty(rows, dtype=F32, device=x.device)
    _call("owc_rmsnorm_quant_fp8", _dev(x), x.data_ptr(), x.stride(0), w.data_ptr(), q.data_ptr(), q.stride(0), s.data_ptr(), rows, d, float(eps))
    return q, s


def gemm_fp8(a8, a_scale, w8, w_scale, bias=None, *, epilogue=EPI_NONE, residual=None, out=None):
    """``out[M,N] bf16 = epilogue((a8 @ w8.T) * a_scale[:, None] * w_scale[None, :] + bias)`` on the scaled fp8 MFMA."""
    assert a8.dtype == torch.uint8 and w8.dtype == torch.uint8 and a8.stride(1) == 1 and w8.stride(1) == 1
    assert a_scale.dtype == F32 and w_scale.dtype == F32 and a8.shape[1] == w8.shape[1]
    m, k = a8.shape
    n = w8.shape[0]
    n_out = n // 2 if epilogue == EPI_SWIGLU else n
    if out is None:
        out = torch.empty((m, n_out), dtype=BF16, device=a8.device)
    _call("owc_gemm_fp8", _dev(a8), a8.data_ptr(), a8.stride(0), a_scale.data_ptr(), w8.data_ptr(), w8.stride(0), w_scale.data_ptr(),
          ptr(bias), ptr(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(), out.stride(0), m, n, k,
          epilogue)
    return out


def gemm_f32(a, w, bias=None, *, epilogue=EPI_NONE, residual=None, out=None):
    assert a.dtype == F32 and w.dtype == F32 and a.stride(1) == 1 and w.stride(1) == 1
    m, k = a.shape
    n = w.shape[0]
    if out is None:
        out = torch.empty((m, n), dtype=F32, device=a.device)
    _call("owc_gemm_f32", _dev(a), a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), ptr(bias),
          ptr(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(),
          out.stride(0), m, n, k, epilogue)
    return out


def layernorm(x, w, b, eps, out=None):
    assert x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    out = torch.empty_like(x) if out is None else out
    _call("owc_layernorm_bf16", _dev(x), x.data_ptr(), x.stride(0), w.data_ptr(), b.data_ptr(),
          out.data_ptr(), out.stride(0), x.shape[0], x.shape[1], float(eps))
    return out


def rmsnorm(x, w, eps, row_index=None, out=None):
    assert x.dtype == BF16 and x.dim() == 2 and x.stride(1) == 1
    rows = x.shape[0] if row_index is None else row_index.numel()
    if out is None:
        out = torch.empty((rows, x.shape[1]), dtype=BF16, device=x.device)
    _call("owc_rmsnorm_bf16", _dev(x), x.data_ptr(), x.stride(0), w.data_ptr(), out.data_ptr(),
          out.stride(0), rows, x.shape[1], float(eps), ptr(row_index))
    return out


def rope_table(n_pos, n_freq, dim, theta, round_bf16, device):
    cos = torch.empty((n_pos, n_freq), dtype=F32, device=device)
    sin = torch.empty_like(cos)
    _call("owc_rope_table", _dev(cos), cos.data_ptr(), sin.data_ptr(), n_pos, n_freq, dim, float(theta),
          int(round_bf16))
    return cos, sin


def vision_rope_(qkv, pos_hw, cos, sin, n_heads, head_dim):
    assert qkv.dtype == BF16 and pos_hw.dtype == I32
    _call("owc_vision_rope", _dev(qkv), qkv.data_ptr(), qkv.stride(0), pos_hw.data_ptr(), cos.data_ptr(),
          sin.data_ptr(), qkv.shape[0], n_heads, head_dim)
    return qkv


def mrope_kv_write_(qkv, pos3, cos, sin, k_cache, v_cache, tok_slot, tok_idx, n_q, n_kv, s_max, sec0, sec1,
                    pos_stride=None):
    t = qkv.shape[0]
    _call("owc_mrope_kv_write", _dev(qkv), qkv.data_ptr(), qkv.stride(0), pos3.data_ptr(),
          t if pos_stride is None else pos_stride, cos.data_ptr(), sin.data_ptr(), k_cache.data_ptr(),
          v_cache.data_ptr(), tok_slot.data_ptr(), tok_idx.data_ptr(), t, n_q, n_kv, s_max, sec0, sec1)
    return qkv


def attention(q, q_ts, q_hs, k, k_ts, k_hs, v, v_ts, v_hs, out, o_ts, o_hs, q_start, k_start, seq_len, *,
              n_seq, n_heads, kv_group, head_dim, max_q_len, causal, scale, o_start=None, q_len=None):
    _call("owc_attention_bf16", _dev(out), q.data_ptr(), q_ts, q_hs, k.data_ptr(), k_ts, k_hs, v.data_ptr(),
          v_ts, v_hs, out.data_ptr(), o_ts, o_hs, q_start.data_ptr(), ptr(o_start), k_start.data_ptr(),
          seq_len.data_ptr(), ptr(q_len), n_seq, n_heads, kv_group, head_dim, max_q_len, int(causal),
          float(scale))
    return out


def decode_attention(qkv, pos, cos, sin, k_cache, v_cache, slot, write_idx, k_len, n_q, n_kv, s_max, scale, out=None):
    """One decode step's rope + KV-cache write + attention for B sequences (owc_decode_attention); returns [B, n_q * 128]."""
    b = qkv.shape[0]
    if out is None:
        out = torch.empty((b, n_q * 128), dtype=BF16, device=qkv.device)
    _call("owc_decode_attention", _dev(qkv), qkv.data_ptr(), qkv.stride(0), pos.data_ptr(), cos.data_ptr(), sin.data_ptr(),
          k_cache.data_ptr(), v_cache.data_ptr(), slot.data_ptr(), write_idx.data_ptr(), k_len.data_ptr(), out.data_ptr(),
          out.stride(0), b, n_q, n_kv, s_max, float(scale))
    return out


def embed_tokens(ids, img_index, table, img_embeds):
    out = torch.empty((ids.numel(), table.shape[1]), dtype=BF16, device=table.device)
    _call("owc_embed_tokens", _dev(table), ids.data_ptr(), ptr(img_index), table.data_ptr(), ptr(img_embeds),
          out.data_ptr(), ids.numel(), table.shape[1])
    return out


def argmax_bf16(logits):
    out = torch.empty((logits.shape[0],), dtype=I32, device=logits.device)
    _call("owc_argmax_bf16", _dev(logits), logits.data_ptr(), logits.stride(0), logits.shape[0],
          logits.shape[1], out.data_ptr())
    return out


def seen_mark_(seen, ids, slot=None):
    """Mark token ids[t] in row slot[t] (None: row t) of the uint32 bitmap `seen` [rows, words] (int32 tensor viewed as bits): `owc_seen_mark`."""
    _call("owc_seen_mark", _dev(seen), ids.data_ptr(), _lib.ptr(slot), ids.numel(), seen.shape[1] * 32, seen.data_ptr(), seen.shape[1])
    return seen


def argmax_penalized_bf16(logits, seen, penalty: float, row_slot=None):
    """Greedy argmax of bf16 logits [rows, vocab] under HF's repetition penalty: ids marked in bitmap row row_slot[r] (None: r) of
    `seen` [slots, words >= vocab / 32] get score < 0 ? score * p : score / p in fp32 first: `owc_argmax_penalized_bf16`."""
    out = torch.empty((logits.shape[0],), dtype=I32, device=logits.device)
    _call("owc_argmax_penalized_bf16", _dev(logits), logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1],
          seen.data_ptr(), seen.shape[1], _lib.ptr(row_slot), float(penalty), out.data_ptr())
    return out


def beam_candidates(logits, k: int):
    """bf16 logits [rows, vocab] -> (logz float32 [rows], top_val float32 [rows, k], top_idx int32 [rows, k]): `owc_beam_candidates`."""
    rows = logits.shape[0]
    logz = torch.empty((rows,), dtype=F32, device=logits.device)
    top_val = torch.empty((rows, k), dtype=F32, device=logits.device)
    top_idx = torch.empty((rows, k), dtype=I32, device=logits.device)
    _call("owc_beam_candidates", _dev(logits), logits.data_ptr(), logits.stride(0), rows, logits.shape[1], int(k), logz.data_ptr(),
          top_val.data_ptr(), top_idx.data_ptr())
    return logz, top_val, top_idx


def sample_bf16(logits, temperature: float, top_k: int = 0, top_p: float | None = None, seed: int = 0, stream_ids=None, row_map=None,
                step: int = 0):
    """One draw per row of bf16 logits [rows, vocab] by owc_sampling (HF temperature -> top-k -> top-p -> multinomial; the library's
    Philox stream: key = seed, counter = (stream id of the row's ORIGINAL row, step))."""
    import ctypes as C

    out = torch.empty((logits.shape[0],), dtype=I32, device=logits.device)
    sp = _lib.Sampling(float(temperature), int(top_k or 0), float(top_p) if top_p else 0.0, int(seed) & 0xFFFFFFFFFFFFFFFF,
                       _lib.ptr(stream_ids), None)
    _call("owc_sample_bf16", _dev(logits), logits.data_ptr(), logits.stride(0), logits.shape[0], logits.shape[1], C.byref(sp),
          _lib.ptr(row_map), int(step), out.data_ptr())
    return out


def token_logprob_bf16(logits, target):
    """log softmax(logits[r])[target[r]] (fp32; 0 where target[r] < 0) for bf16 logits [rows, vocab], int32 targets [rows]."""
    out = torch.empty((logits.shape[0],), dtype=torch.float32, device=logits.device)
    _call("owc_token_logprob_bf16", _dev(logits), logits.data_ptr(), logits.stride(0), target.data_ptr(), logits.shape[0],
          logits.shape[1], out.data_ptr())
    return out


def patchify_u8(images, mean, std, out=None):
    """uint8 [n,3,H,W] (H, W multiples of 28) -> pixel_values [n*(H/14)*(W/14), 1176] bf16."""
    assert images.dtype == torch.uint8 and images.dim() == 4 and images.is_contiguous()
    n, _, h, w = images.shape
    rows = n * (h // 14) * (w // 14)
    if out is None:
        out = torch.empty((rows, 1176), dtype=BF16, device=images.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    _call("owc_patchify_u8", _dev(images), images.data_ptr(), out.data_ptr(), out.stride(0), n, h, w, m, s)
    return out


def paired_dot(a, b):
    assert a.dtype == F32 and b.dtype == F32 and a.shape == b.shape and a.is_contiguous() and b.is_contiguous()
    out = torch.empty((a.shape[0],), dtype=F32, device=a.device)
    _call("owc_paired_dot", _dev(a), a.data_ptr(), b.data_ptr(), a.shape[0], a.shape[1], out.data_ptr())
    return out


def cosine_topk(preds, classes, k, label=None):
    """Top-k classes per prediction (+ the paired cosine with `label`), N x C never materialised."""
    assert preds.dtype == F32 and classes.dtype == F32 and preds.is_contiguous() and classes.is_contiguous()
    n, d = preds.shape
    c = classes.shape[0]
    top_val = torch.empty((n, k), dtype=F32, device=preds.device)
    top_idx = torch.empty((n, k), dtype=I32, device=preds.device)
    paired = torch.empty((n,), dtype=F32, device=preds.device) if label is not None else None
    _call("owc_cosine_topk", _dev(preds), preds.data_ptr(), classes.data_ptr(), ptr(label), n, c, d, k,
          top_val.data_ptr(), top_idx.data_ptr(), ptr(paired))
    return top_val, top_idx, paired
