"""numpy restatements of the primitive ops (TEST INFRASTRUCTURE — see oracle/__init__.py).

Each function restates what the reference's third-party arithmetic (torch / HF transformers, the
code `src/models/_qwen2_vl.py:319-329` and `src/data/pipelines/text/_text.py:197-202` call) computes,
in float32/float64 numpy.  `bf16=True` rounds to bfloat16 (round-to-nearest-even) at the points where a
torch.bfloat16 module rounds, so the HIP kernels can be compared op by op.
"""

from __future__ import annotations

import math

import numpy as np

_erf = np.vectorize(math.erf, otypes=[np.float64])


def bf16_round(x: np.ndarray) -> np.ndarray:
    """Round float32 values to the nearest bfloat16 (ties to even); returns float32."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + r) & np.uint32(0xFFFF0000)).view(np.float32)


def maybe_bf16(x: np.ndarray, bf16: bool) -> np.ndarray:
    return bf16_round(x) if bf16 else np.asarray(x, dtype=np.float32)


def linear(x, w, b=None, *, bf16=False):
    """torch.nn.functional.linear: x[M,K] @ w[N,K].T + b (fp32 accumulate, one rounding)."""
    y = np.asarray(x, np.float32) @ np.asarray(w, np.float32).T   # asarray: no copy of an already-fp32 weight
    if b is not None:
        y = y + np.asarray(b, np.float32)
    return maybe_bf16(y, bf16)


def quick_gelu(x, *, bf16=False):
    """HF ACT2FN['quick_gelu']: x * sigmoid(1.702 x)."""
    x = x.astype(np.float32)
    return maybe_bf16(x / (1.0 + np.exp(-1.702 * x)), bf16)


def gelu_erf(x, *, bf16=False):
    """torch.nn.GELU() (exact erf form)."""
    x64 = x.astype(np.float64)
    return maybe_bf16((0.5 * x64 * (1.0 + _erf(x64 / math.sqrt(2.0)))).astype(np.float32), bf16)


def silu(x, *, bf16=False):
    x = x.astype(np.float32)
    return maybe_bf16(x / (1.0 + np.exp(-x)), bf16)


def layer_norm(x, w, b, eps, *, bf16=False):
    """torch.nn.LayerNorm over the last dim (statistics in fp32, biased variance)."""
    x = x.astype(np.float32)
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    y = (x - mu) / np.sqrt(var + eps) * w.astype(np.float32) + b.astype(np.float32)
    return maybe_bf16(y, bf16)


def rms_norm(x, w, eps, *, bf16=False):
    """Qwen2VLRMSNorm (HF modeling_qwen2_vl.py:105-110): fp32 stats, cast, then weight multiply."""
    x = x.astype(np.float32)
    var = (x * x).mean(-1, keepdims=True)
    xn = maybe_bf16(x / np.sqrt(var + eps), bf16)
    return maybe_bf16(w.astype(np.float32) * xn, bf16)


def softmax(x, axis=-1):
    x = x.astype(np.float32)
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)
