"""Torch-tensor front end of the op-level C ABI (include/owc.h).

Tensors are only device memory handles here: every function passes raw pointers and sizes to
libowc_hip.so on torch's current HIP stream.  No torch arithmetic happens in this module.
"""

from __future__ import annotations

import torch

from . import _lib

__all__ = ["gemm_bf16"]


def _dev(t: torch.Tensor) -> int:
    if not t.is_cuda:
        raise _lib.OwcError("libowc_hip needs device tensors (no CPU fallback on the product path)")
    return t.device.index or 0


def gemm_bf16(
    a: torch.Tensor,
    w: torch.Tensor,
    bias: torch.Tensor | None = None,
    *,
    epilogue: int = _lib.EPI_NONE,
    residual: torch.Tensor | None = None,
    out: torch.Tensor | None = None,
) -> torch.Tensor:
    """``out[M,N] = a[M,K] @ w[N,K].T (+bias)`` with a fused epilogue (nn.Linear semantics)."""
    assert a.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert a.dim() == 2 and w.dim() == 2 and a.shape[1] == w.shape[1]
    assert a.stride(1) == 1 and w.stride(1) == 1
    m, k = a.shape
    n = w.shape[0]
    n_out = n // 2 if epilogue == _lib.EPI_SWIGLU else n
    if out is None:
        dt = torch.float32 if epilogue == _lib.EPI_F32 else torch.bfloat16
        out = torch.empty((m, n_out), dtype=dt, device=a.device)
    assert out.stride(1) == 1 and out.shape == (m, n_out)
    dev = _dev(a)
    rc = _lib.load().owc_gemm_bf16(
        _lib.ctx(dev), a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), _lib.ptr(bias),
        _lib.ptr(residual), residual.stride(0) if residual is not None else 0, out.data_ptr(),
        out.stride(0), m, n, k, epilogue, _lib.stream_ptr(),
    )
    _lib.check(rc, dev)
    return out
