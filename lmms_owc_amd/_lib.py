"""ctypes binding of libowc_hip.so (C ABI: include/owc.h).

The product path has no CPU fallback: if the shared library is missing or a call fails the
error is raised, never swallowed.  ``oracle/`` is test infrastructure and is never imported here.
"""

from __future__ import annotations

import ctypes as C
import threading
from pathlib import Path

_LIB_PATH = Path(__file__).resolve().parent / "libowc_hip.so"
_lock = threading.Lock()
_lib: C.CDLL | None = None
_ctx: dict[int, C.c_void_p] = {}

ABI_VERSION = 1

EPI_NONE, EPI_QUICK_GELU, EPI_GELU_ERF, EPI_RESIDUAL, EPI_SWIGLU, EPI_F32 = range(6)


class OwcError(RuntimeError):
    """A libowc_hip.so call returned a negative status."""


def lib_path() -> Path:
    return _LIB_PATH


def _declare(lib: C.CDLL) -> None:
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    lib.owc_abi_version.restype = i32
    lib.owc_abi_version.argtypes = []
    lib.owc_init.restype = i32
    lib.owc_init.argtypes = [i32, C.POINTER(vp)]
    lib.owc_destroy.restype = i32
    lib.owc_destroy.argtypes = [vp]
    lib.owc_last_error.restype = C.c_char_p
    lib.owc_last_error.argtypes = [vp]
    lib.owc_gemm_bf16.restype = i32
    lib.owc_gemm_bf16.argtypes = [vp, vp, i64, vp, i64, vp, vp, i64, vp, i64, i32, i32, i32, i32, vp]


def load() -> C.CDLL:
    """Load libowc_hip.so; raises (loudly) if it has not been built."""
    global _lib
    with _lock:
        if _lib is None:
            if not _LIB_PATH.exists():
                raise OwcError(
                    f"{_LIB_PATH} is missing: run `python -m lmms_owc_amd.build` "
                    "(the HIP extension is mandatory, there is no fallback path)"
                )
            lib = C.CDLL(str(_LIB_PATH))
            _declare(lib)
            if lib.owc_abi_version() != ABI_VERSION:
                raise OwcError("libowc_hip.so ABI version mismatch: rebuild the extension")
            _lib = lib
        return _lib


def ctx(device: int = 0) -> C.c_void_p:
    """Per-device library context (needs a GPU)."""
    lib = load()
    with _lock:
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


def stream_ptr() -> int:
    import torch

    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    return None if t is None else t.data_ptr()
