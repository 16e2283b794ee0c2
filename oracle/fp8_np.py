"""numpy restatement of the fp8 (OCP e4m3fn) decoder arithmetic (TEST INFRASTRUCTURE ONLY).

No reference counterpart: the reference's only reduced-precision path is bitsandbytes int8 / nf4
(`src/models/_base.py:116-121`); BASELINE.json config #5 asks for an fp8 MFMA decoder instead, so parity is
REDEFINED here as: (1) the quantiser and the fp8 GEMM are bit-defined by this file and the
HIP kernels must reproduce it (bytes and scales exactly; products are exact in fp32, sums to accumulation order),
(2) the fp8 model is compared with the bf16 model through a stated logit tolerance and top-1 agreement rate.

Format: OCP FP8 E4M3 "fn" (1-4-3, bias 7, no infinities, max 448, 0x7f/0xff = NaN) - what gfx950's v_cvt_pk_fp8_f32 and
the f8f6f4 MFMA use (not MI300's fnuz).  Scheme: per-row symmetric scales, s = max|x| / 448 (1 for an all-zero row),
q = rne_e4m3(x / s); activations per token (dynamic), weights per output channel (static).
"""

from __future__ import annotations

import numpy as np

from .np_ops import bf16_round, maybe_bf16

E4M3_MAX = np.float32(448.0)


def _decode_table() -> np.ndarray:
    t = np.zeros(256, np.float32)
    for c in range(256):
        s, e, m = c >> 7, (c >> 3) & 15, c & 7
        if e == 15 and m == 7:
            v = np.nan
        elif e == 0:
            v = m * 2.0 ** -9
        else:
            v = (1 + m / 8.0) * 2.0 ** (e - 7)
        t[c] = -v if s else v
    return t


E4M3_DECODE = _decode_table()
_POS = E4M3_DECODE[:127].astype(np.float64)  # codes 0..126 ascending (0 .. 448)


def e4m3_encode_search(x: np.ndarray) -> np.ndarray:
    """float32 -> e4m3fn codes, round to nearest even, saturating at +-448: the DEFINITION (nearest entry of the decode
    table, ties to the even code).  `e4m3_encode` is the fast form and is tested equal to this one."""
    x = np.asarray(x, np.float32)
    a = np.minimum(np.abs(x).astype(np.float64), 448.0)
    hi = np.searchsorted(_POS, a, side="left").clip(0, 126)
    lo = (hi - 1).clip(0, 126)
    dlo, dhi = a - _POS[lo], _POS[hi] - a
    pick_hi = (dhi < dlo) | ((dhi == dlo) & ((hi & 1) == 0))   # tie -> even code (mantissa LSB 0)
    code = np.where(pick_hi, hi, lo).astype(np.uint8)
    return np.where(np.signbit(x), code | 0x80, code).astype(np.uint8)


def _encode_block(x: np.ndarray) -> np.ndarray:
    a = np.minimum(np.abs(x), E4M3_MAX)
    # |x| >= 2^-6 (normal codes): round the float32 mantissa to 3 bits, ties to even, on the bit pattern
    u = a.view(np.uint32)
    u = (u + (((u >> np.uint32(20)) & np.uint32(1)) + np.uint32(0x7FFFF))) & np.uint32(0xFFF00000)
    normal = (((u >> np.uint32(23)) - np.uint32(120)) << np.uint32(3)) | ((u >> np.uint32(20)) & np.uint32(7))
    # |x| < 2^-6 (subnormal codes 0..7, and 8 = 2^-6 when rounding carries up): multiples of 2^-9, np.rint is ties-to-even
    sub = np.rint(a * np.float32(512.0)).astype(np.uint32)
    code = np.where(a >= np.float32(2.0 ** -6), normal, sub).astype(np.uint8)
    return np.where(np.signbit(x), code | np.uint8(0x80), code).astype(np.uint8)


def e4m3_encode(x: np.ndarray) -> np.ndarray:
    """float32 -> e4m3fn codes (RNE, saturating at +-448); bit arithmetic in row blocks on a thread pool (the 72B-width slice
    quantises 1.7 G weights).  Equal to `e4m3_encode_search` on every input (tests/test_oracle_fp8.py)."""
    x = np.ascontiguousarray(x, np.float32)
    if x.size <= (1 << 22):
        return _encode_block(x.reshape(-1)).reshape(x.shape)
    import concurrent.futures as cf
    import os

    flat, out = x.reshape(-1), np.empty(x.size, np.uint8)
    step = 1 << 22

    def block(i):
        out[i:i + step] = _encode_block(flat[i:i + step])

    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        list(ex.map(block, range(0, x.size, step)))
    return out.reshape(x.shape)


def e4m3_decode(q: np.ndarray) -> np.ndarray:
    return E4M3_DECODE[np.asarray(q, np.uint8)]


def quantize_rows(x: np.ndarray):
    """[rows, cols] float32 (bf16-representable in the decoder) -> (codes uint8, scale float32[rows])."""
    x = np.asarray(x, np.float32)
    amax = np.abs(x).max(axis=-1)
    scale = np.where(amax > 0, amax / E4M3_MAX, np.float32(1.0)).astype(np.float32)
    return e4m3_encode((x / scale[..., None]).astype(np.float32)), scale


_DECODED: dict = {}          # id(codes) -> (codes, decoded weight): decode a weight once per process, not once per call
_F64_MAX_ELEMS = 1 << 24     # above this a weight is kept / multiplied in float32 (see linear_fp8)


def _decoded_weight(wq: np.ndarray) -> np.ndarray:
    hit = _DECODED.get(id(wq))
    if hit is None or hit[0] is not wq:
        dt = np.float64 if wq.size <= _F64_MAX_ELEMS else np.float32
        if len(_DECODED) > 64:
            _DECODED.clear()
        hit = _DECODED[id(wq)] = (wq, e4m3_decode(wq).astype(dt))
    return hit[1]


def linear_fp8(x: np.ndarray, wq: np.ndarray, ws: np.ndarray, bias=None, *, bf16=True, xq=None, xs=None) -> np.ndarray:
    """y = bf16(((q(x) . wq^T) * sx[m] * sw[n]) + bias): x [T, K] -> [T, N]; wq uint8 [N, K], ws float32 [N].
    Products of two e4m3 values are exact in fp32; the sum over K is exact in float64 (weights up to 2^24 elements).  Larger
    weights (the 7B..72B-width slices of tests/test_decode_parity_gpu.py) are decoded to float32 and summed by a float32
    sgemm - relative error <= 1e-6, three orders below the bf16 rounding of the output - so those oracles run in seconds."""
    if xq is None:
        xq, xs = quantize_rows(x)
    wd = _decoded_weight(wq)
    acc = e4m3_decode(xq).astype(wd.dtype) @ wd.T
    y = acc.astype(np.float32) * xs[:, None].astype(np.float32) * ws[None, :].astype(np.float32)
    if bias is not None:
        y = y + np.asarray(bias, np.float32)
    return maybe_bf16(y.astype(np.float32), bf16)


def quantize_decoder(w: dict, prefix: str, n_layers: int) -> dict:
    """Per-output-channel e4m3 weights for the decoder projections of a HF state dict (q/k/v/o/gate/up/down);
    returns name -> (codes, scales).  Embedding, norms and lm_head stay bf16."""
    out = {}
    for i in range(n_layers):
        for n in ("self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "self_attn.o_proj", "mlp.gate_proj", "mlp.up_proj", "mlp.down_proj"):
            k = f"{prefix}layers.{i}.{n}.weight"
            out[k] = quantize_rows(bf16_round(w[k]))
    return out
