"""fp8 (OCP e4m3fn) decoder path on the GPU, through the C ABI, against oracle/fp8_np.py.

Parity bars (stated; the reference has no fp8 path):
* quantiser: codes and scales bit-identical to the oracle;
* fp8 GEMM on identical quantised operands: products are exact in fp32, only the summation order differs ->
  bf16 outputs within 1 bf16 ulp of the oracle's exactly-summed result, >= 95 % bit-identical;
* model level (tests/test_fp8_model_gpu.py): logits vs the fp8 oracle within 3 %, vs the bf16 model within a stated bound.
"""
import numpy as np
import pytest
import torch

from oracle import fp8_np as F
from oracle import np_ops
from tests.util import assert_bf16_close, bf16_randn, to_np

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("rows,cols", [(1, 8), (5, 256), (131, 3584), (64, 8192), (9, 29568)])
def test_quantize_rows_exact(gpu, rows, cols):
    from lmms_owc_amd import ops

    x = bf16_randn((rows, cols), rows * 7 + cols, 3.0, gpu)
    x[0, : min(cols, 16)] = 0.0
    if rows > 2:
        x[2] = 0.0  # all-zero row -> scale 1, codes 0
        x[1, 3] = 1000.0  # an outlier sets that row's scale
    q, s = ops.quantize_rows_fp8(x)
    wq, ws = F.quantize_rows(to_np(x))
    assert np.array_equal(to_np(s), ws)
    got = q.cpu().numpy()
    same = got == wq
    # +0 and -0 are the same number (0x00 / 0x80)
    assert np.all(same | (((got & 0x7F) == 0) & ((wq & 0x7F) == 0)))


@pytest.mark.parametrize("m,n,k,epi", [(256, 256, 128, "none"), (300, 520, 384, "bias"), (1024, 1536, 3584, "none"),
                                       (129, 264, 256, "res"), (512, 1024, 512, "swiglu"), (40, 64, 8192, "bias")])
def test_gemm_fp8_matches_oracle(gpu, m, n, k, epi):
    from lmms_owc_amd import _lib, ops
    from lmms_owc_amd.engine.qwen2vl import interleave_gate_up

    x = bf16_randn((m, k), m + n, 1.0, gpu)
    w = bf16_randn((n, k), k + 1, 0.05, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    nxq, nxs, nwq, nws = xq.cpu().numpy(), to_np(xs), wq.cpu().numpy(), to_np(ws)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    if epi == "swiglu":
        f = n // 2
        # rows [0, f) = gate, [f, 2f) = up; the kernel wants them interleaved per 16 (codes and scales alike)
        gq = interleave_gate_up(wq[:f].view(torch.int8), wq[f:].view(torch.int8)).view(torch.uint8)
        gs = interleave_gate_up(ws[:f, None], ws[f:, None])[:, 0].contiguous()
        out = to_np(ops.gemm_fp8(xq, xs, gq, gs, epilogue=_lib.EPI_SWIGLU))
        y = F.linear_fp8(None, nwq, nws, xq=nxq, xs=nxs)
        g, u = y[:, :f], y[:, f:]
        want = np_ops.bf16_round(np_ops.bf16_round(g / (1.0 + np.exp(-g))) * u)
        assert np.abs(out - want).max() <= 2.0 ** -6 * np.abs(want).max()
        return
    if epi == "res":
        r = bf16_randn((m, n), 9, 1.0, gpu)
        out = to_np(ops.gemm_fp8(xq, xs, wq, ws, epilogue=_lib.EPI_RESIDUAL, residual=r))
        want = np_ops.bf16_round(F.linear_fp8(None, nwq, nws, xq=nxq, xs=nxs) + to_np(r))
        assert np.abs(out - want).max() <= 2.0 ** -7 * np.abs(want).max()
        return
    out = to_np(ops.gemm_fp8(xq, xs, wq, ws, bias))
    want = F.linear_fp8(None, nwq, nws, None if bias is None else to_np(bias), xq=nxq, xs=nxs)
    # helper unit: 2^-8 relative, so 2.0 = one true bf16 ulp (a rounding-boundary flip under a different summation order)
    assert_bf16_close(out, want, ulps=2.0, min_exact=0.95, atol=2.0 ** -9 * np.abs(want).max())


def test_fp8_quantisation_error_is_small(gpu):
    """The redefined-parity yardstick at op level: fp8 linear vs the bf16 linear on the same operands."""
    from lmms_owc_amd import ops

    x = bf16_randn((512, 4096), 1, 1.0, gpu)
    w = bf16_randn((1024, 4096), 2, 0.02, gpu)
    y16 = to_np(ops.gemm_bf16(x, w))
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    y8 = to_np(ops.gemm_fp8(xq, xs, wq, ws))
    rel = np.linalg.norm(y8 - y16) / np.linalg.norm(y16)
    assert rel < 0.06, rel   # e4m3 (3 mantissa bits) on both operands: ~2^-4 per element / sqrt statistics


@pytest.mark.parametrize("rows,d", [(7, 256), (300, 1536), (65, 3584), (33, 8192)])
def test_rmsnorm_quant_is_the_two_kernels_fused(gpu, rows, d):
    """The fused RMSNorm + quantise kernel of the fp8 decoder is bit-identical to owc_rmsnorm_bf16 followed by owc_quantize_rows_fp8."""
    from lmms_owc_amd import ops

    x = bf16_randn((rows, d), rows + d, 2.0, gpu)
    w = (1.0 + 0.1 * bf16_randn((d,), 3, 1.0, gpu).float()).to(torch.bfloat16)
    q1, s1 = ops.quantize_rows_fp8(ops.rmsnorm(x, w, 1e-6))
    q2, s2 = ops.rmsnorm_quant_fp8(x, w, 1e-6)
    assert torch.equal(s1, s2) and torch.equal(q1, q2)


@pytest.mark.parametrize("m", [1, 16, 33, 64])
@pytest.mark.parametrize("epi", ["bias", "res", "swiglu"])
def test_gemm_fp8_skinny_kernel(gpu, m, epi):
    """The weight-streaming fp8 kernel for M <= 64 (behind "gemm_skinny_max_m" since round 3): oracle parity, BIT-IDENTICAL to the tiled
    fp8 kernels, and row 0 alone equals row 0 inside the batch."""
    from lmms_owc_amd import _lib, ops
    from lmms_owc_amd.engine.qwen2vl import interleave_gate_up

    n, k = 1216, 3584
    lib = _lib.load()
    x = bf16_randn((m, k), 70 + m, 1.0, gpu)
    w = bf16_randn((n, k), 71, 0.03, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    r = bf16_randn((m, n), 9, 1.0, gpu) if epi == "res" else None
    if epi == "swiglu":
        f = n // 2
        wq = interleave_gate_up(wq[:f].view(torch.int8), wq[f:].view(torch.int8)).view(torch.uint8)
        ws = interleave_gate_up(ws[:f, None], ws[f:, None])[:, 0].contiguous()
    kw = dict(epilogue={"bias": _lib.EPI_NONE, "res": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi])
    run = lambda q, s_, rr: ops.gemm_fp8(q, s_, wq, ws, bias, residual=rr, **kw)  # noqa: E731
    try:
        lib.owc_tuning_set(b"gemm_skinny_max_m", 64)   # (off by default since round 3: the ring kernel is faster; kept behind the knob)
        out = run(xq, xs, r)
        row0 = run(xq[:1], xs[:1], None if r is None else r[:1])[0]
        lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
        tiled = run(xq, xs, r)
    finally:
        lib.owc_tuning_set(b"gemm_skinny_max_m", -1)
    assert torch.equal(out, tiled)
    assert torch.equal(row0, out[0])
    assert torch.equal(run(xq[:1], xs[:1], None if r is None else r[:1])[0], out[0])   # the default dispatch (ring kernel), one row
    if epi == "bias":
        want = F.linear_fp8(None, wq.cpu().numpy(), to_np(ws), to_np(bias), xq=xq.cpu().numpy(), xs=to_np(xs))
        assert_bf16_close(to_np(out), want, ulps=2.0, min_exact=0.95, atol=2.0 ** -9 * np.abs(want).max())


@pytest.mark.parametrize("m,n,k,epi", [(300, 520, 384, "bias"), (1000, 1536, 1024, "res"), (2048, 4096, 512, "bias"), (512, 1024, 512, "swiglu")])
def test_gemm_fp8_tile_sizes_bit_identical(gpu, m, n, k, epi):
    """64x64 tiles (few-tile shapes) and 256x256 tiles give the same bits: one ascending chain of scaled MFMAs per output."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    x = bf16_randn((m, k), 80 + m, 1.0, gpu)
    w = bf16_randn((n, k), 81, 0.05, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    r = bf16_randn((m, n), 9, 1.0, gpu) if epi == "res" else None
    e = {"bias": _lib.EPI_NONE, "res": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi]
    outs = []
    for knob in (1, 0):
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256 if knob else 0)
        outs.append(ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r))
    lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("m,n,k,epi", [(256, 8192, 29568, "res"), (130, 1024, 128, "bias"), (70, 520, 256, "bias"), (512, 2048, 8192, "swiglu")])
def test_gemm_fp8_mid_kernel_race_screen(gpu, m, n, k, epi):
    """The fp8 64x64 kernel's four-stage LDS-DMA ring (counted vmcnt + one raw barrier per K-tile) against the 256x256 kernel, 15 times
    per shape: 231 K-tiles (the 72B down projection at decode batch 256), 1 and 2 K-tiles (fewer than stages: the over-issued pieces),
    ragged M / N, SwiGLU.  Bit-identical every time."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    x = bf16_randn((m, k), 180 + m, 1.0, gpu)
    w = bf16_randn((n, k), 181, 0.05, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    r = bf16_randn((m, n), 9, 1.0, gpu) if epi == "res" else None
    e = {"bias": _lib.EPI_NONE, "res": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi]
    try:
        lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)
        want = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        for i in range(15):
            got = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)
            assert torch.equal(got, want), (i, int((got != want).sum()))
    finally:
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        lib.owc_tuning_set(b"gemm_skinny_max_m", -1)


@pytest.mark.parametrize("m,n,k,epi", [(64, 8192, 29696, "res"), (40, 1096, 128, "bias"), (130, 8192, 8192, "res"), (100, 10240, 8192, "bias"),
                                       (128, 59392, 8192, "swiglu"), (24, 40000, 256, "bias"), (70, 33024, 3584, "swiglu"), (300, 2048, 8192, "res"),
                                       (520, 4160, 512, "bias"), (700, 8192, 1024, "res"), (513, 10240, 256, "bias")])
def test_gemm_fp8_ring_shapes_race_screen(gpu, m, n, k, epi):
    """The fp8 ring kernel's round-3 forms - 32x32 / 64x32 tiles and several K-tiles per stage for the narrow projections (72B o / down /
    qkv at decode batch 17-256), 32/64/128 x 256 tiles for <= 128 rows x tens of thousands of columns (gate/up), 64x64 with two K-tiles
    per stage, round 4's 128x64 tiles for several hundred rows (the last three cases: 72B o / qkv at decode batch 257-768) - against the
    256x256 kernel, 8 times per shape: ragged M / N, 1 and 2 K-tiles (fewer than stages), SwiGLU.  Bit-identical."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    x = bf16_randn((m, k), 280 + m, 1.0, gpu)
    w = bf16_randn((n, k), 281, 0.05, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    r = bf16_randn((m, n), 9, 1.0, gpu) if epi == "res" else None
    e = {"bias": _lib.EPI_NONE, "res": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi]
    try:
        lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)
        lib.owc_tuning_set(b"gemm_small_tiles", 0)
        want = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)      # 256x256 kernels
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        mid = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)       # 64x64 tiles, one K-tile per stage
        assert torch.equal(mid, want)
        lib.owc_tuning_set(b"gemm_small_tiles", 1)
        for i in range(8):
            got = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)
            assert torch.equal(got, want), (i, int((got != want).sum()))
    finally:
        lib.owc_tuning_set(b"gemm_small_tiles", 1)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        lib.owc_tuning_set(b"gemm_skinny_max_m", -1)


@pytest.mark.parametrize("m,n,k,epi", [(2048, 4096, 512, "bias"), (1024, 2048, 8192, "res"), (1000, 1536, 1024, "res"), (1024, 2048, 3584, "swiglu"),
                                       (512, 1024, 256, "bias"), (700, 1096, 768, "bias")])
def test_gemm_fp8_pingpong_bit_identical_race_screen(gpu, m, n, k, epi):
    """gemm_fp8_nt_256pp_kernel (two waves per SIMD alternate MFMA / load roles, counted vmcnt, two W half-0 register sets
    alternating per K-tile, peeled two-tile tail) against the lock-step 256x256 fp8 kernel: 2 ... 64 K-tiles (the minimum: only the
    peeled tail runs), ragged M / N, every epilogue, 12 launches each - the same bits every time.  An odd K-tile count (768 = 6 tiles
    is even; 29568 = 231 is not) stays on the lock-step kernel by dispatch."""
    from lmms_owc_amd import _lib, ops
    from lmms_owc_amd.engine.qwen2vl import interleave_gate_up

    lib = _lib.load()
    x = bf16_randn((m, k), 280 + m, 1.0, gpu)
    w = bf16_randn((n, k), 281, 0.05, gpu)
    xq, xs = ops.quantize_rows_fp8(x)
    wq, ws = ops.quantize_rows_fp8(w)
    bias = bf16_randn((n,), 5, 0.5, gpu) if epi == "bias" else None
    r = bf16_randn((m, n), 9, 1.0, gpu) if epi == "res" else None
    if epi == "swiglu":
        f = n // 2
        wq = interleave_gate_up(wq[:f].view(torch.int8), wq[f:].view(torch.int8)).view(torch.uint8)
        ws = interleave_gate_up(ws[:f, None], ws[f:, None])[:, 0].contiguous()
    e = {"bias": _lib.EPI_NONE, "res": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi]
    try:
        lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)          # always the 256x256 kernels
        assert lib.owc_tuning_set(b"gemm_pingpong", 1) == 0    # bf16 ping-pong, fp8 lock-step
        want = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)
        assert lib.owc_tuning_set(b"gemm_pingpong", 2) == 0    # fp8 ping-pong as well
        for i in range(12):
            got = ops.gemm_fp8(xq, xs, wq, ws, bias, epilogue=e, residual=r)
            assert torch.equal(got, want), (i, int((got != want).sum()))
    finally:
        lib.owc_tuning_set(b"gemm_pingpong", -1)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        lib.owc_tuning_set(b"gemm_skinny_max_m", -1)
