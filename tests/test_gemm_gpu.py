"""HIP bf16 GEMM (owc_gemm_bf16) vs the numpy oracle — parity through the C ABI."""
import numpy as np
import pytest
import torch

from oracle import np_ops
from tests.util import assert_bf16_close, bf16_randn, to_np

pytestmark = pytest.mark.gpu
SKINNY_MAX_M = -1   # "gemm_skinny_max_m" < 0 restores the library's defaults (tests that move the knob put it back)

SHAPES = [
    # (M, N, K): ragged M / N, K tails (K % 64 != 0), patch-embed K = 1176
    (1, 128, 64),
    (7, 136, 72),
    (128, 128, 64),
    (286, 1536, 1536),
    (300, 264, 1176),
    (1024, 3840, 1280),
    (513, 2048, 8960),
    (64, 1280, 5120),
    # >= 144 tiles of 256x256 and K % 64 == 0 -> the 256x256 quadrant-pipelined kernel (full and ragged tiles)
    (4096, 3072, 1280),
    (4000, 3080, 1344),
    (3700, 4096, 64),
]


def _oracle(a, w, b):
    return np_ops.linear(to_np(a), to_np(w), None if b is None else to_np(b), bf16=True)


@pytest.mark.parametrize("m,n,k", SHAPES)
def test_gemm_none(gpu, m, n, k):
    from lmms_owc_amd import _lib, ops

    a = bf16_randn((m, k), 1, device=gpu)
    w = bf16_randn((n, k), 2, 0.05, device=gpu)
    b = bf16_randn((n,), 3, device=gpu)
    out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_NONE)
    torch.cuda.synchronize()
    assert_bf16_close(to_np(out), _oracle(a, w, b), atol=1e-4)


def test_gemm_asymmetric_identity(gpu):
    """A = I with an asymmetric W catches a transposed C-write or swapped MFMA operand maps."""
    from lmms_owc_amd import ops

    n = k = 256
    a = torch.eye(k, dtype=torch.bfloat16, device=gpu)
    w = (torch.arange(n * k, device=gpu).reshape(n, k) % 251).to(torch.bfloat16)
    out = ops.gemm_bf16(a, w)
    torch.cuda.synchronize()
    assert torch.equal(out, w.t().contiguous())


@pytest.mark.parametrize("epi", ["quick_gelu", "gelu_erf", "residual", "swiglu", "f32"])
def test_gemm_epilogues(gpu, epi):
    from lmms_owc_amd import _lib, ops

    m, n, k = 200, 512, 320
    a = bf16_randn((m, k), 4, device=gpu)
    w = bf16_randn((n, k), 5, 0.08, device=gpu)
    b = bf16_randn((n,), 6, device=gpu)
    y = _oracle(a, w, b)
    if epi == "quick_gelu":
        out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_QUICK_GELU)
        want = np_ops.quick_gelu(y, bf16=True)
    elif epi == "gelu_erf":
        out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_GELU_ERF)
        want = np_ops.gelu_erf(y, bf16=True)
    elif epi == "residual":
        r = bf16_randn((m, n), 7, device=gpu)
        out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_RESIDUAL, residual=r)
        want = np_ops.bf16_round(to_np(r) + y)
    elif epi == "f32":
        out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_F32)
        want = np_ops.linear(to_np(a), to_np(w), to_np(b))
        torch.cuda.synchronize()
        np.testing.assert_allclose(to_np(out), want, rtol=2e-5, atol=2e-4)
        return
    else:
        # gate/up rows interleaved in groups of 16: [g0..g15, u0..u15, g16.., u16..]
        f = n // 2
        wg = bf16_randn((f, k), 8, 0.08, device=gpu)
        wu = bf16_randn((f, k), 9, 0.08, device=gpu)
        wi = torch.stack([wg.view(f // 16, 16, k), wu.view(f // 16, 16, k)], dim=1).reshape(n, k).contiguous()
        out = ops.gemm_bf16(a, wi, None, epilogue=_lib.EPI_SWIGLU)
        g = np_ops.linear(to_np(a), to_np(wg), bf16=True)
        u = np_ops.linear(to_np(a), to_np(wu), bf16=True)
        want = np_ops.bf16_round(np_ops.silu(g, bf16=True) * u)
    torch.cuda.synchronize()
    # a residual add can cancel: allow one bf16 ulp of the pre-add magnitude as absolute slack
    atol = 2.0**-7 * float(np.abs(y).max()) if epi == "residual" else 1e-4
    assert_bf16_close(to_np(out), want, ulps=4.0, min_exact=0.80, atol=atol)


@pytest.mark.parametrize("epi", ["quick_gelu", "residual", "swiglu"])
def test_gemm_epilogues_large_kernel(gpu, epi):
    """Same epilogues on the 256x256 kernel (M, N ragged against the tile, in-place residual as the model uses it)."""
    from lmms_owc_amd import _lib, ops

    m, n, k = 3900, 4104, 512
    a = bf16_randn((m, k), 14, device=gpu)
    b = bf16_randn((n,), 16, device=gpu)
    if epi == "swiglu":
        n = 8192
        f = n // 2
        wg = bf16_randn((f, k), 18, 0.06, device=gpu)
        wu = bf16_randn((f, k), 19, 0.06, device=gpu)
        wi = torch.stack([wg.view(f // 16, 16, k), wu.view(f // 16, 16, k)], dim=1).reshape(n, k).contiguous()
        out = ops.gemm_bf16(a, wi, None, epilogue=_lib.EPI_SWIGLU)
        g = np_ops.linear(to_np(a), to_np(wg), bf16=True)
        u = np_ops.linear(to_np(a), to_np(wu), bf16=True)
        want = np_ops.bf16_round(np_ops.silu(g, bf16=True) * u)
        atol = 2.0**-7 * float(np.abs(u).max())  # silu(g)*u: one ulp of either factor, scaled by the other
    else:
        w = bf16_randn((n, k), 15, 0.06, device=gpu)
        y = _oracle(a, w, b)
        if epi == "quick_gelu":
            out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_QUICK_GELU)
            want = np_ops.quick_gelu(y, bf16=True)
            # the activation maps a 1-ulp difference of its INPUT (accumulation order) to up to ~1.1 input ulps of
            # output, which is many output ulps where gelu(x) ~ 0: allow one input ulp of absolute slack
            atol = 2.0**-7 * float(np.abs(y).max())
        else:
            r = bf16_randn((m, n), 17, device=gpu)
            want = np_ops.bf16_round(to_np(r) + y)
            out = ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_RESIDUAL, residual=r, out=r)  # in place: x += f(x)
            atol = 2.0**-7 * float(np.abs(y).max())
    torch.cuda.synchronize()
    assert_bf16_close(to_np(out), want, ulps=4.0, min_exact=0.80, atol=atol)


@pytest.mark.parametrize("m", [1, 7, 16, 17, 33, 64])
@pytest.mark.parametrize("epi", ["bias", "residual", "swiglu"])
def test_gemm_skinny_kernel(gpu, m, epi):
    """M <= 64 with K % 128 == 0, N % 16 == 0 runs the weight-streaming kernel (decode at small batch): same oracle, and the
    result is BIT-IDENTICAL to the tiled kernel's (one ascending MFMA accumulation chain per output in both) and independent
    of M (batch invariance)."""
    from lmms_owc_amd import _lib, ops

    n, k = 1216, 3584
    a = bf16_randn((m, k), 40 + m, device=gpu)
    b = bf16_randn((n,), 6, device=gpu)
    lib = _lib.load()
    if epi == "swiglu":
        f = n // 2
        wg = bf16_randn((f, k), 8, 0.03, device=gpu)
        wu = bf16_randn((f, k), 9, 0.03, device=gpu)
        w = torch.stack([wg.view(f // 16, 16, k), wu.view(f // 16, 16, k)], dim=1).reshape(n, k).contiguous()
        run = lambda x: ops.gemm_bf16(x, w, None, epilogue=_lib.EPI_SWIGLU)  # noqa: E731
        g = np_ops.linear(to_np(a), to_np(wg), bf16=True)
        u = np_ops.linear(to_np(a), to_np(wu), bf16=True)
        want = np_ops.bf16_round(np_ops.silu(g, bf16=True) * u)
        atol = 2.0 ** -7 * float(np.abs(want).max())   # a product of two rounded factors: one ulp of the largest output
    else:
        w = bf16_randn((n, k), 5, 0.03, device=gpu)
        y = _oracle(a, w, b)
        if epi == "residual":
            r = bf16_randn((m, n), 7, device=gpu)
            run = lambda x: ops.gemm_bf16(x, w, b, epilogue=_lib.EPI_RESIDUAL, residual=r[: x.shape[0]])  # noqa: E731
            want = np_ops.bf16_round(to_np(r) + y)
            atol = 2.0 ** -7 * float(np.abs(y).max())
        else:
            run = lambda x: ops.gemm_bf16(x, w, b)  # noqa: E731
            want, atol = y, 1e-4
    lib.owc_tuning_set(b"gemm_skinny_max_m", 64)   # every MT variant of the kernel, whatever the dispatch default is
    out = to_np(run(a))
    assert_bf16_close(out, want, ulps=4.0, min_exact=0.80, atol=atol)
    # the tiled kernel on the same operands (skinny kernel switched off)
    lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
    try:
        tiled = to_np(run(a))
    finally:
        lib.owc_tuning_set(b"gemm_skinny_max_m", SKINNY_MAX_M)
    assert np.array_equal(out, tiled)
    # row 0 alone gives the same bits as row 0 inside the batch
    assert np.array_equal(to_np(run(a[:1]))[0], out[0])


@pytest.mark.parametrize("m,n,k", [(286, 3584, 3584), (128, 1216, 1176), (500, 264, 640)])
def test_gemm_mid_kernel_bit_identical_to_128_tiles(gpu, m, n, k):
    """Shapes with few 128x128 tiles run 64x64 tiles; every kernel accumulates one ascending K chain per output, so the bits
    are the same whichever tile size computed them."""
    from lmms_owc_amd import _lib, ops

    a = bf16_randn((m, k), 60 + m, device=gpu)
    w = bf16_randn((n, k), 61, 0.05, device=gpu)
    b = bf16_randn((n,), 62, device=gpu)
    r = bf16_randn((m, n), 63, device=gpu)
    lib = _lib.load()
    outs = []
    for knob in (256, 0):
        lib.owc_tuning_set(b"gemm_mid_max_tiles", knob)
        outs.append((ops.gemm_bf16(a, w, b), ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_RESIDUAL, residual=r),
                     ops.gemm_bf16(a, w, b, epilogue=_lib.EPI_QUICK_GELU)))
    lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
    for x, y in zip(*outs):
        assert torch.equal(x, y)
    assert_bf16_close(to_np(outs[0][0]), _oracle(a, w, b), atol=1e-4)


@pytest.mark.parametrize("m,n,k,epi", [(256, 3584, 18944, "residual"), (64, 4736, 3584, "swiglu"), (130, 512, 128, "none"),
                                       (70, 264, 64, "none"), (300, 1216, 1176, "quick_gelu"), (512, 3584, 3584, "residual")])
def test_gemm_mid_kernel_race_screen(gpu, m, n, k, epi):
    """The 64x64 kernel's 4-stage LDS-DMA ring (counted vmcnt + one raw barrier per K-tile are its only ordering) against the
    128x128 kernel on the same operands, 20 times per shape: 296 K-tiles (the 7B down projection at decode batch 256), fewer K-tiles
    than stages (1 and 2: the over-issued zero-page pieces), a ragged K tail, ragged M / N, SwiGLU.  Bit-identical every time."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 80 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 81, 0.05, device=gpu)
    b = bf16_randn((n,), 82, device=gpu)
    r = bf16_randn((m, n), 83, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU, "quick_gelu": _lib.EPI_QUICK_GELU}[epi]

    def run():
        if epi == "swiglu":
            return ops.gemm_bf16(a, w, None, epilogue=E)
        return ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None)

    lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
    try:
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)
        want = run()
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 1 << 30)
        for i in range(20):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        lib.owc_tuning_set(b"gemm_skinny_max_m", SKINNY_MAX_M)


@pytest.mark.parametrize("m,n,k,epi", [(64, 3584, 18944, "residual"), (128, 3584, 3584, "residual"), (100, 4608, 3584, "none"),
                                       (70, 1096, 8256, "residual"),   # K >= 8192: two K-tiles per stage, an odd number of K-tiles
                                       (40, 1096, 128, "none"), (130, 520, 64, "residual"), (96, 4736, 3584, "swiglu"),
                                       (33, 264, 1024, "none")])
def test_gemm_small_tile_shapes_race_screen(gpu, m, n, k, epi):
    """The 64x64 ring kernel's 64x32 / 32x32 tile shapes (deeper rings, two of the four waves only staging when TM = 32),
    each FORCED on shapes with ragged M / N, one and two K-tiles (fewer than ring stages) and the 7B decode projections, against the
    64x64 shape and the 128x128 kernel, 10 times each: bit-identical (one ascending K chain per output whatever the tile)."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 90 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 91, 0.05, device=gpu)
    b = bf16_randn((n,), 92, device=gpu)
    r = bf16_randn((m, n), 93, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU}[epi]

    def run():
        if epi == "swiglu":
            return ops.gemm_bf16(a, w, None, epilogue=E)
        return ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None)

    lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
    try:
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)
        want = run()
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 1 << 30)
        for shape in (2, 3, 4, 1):   # forced 64x64, 64x32, 32x32 (SwiGLU always runs 64x64), then the default
            assert lib.owc_tuning_set(b"gemm_small_tiles", shape) == 0
            for i in range(10):
                got = run()
                assert torch.equal(got, want), (shape, i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_small_tiles", 1)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        lib.owc_tuning_set(b"gemm_skinny_max_m", SKINNY_MAX_M)


@pytest.mark.parametrize("m,n,k,epi", [(512, 3584, 18944, "residual"), (1100, 3584, 3584, "residual"), (700, 4608, 3584, "none"), (300, 3584, 128, "none"),
                                       (513, 3600, 64, "residual")])
def test_gemm_ring_128_race_screen(gpu, m, n, k, epi):
    """The ring kernel's 128x64 tiles (round 4: 257-1169 rows x few thousand columns - the narrow projections of a mid-batch decode
    step; one block per CU with two K-tiles per stage up to 256 blocks, two per CU above) against the 64x64 tiles and the
    128x128 kernel, 8 times each: bit-identical; ragged M / N, one and two K-tiles (fewer than ring stages), K = 18944."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 70 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 71, 0.05, device=gpu)
    b = bf16_randn((n,), 72, device=gpu)
    r = bf16_randn((m, n), 73, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL}[epi]

    def run():
        return ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None)

    try:
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)          # the 128x128 kernel
        want = run()
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 1 << 30)
        for knob in (0, 1):                                  # 64x64 ring tiles, then 128x64
            assert lib.owc_tuning_set(b"gemm_ring_128", knob) == 0
            for i in range(8):
                got = run()
                assert torch.equal(got, want), (knob, i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_ring_128", -1)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)


@pytest.mark.parametrize("m,n,k,epi", [(2048, 3584, 18944, "residual"), (2048, 3584, 3584, "residual"), (1300, 3584, 3584, "none"),
                                       (1500, 3000, 128, "residual"),    # two K-tiles: prologue + the peeled tail only
                                       (1100, 3520, 192, "none"),         # three K-tiles, ragged N (a 64-column last tile)
                                       (1281, 2600, 256, "residual"),     # four K-tiles (the 4-tile tail), one row into the sixth row tile
                                       (1900, 3584, 320, "none"),         # five K-tiles: one loop iteration + a 2-tile tail
                                       (2048, 4096, 448, "swiglu"), (1792, 3584, 1280, "quick_gelu")])
def test_gemm_256x128_pingpong_race_screen(gpu, m, n, k, epi):
    """Round 5: `gemm_bf16_nt_256x128pp_kernel` (role-alternating waves on a 256 x 128 tile behind a three-stage LDS-DMA ring, two
    K-tiles ahead, counted vmcnt(8) / vmcnt(6)) - the o / down projections of a decode step at 1024-2048 rows, where 256 x 256 tiles
    would leave more than half of the chip idle.  Against the 128 x 128 kernel (knob off), 10 launches per shape: bit-identical
    (one ascending K chain per output whatever the tile), for every length of the peeled tail (2-4 K-tiles), ragged M / N, every
    epilogue family the LDS-staged store takes."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 50 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 51, 0.05, device=gpu)
    b = bf16_randn((n,), 52, device=gpu)
    r = bf16_randn((m, n), 53, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU, "quick_gelu": _lib.EPI_QUICK_GELU}[epi]

    def run():
        if epi == "swiglu":
            return ops.gemm_bf16(a, w, None, epilogue=E)
        return ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None)

    try:
        assert lib.owc_tuning_set(b"gemm_pp128", 0) == 0
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 0)          # the 128x128 kernel whatever the tile count
        want = run()
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)
        assert lib.owc_tuning_set(b"gemm_pp128", 1) == 0      # from one tile of 256 x 128 (every shape here has <= 256 of them)
        for i in range(10):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_pp128", -1)
        lib.owc_tuning_set(b"gemm_mid_max_tiles", 256)


@pytest.mark.parametrize("m,n,k,epi", [(4352, 4096, 384, "none"),          # 272 tiles: 16 blocks walk two tiles; six K-tiles (the steady-state loop runs once)
                                       (4100, 3848, 512, "residual"),      # ragged M and N (clamped rows in the last tile row / column), eight K-tiles
                                       (8192, 5120, 1280, "quick_gelu"),   # 640 tiles = 2.5 rounds (the vision tower's fc1 shape)
                                       (16384, 1280, 1280, "inplace"),     # C aliases R (the residual stream): proj / fc2
                                       (6000, 2816, 512, "gelu_erf")])
def test_gemm_persistent_pingpong_race_screen(gpu, m, n, k, epi):
    """Round 5: `gemm_bf16_nt_256pp_persist_kernel` (off by default: `owc_tuning_set("gemm_persist", 1)`) - one block per CU walks its
    output tiles, the LDS-DMA ring runs on across tile boundaries (the next tile's K-tiles 0 / 1 ride in the last two K-tiles' spare
    slots), the C tile leaves in two batches as full 128-byte lines straight from registers (lanes fr / fr ^ 8 trade 16-byte pieces),
    counted waits that leave a batch of stores in flight.  Against the
    one-tile-per-block ping-pong kernel (knob off), 10 launches per shape: bit-identical - same K chain per output, same epilogue
    arithmetic; blocks with one / two / three tiles, ragged edges, the epilogue families the C ABI reaches, an output that aliases the
    residual.  (The rotary epilogue of the vision qkv projection: tests/test_fullsize_gpu.py::test_vision_tower_bits_with_and_without_the_persistent_gemm.)"""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 150 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 151, 0.05, device=gpu)
    b = bf16_randn((n,), 152, device=gpu)
    r = bf16_randn((m, n), 153, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "inplace": _lib.EPI_RESIDUAL, "quick_gelu": _lib.EPI_QUICK_GELU,
         "gelu_erf": _lib.EPI_GELU_ERF}[epi]

    def run():
        if epi == "inplace":
            out = r.clone()
            return ops.gemm_bf16(a, w, b, epilogue=E, residual=out, out=out)
        return ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None)

    try:
        assert lib.owc_tuning_set(b"gemm_persist", 0) == 0
        want = run()
        assert lib.owc_tuning_set(b"gemm_persist", 1) == 0
        for i in range(10):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_persist", -1)


@pytest.mark.parametrize("persist", [0, 1])
@pytest.mark.parametrize("m,n,k,epi", [(16384, 3584, 512, "residual"),     # 64 x 14 tiles: two column groups of 7
                                       (8192, 4608, 384, "none"),          # 32 x 18: three groups of 6
                                       (12288, 1280, 1280, "quick_gelu"),  # 48 x 5: one group of 5 (the vision proj / fc2 width)
                                       (9000, 3848, 256, "none"),          # ragged M and N; 36 x 16 tiles: two groups of 8
                                       (5000, 5120, 384, "gelu_erf"),      # 20 x 20 tiles: tiles_m == tiles_n, groups 7 / 7 / 6
                                       (2304, 6400, 384, "none")])         # 9 x 25: fewer tile rows than columns (4 groups of 7 / 6 / 6 / 6)
def test_gemm_tile_walk_does_not_change_a_bit(gpu, m, n, k, epi, persist):
    """Round 6: the block id -> output tile map of the 256x256 ping-pong kernels (`tile_origin`, gemm_bf16.hip) walks column groups of
    <= 8 tile columns down all tile rows (knob "gemm_walk": 2 for every shape, default 1 = where it measured faster: K >= 2.5 N),
    instead of rows of 4 across all columns: an XCD's 32 co-resident tiles then always span 12-12.6 operand panels instead of up to 16.  A tile is computed the
    same way wherever it runs: outputs must equal the rounds-1-5 walk bit for bit - which also proves the new map covers every
    tile exactly once (a tile computed twice leaves another one unwritten: the canary below would survive)."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 160 + (m % 89), device=gpu)
    w = bf16_randn((n, k), 161, 0.05, device=gpu)
    b = bf16_randn((n,), 162, device=gpu)
    r = bf16_randn((m, n), 163, device=gpu)
    E = {"none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "quick_gelu": _lib.EPI_QUICK_GELU, "gelu_erf": _lib.EPI_GELU_ERF}[epi]

    def run():
        out = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=gpu)   # canary: every element must be written
        ops.gemm_bf16(a, w, b, epilogue=E, residual=r if epi == "residual" else None, out=out)
        return out

    try:
        assert lib.owc_tuning_set(b"gemm_persist", persist) == 0
        assert lib.owc_tuning_set(b"gemm_walk", 0) == 0
        want = run()
        assert not torch.isnan(want).any()
        assert lib.owc_tuning_set(b"gemm_walk", 2) == 0     # column groups whatever the shape
        for i in range(3):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_walk", -1)
        lib.owc_tuning_set(b"gemm_persist", -1)


@pytest.mark.parametrize("m,n,k,epi", [(128, 37888, 3584, "swiglu"), (100, 33000, 1024, "none"), (40, 37888, 128, "swiglu"),
                                       (64, 40960, 64, "none"), (65, 20512, 3584, "swiglu")])
def test_gemm_wide_tiles_race_screen(gpu, m, n, k, epi):
    """At most 128 rows x tens of thousands of columns (gate/up at decode batch 33-128): the ring kernel's 64x160 / 128x160 tiles
    (ten n tiles per wave: the shared epilogue on two groups of four + a single tile pair; four stages) against the other kernels, 8
    times: bit-identical; ragged M / N (N not a multiple of 160), one and two K-tiles (fewer than stages)."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 95 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 96, 0.05, device=gpu)
    b = bf16_randn((n,), 97, device=gpu)
    E = {"none": _lib.EPI_NONE, "swiglu": _lib.EPI_SWIGLU}[epi]

    def run():
        return ops.gemm_bf16(a, w, None if epi == "swiglu" else b, epilogue=E)

    lib.owc_tuning_set(b"gemm_skinny_max_m", 0)
    try:
        assert lib.owc_tuning_set(b"gemm_wide_tiles", 0) == 0
        want = run()
        assert lib.owc_tuning_set(b"gemm_wide_tiles", 1) == 0
        for i in range(8):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_wide_tiles", 1)
        lib.owc_tuning_set(b"gemm_skinny_max_m", SKINNY_MAX_M)


@pytest.mark.parametrize("m,n,k,epi", [(512, 37888, 3584, "swiglu"),     # the 7B gate/up of a 512-row decode step: 296 tiles -> 256 + 80 half tiles
                                       (1024, 37888, 3584, "swiglu"),    # 592 -> 512 + 160
                                       (1792, 37888, 512, "swiglu"),     # 7 tile rows: 1036 -> 1022 + 28
                                       (500, 37888, 256, "swiglu"),      # ragged M (clamped rows in both kernels), two K-tiles
                                       (768, 25000, 384, "none"),        # 3 x 98 = 294 tiles -> 255 + 39 whole tiles' worth; ragged N in the second launch
                                       (512, 33024, 384, "none"),        # 258 tiles -> 256 + one tile column
                                       (512, 45056, 384, "none")])       # 352 tiles: the rest (96) is more than half a round -> no split (same path both ways)
def test_gemm_short_last_round_split_does_not_change_a_bit(gpu, m, n, k, epi):
    """Round 6: when the 256x256 tiles beyond the whole rounds of 256 are at most half a round, `launch` gives the tile columns of the
    whole rounds to `gemm_bf16_nt_256pp_kernel` and the rest of N to one round of `gemm_bf16_nt_256x128pp_kernel` (knob
    "gemm_tail_split", 0 = one launch as in rounds 2-5).  A column split leaves every output element's K chain alone: outputs must
    be equal bit for bit, with a canary in the output buffer (a column range that neither launch writes would keep it), and 8 launches
    per shape (the second kernel starts while the first one's last tiles drain)."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 170 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 171, 0.05, device=gpu)
    E = {"none": _lib.EPI_NONE, "swiglu": _lib.EPI_SWIGLU}[epi]
    cols = n // 2 if epi == "swiglu" else n

    def run():
        out = torch.full((m, cols), 777.0, dtype=torch.bfloat16, device=gpu)
        return ops.gemm_bf16(a, w, None, epilogue=E, out=out)

    try:
        assert lib.owc_tuning_set(b"gemm_tail_split", 0) == 0
        want = run()
        assert lib.owc_tuning_set(b"gemm_tail_split", 1) == 0
        for i in range(8):
            got = run()
            assert torch.equal(got, want), (i, (got != want).sum().item())
    finally:
        lib.owc_tuning_set(b"gemm_tail_split", -1)
    if epi == "none":
        assert_bf16_close(to_np(want[:128]), _oracle(a[:128], w, None), atol=1e-4)


@pytest.mark.parametrize("m,n,k,epi", [(4096, 4096, 4096, "none"), (2048, 37888, 3584, "swiglu"), (32768, 1280, 1280, "residual"),
                                       (1100, 13000, 384, "none"), (65536, 1280, 256, "quick_gelu"), (3000, 5120, 5120, "f32")])
def test_pingpong_kernel_bit_identical_to_lockstep_race_screen(gpu, m, n, k, epi):
    """The ping-pong 256x256 kernel (two waves per SIMD alternate MFMA / load roles; counted vmcnt + one-phase-later reads are
    its only ordering) against the lock-step kernel on the same operands: every output element is the same ascending MFMA
    chain, so the results must be BIT-identical - repeated 25 times per shape (a mis-ordered LDS-DMA / ds_read shows up as
    rare wrong tiles, not as a tolerance drift), on ragged M / N edges, the shortest legal K (2 K-tiles: prologue + peeled tail
    only), 4, 6 and 80 K-tiles, and every epilogue family."""
    from lmms_owc_amd import _lib, ops

    lib = _lib.load()
    a = bf16_randn((m, k), 70 + (m % 97), device=gpu)
    w = bf16_randn((n, k), 71, 0.05, device=gpu)
    b = bf16_randn((n,), 72, device=gpu)
    r = bf16_randn((m, n), 73, device=gpu) if epi == "residual" else None
    code = {"none": _lib.EPI_NONE, "swiglu": _lib.EPI_SWIGLU, "residual": _lib.EPI_RESIDUAL, "quick_gelu": _lib.EPI_QUICK_GELU,
            "f32": _lib.EPI_F32}[epi]
    run = lambda: ops.gemm_bf16(a, w, None if epi == "swiglu" else b, epilogue=code, residual=r)  # noqa: E731
    try:
        assert lib.owc_tuning_set(b"gemm_pingpong", 0) == 0
        ref = run()
        assert lib.owc_tuning_set(b"gemm_pingpong", 1) == 0
        bad = 0
        for _ in range(25):
            bad += int(not torch.equal(run(), ref))
    finally:
        lib.owc_tuning_set(b"gemm_pingpong", -1)   # (negative: the defaults of both dtypes)
    assert bad == 0, f"{bad} / 25 launches differ from the lock-step kernel"
    if epi == "none":
        assert_bf16_close(to_np(ref[:256]), _oracle(a[:256], w, b), atol=1e-4)


@pytest.mark.parametrize("name,n,k,epi", [("qkv", 4608, 3584, "bias"), ("o", 3584, 3584, "residual"), ("gateup", 37888, 3584, "swiglu"),
                                          ("down", 3584, 18944, "residual"), ("vit.fc1", 5120, 1280, "quick_gelu"), ("vit.patch", 1280, 1176, "none")])
def test_row0_bits_do_not_depend_on_m(gpu, name, n, k, epi):
    """Batch invariance at the op: row 0 of a projection of the 7B decoder / vision tower must be the same bits whether 1 or 18 304 rows
    ride along - i.e. whichever of the weight-streaming, 64x64-ring, 128x128 and 256x256 ping-pong kernels (with or without K tail) the
    dispatch picks at that M: every threshold of `launch()` is crossed here (32 | 33, 64 | 65, 255 | 256 with the padding rule at 384,
    1023 | 1024, 144 tiles of 256x256)."""
    from lmms_owc_amd import _lib, ops

    ms = [1, 8, 32, 33, 64, 65, 128, 255, 256, 286, 384, 512, 1023, 1024, 2048, 18304]
    a = bf16_randn((max(ms), k), 90, device=gpu)
    w = bf16_randn((n, k), 91, 0.05, device=gpu)
    b = bf16_randn((n,), 92, device=gpu)
    r = bf16_randn((max(ms), n), 93, device=gpu)
    E = {"bias": _lib.EPI_NONE, "none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU, "quick_gelu": _lib.EPI_QUICK_GELU}[epi]
    ref = None
    for m in ms:
        out = ops.gemm_bf16(a[:m], w, None if epi in ("none", "swiglu") else b, epilogue=E, residual=r[:m] if epi == "residual" else None)
        row = out[0].clone()
        if ref is None:
            ref = row
        assert torch.equal(row, ref), (name, m, int((row != ref).sum()))
