"""Pin the numpy e4m3fn codec of oracle/fp8_np.py against an independent implementation: torch's CPU cast to
torch.float8_e4m3fn (round to nearest even), on random values, every representable value, the rounding midpoints and the
saturation edge; and check the quantiser's algebra."""
import numpy as np
import torch

from oracle import fp8_np as F


def test_e4m3_encode_matches_torch_cast():
    r = np.random.default_rng(0)
    finite = F.E4M3_DECODE[~np.isnan(F.E4M3_DECODE)]
    pos = np.sort(np.unique(np.abs(finite)))
    mids = ((pos[:-1] + pos[1:]) / 2).astype(np.float32)          # exact ties: must go to the even code
    x = np.concatenate([r.standard_normal(100000).astype(np.float32) * 60, np.linspace(-448, 448, 40001, dtype=np.float32), finite,
                        mids, -mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                        np.array([0.0, -0.0, 2.0 ** -10, 2.0 ** -9, 1e-30, 447.99, 448.0], np.float32)])
    got = F.e4m3_encode(x)
    want = torch.from_numpy(x).to(torch.float8_e4m3fn).view(torch.uint8).numpy()
    zero = ((got & 0x7F) == 0) & ((want & 0x7F) == 0)
    assert np.all((got == want) | zero)
    assert np.array_equal(F.e4m3_decode(want[~zero]), torch.from_numpy(want[~zero]).view(torch.float8_e4m3fn).float().numpy())
    # saturation (torch's cast does not saturate, the quantiser must)
    assert F.e4m3_encode(np.array([449.0, 1e9, -5000.0], np.float32)).tolist() == [0x7E, 0x7E, 0xFE]


def test_quantize_rows_algebra():
    r = np.random.default_rng(1)
    x = (r.standard_normal((7, 64)) * 3).astype(np.float32)
    x[3] = 0
    q, s = F.quantize_rows(x)
    assert s[3] == 1.0 and not q[3].any()
    assert np.allclose(s[[0, 1, 2, 4, 5, 6]], np.abs(x[[0, 1, 2, 4, 5, 6]]).max(1) / 448.0)
    assert (F.e4m3_decode(q).max(1)[[0, 1]] <= 448).all() and np.abs(F.e4m3_decode(q)).max() == 448.0
    err = np.abs(F.e4m3_decode(q) * s[:, None] - x)
    assert (err <= np.abs(x) * 2.0 ** -4 + s[:, None] * 2.0 ** -10).all()   # half an ulp of a 3-bit mantissa (+ subnormal step)
    y = F.linear_fp8(x, *F.quantize_rows((r.standard_normal((5, 64)) * 0.1).astype(np.float32)), bf16=False)
    assert y.shape == (7, 5) and np.isfinite(y).all()


def test_fast_encoder_equals_the_table_search_definition():
    """e4m3_encode (bit arithmetic, threaded for big inputs) == e4m3_encode_search (nearest table entry, ties to even)."""
    r = np.random.default_rng(5)
    finite = F.E4M3_DECODE[~np.isnan(F.E4M3_DECODE)]
    pos = np.sort(np.unique(np.abs(finite)))
    mids = ((pos[:-1] + pos[1:]) / 2).astype(np.float32)
    x = np.concatenate([r.standard_normal(300000).astype(np.float32) * r.choice([1e-3, 0.02, 1.0, 100.0, 1000.0], 300000).astype(np.float32),
                        finite, mids, -mids, np.nextafter(mids, np.float32(0)), np.nextafter(mids, np.float32(1e9)),
                        np.array([0.0, -0.0, 1e-30, 2.0 ** -10, 2.0 ** -9, 2.0 ** -6, 447.99, 448.0, 449.0, 1e9, -5000.0], np.float32)])
    assert np.array_equal(F.e4m3_encode(x), F.e4m3_encode_search(x))
    big = (r.standard_normal((1 << 22) + 12345).astype(np.float32) * 50)          # threaded path
    assert np.array_equal(F.e4m3_encode(big), F.e4m3_encode_search(big))
