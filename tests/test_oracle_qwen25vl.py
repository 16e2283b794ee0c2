"""Pin the numpy Qwen2.5-VL oracle (oracle/qwen25vl_np.py: window attention, RMSNorm + gated-MLP vision blocks) against golden
vectors produced by HF's Qwen2_5_VLForConditionalGeneration (tools/gen_golden.py qwen25) - the model class the reference loads for
`qwen2.5-vl-*` (/root/reference/src/models/_qwen2_vl.py:106-115).  fp32 tight, bf16 loose (same rounding points, other summation order)."""
import json
import zlib
from pathlib import Path

import numpy as np
import pytest

from oracle import qwen2vl_np as Q
from oracle import qwen25vl_np as Q25
from tests import recipes

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "qwen25vl_tiny.npz"), json.loads((GOLD / "qwen25vl_tiny.json").read_text())


@pytest.fixture(scope="module")
def model():
    cfg = recipes.tiny_cfg25()
    return cfg, recipes.qwen25vl_weights(cfg, 1234)


def test_window_index_at_real_geometry(gold):
    """window_index / cu_window_seqlens of HF's get_vision_window_index for the 448 x 448 bench image and ragged sizes (sides that
    are a window multiple, smaller than a window, 5 : 1 aspect): integers, exact."""
    _, meta = gold
    for name, geo in meta["window_geometry"].items():
        wi, cu = Q25.vision_window_index(geo["grid"], 2, 112, 14)
        assert len(wi) == geo["n"] and zlib.crc32(wi.astype(np.int64).tobytes()) == geo["window_index_crc"], name
        assert cu.tolist() == geo["cu_window_seqlens"], name
        assert sorted(wi.tolist()) == list(range(geo["n"]))   # a permutation of the merged groups


@pytest.mark.parametrize("case", ["a", "b"])
def test_rope_index_matches_hf(gold, model, case):
    g, _ = gold
    cfg, _ = model
    pos, delta = Q.rope_index(g[f"{case}_ids"], g[f"{case}_grid"], cfg)   # images: identical to Qwen2-VL's
    assert np.array_equal(pos, g[f"{case}_pos3"]) and delta == int(g[f"{case}_delta"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_vit_fp32_matches_hf(gold, model, case):
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    out = Q25.vit_forward(w, cfg, recipes.pixel_values(grid, 7), grid, bf16=False)
    np.testing.assert_allclose(out, g[f"{case}_f32_vit"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["a", "b"])
def test_generate_fp32_matches_hf(gold, model, case):
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    toks, logits = Q25.generate(w, cfg, g[f"{case}_ids"], recipes.pixel_values(grid, 7), grid, 8, return_logits=True)
    np.testing.assert_allclose(logits, g[f"{case}_f32_logits"], rtol=1e-3, atol=1e-3)
    assert np.array_equal(toks, g[f"{case}_f32_tokens"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_bf16_close_to_hf_bf16(gold, model, case):
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    pix = recipes.pixel_values(grid, 7)
    vit = Q25.vit_forward(w, cfg, pix, grid, bf16=True)
    ref = g[f"{case}_bf16_vit"]
    assert np.abs(vit - ref).max() <= 0.03 * np.abs(ref).max()
    forced = g[f"{case}_bf16_tokens"]
    _, logits = Q25.generate(w, cfg, g[f"{case}_ids"], pix, grid, 8, bf16=True, return_logits=True, forced_tokens=forced)
    ref = g[f"{case}_bf16_logits"]
    for j in range(8):   # teacher-forced on HF's continuation: every step comparable
        assert np.abs(logits[j] - ref[j]).max() <= 0.03 * np.abs(ref[j]).max(), j
