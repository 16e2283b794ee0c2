"""Pin the numpy oracle against golden vectors produced by HF transformers (tools/gen_golden.py).

fp32 oracle vs HF fp32: tight tolerance (same maths, different summation order).
bf16 oracle vs HF bf16 (CPU): loose tolerance (rounding points match, accumulation order does not).
"""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import qwen2vl_np as Q
from tests import recipes

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "qwen2vl_tiny.npz"), json.loads((GOLD / "qwen2vl_tiny.json").read_text())


@pytest.fixture(scope="module")
def model():
    cfg = recipes.tiny_cfg()
    return cfg, recipes.qwen2vl_weights(cfg, 1234)


@pytest.mark.parametrize("case", ["a", "b"])
def test_rope_index_matches_hf(gold, model, case):
    g, _ = gold
    cfg, _ = model
    pos, delta = Q.rope_index(g[f"{case}_ids"], g[f"{case}_grid"], cfg)
    assert np.array_equal(pos, g[f"{case}_pos3"])
    assert delta == int(g[f"{case}_delta"])


def test_rope_index_448_prompt(gold, model):
    g, _ = gold
    cfg, _ = model
    pos, delta = Q.rope_index(g["p448_ids"], np.array([[1, 32, 32]]), cfg)
    assert pos.shape == (3, 286)
    assert np.array_equal(pos, g["p448_pos3"]) and delta == int(g["p448_delta"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_vit_fp32_matches_hf(gold, model, case):
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    out = Q.vit_forward(w, cfg, recipes.pixel_values(grid, 7), grid, bf16=False)
    np.testing.assert_allclose(out, g[f"{case}_f32_vit"], rtol=2e-4, atol=2e-4)


@pytest.mark.parametrize("case", ["a", "b"])
def test_generate_fp32_matches_hf(gold, model, case):
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    toks, logits = Q.generate(w, cfg, g[f"{case}_ids"], recipes.pixel_values(grid, 7), grid, 8, return_logits=True)
    np.testing.assert_allclose(logits, g[f"{case}_f32_logits"], rtol=1e-3, atol=1e-3)
    assert np.array_equal(toks, g[f"{case}_f32_tokens"])


@pytest.mark.parametrize("case", ["a", "b"])
def test_generate_bf16_close_to_hf_bf16(gold, model, case):
    """bf16 restatement: same rounding points as torch.bfloat16 modules; tolerance = a few bf16 ulps of the logit scale."""
    g, _ = gold
    cfg, w = model
    grid = [tuple(r) for r in g[f"{case}_grid"].tolist()]
    vit = Q.vit_forward(w, cfg, recipes.pixel_values(grid, 7), grid, bf16=True)
    ref = g[f"{case}_bf16_vit"]
    assert np.abs(vit - ref).max() <= 0.03 * np.abs(ref).max()
    toks, logits = Q.generate(w, cfg, g[f"{case}_ids"], recipes.pixel_values(grid, 7), grid, 8, bf16=True, return_logits=True)
    ref = g[f"{case}_bf16_logits"]
    n = min(len(logits), len(ref))
    # compare step 0 everywhere (later steps may diverge after a near-tie flips a token)
    assert np.abs(logits[0] - ref[0]).max() <= 0.05 * np.abs(ref[0]).max()
    same = toks[:n] == g[f"{case}_bf16_tokens"][:n]
    assert same[0]
