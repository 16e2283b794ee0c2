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


def test_repetition_penalty_against_hf_generate_golden():
    """tests/golden/qwen2vl_tiny_rep.npz (tools/gen_golden.py qwen_rep): HF's Qwen2VLForConditionalGeneration.generate of the tiny model
    with `repetition_penalty = 1.3` on the model's GENERATION CONFIG and the reference's argument list (no repetition_penalty
    argument, /root/reference/src/models/_qwen2_vl.py:319-329).  (1) HF applied it: its processed scores differ from its raw logits
    and the tokens differ from the run with the penalty switched off (prompts chosen so that the seeded model loops without it);
    (2) the oracle's restatement maps HF's raw logits onto HF's processed scores EXACTLY, history = prompt ids + tokens so far;
    (3) the oracle's own generate() with the penalty reproduces HF's fp32 tokens wherever HF's processed top-2 margin is decisive."""
    import json

    from oracle import qwen2vl_np as Q

    g = np.load(GOLD / "qwen2vl_tiny_rep.npz")
    meta = json.loads((GOLD / "qwen2vl_tiny_rep.json").read_text())
    p = meta["repetition_penalty"]
    flipped = 0
    for name, case in meta["cases"].items():
        ids = g[f"{name}_ids"]
        for tag in ("f32", "bf16"):
            toks, raw, proc = g[f"{name}_{tag}_tokens"], g[f"{name}_{tag}_logits"], g[f"{name}_{tag}_scores"]
            hist = [int(t) for t in ids]
            for j in range(len(toks)):
                want = Q.repetition_penalty_scores(raw[j], hist, p)
                assert np.array_equal(want, proc[j]), (name, tag, j, np.abs(want - proc[j]).max())
                assert int(np.argmax(proc[j])) == int(toks[j])
                hist.append(int(toks[j]))
            assert not np.array_equal(raw, proc)
        flipped += int(not np.array_equal(g[f"{name}_f32_tokens"], g[f"{name}_f32_tokens_without_penalty"]))
    assert flipped >= 3
    # (3) the oracle end to end (fp32), one case: same tokens as HF up to the first step whose processed margin is within noise
    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, meta["weights_seed"])
    name = "s16"
    grid = [tuple(r) for r in meta["cases"][name]["grid"]]
    pix = recipes.pixel_values(grid, 7)
    toks = Q.generate(w, cfg, g[f"{name}_ids"], pix, grid, 12, repetition_penalty=p)
    proc = g[f"{name}_f32_scores"]
    sure = [np.sort(proc[j])[-1] - np.sort(proc[j])[-2] > 1e-3 * np.abs(proc[j]).max() for j in range(12)]
    n_sure = sure.index(False) if False in sure else 12
    assert n_sure >= 6 and np.array_equal(toks[:n_sure], g[f"{name}_f32_tokens"][:n_sure]), (toks, g[f"{name}_f32_tokens"])
    assert not np.array_equal(toks, g[f"{name}_f32_tokens_without_penalty"])
