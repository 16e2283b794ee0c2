"""Pin the LLaVA numpy oracle (oracle/llava_np.py) against golden vectors produced by HF transformers'
LlavaForConditionalGeneration (tools/gen_golden.py gen_llava) — the model class the reference drives in
`src/models/_llava_hf.py:365-376`."""

import numpy as np

from oracle import llava_np as L
from tests import recipes
from pathlib import Path

GOLDEN = Path(__file__).parent / "golden"


def _setup():
    g = np.load(GOLDEN / "llava_tiny.npz")
    cfg = recipes.tiny_llava_cfg()
    return g, cfg, recipes.llava_weights(cfg, 1234), recipes.clip_pixels(2, cfg.vision.image_size)


def test_clip_features_fp32_match_hf():
    g, cfg, w, pix = _setup()
    feats = L.project(w, L.clip_features(w, cfg, pix), bf16=False).reshape(-1, cfg.text.hidden_size)
    np.testing.assert_allclose(feats, g["f32_feats"], rtol=2e-4, atol=2e-4)


def test_generate_fp32_matches_hf():
    g, cfg, w, pix = _setup()
    toks, logits = L.generate(w, cfg, g["ids"], pix, 8, return_logits=True)
    assert toks.tolist() == g["f32_tokens"].tolist()
    np.testing.assert_allclose(logits, g["f32_logits"], rtol=3e-4, atol=3e-4)


def test_generate_bf16_matches_hf():
    g, cfg, w, pix = _setup()
    feats = L.project(w, L.clip_features(w, cfg, pix, bf16=True), bf16=True).reshape(-1, cfg.text.hidden_size)
    # bf16: same rounding points, different fp32 accumulation order -> a few ulps on isolated elements
    err = np.abs(feats - g["bf16_feats"])
    assert np.quantile(err, 0.99) <= 2.0 ** -6 * np.abs(g["bf16_feats"]).max()
    toks, logits = L.generate(w, cfg, g["ids"], pix, 8, bf16=True, return_logits=True)
    assert toks.tolist() == g["bf16_tokens"].tolist()
    assert np.abs(logits - g["bf16_logits"]).max() <= 0.05 * np.abs(g["bf16_logits"]).max()


# ---------------------------------------------------------------- LLaVA-NeXT (anyres)
def _setup_next():
    g = np.load(GOLDEN / "llava_next_tiny.npz")
    cfg = recipes.tiny_llava_next_cfg()
    return g, cfg, recipes.llava_weights(cfg, 1234), recipes.clip_pixels(int(g["views"].sum()), cfg.vision.image_size, seed=41)


def test_next_generate_fp32_matches_hf():
    g, cfg, w, pix = _setup_next()
    toks, logits = L.generate(w, cfg, g["ids"], pix, 8, return_logits=True, image_sizes=g["image_sizes"].tolist(),
                              views_per_image=g["views"].tolist())
    assert toks.tolist() == g["f32_tokens"].tolist()
    np.testing.assert_allclose(logits, g["f32_logits"], rtol=3e-4, atol=3e-4)


def test_next_packed_features_match_hf():
    g, cfg, w, pix = _setup_next()
    feats = L.project(w, L.clip_features(w, cfg, pix), bf16=False)
    packed, v0 = [], 0
    for nv, size in zip(g["views"].tolist(), g["image_sizes"].tolist()):
        packed.append(L.pack_anyres(w, cfg, feats[v0:v0 + nv], size))
        v0 += nv
    assert [len(p) for p in packed] == g["n_tok"].tolist()
    np.testing.assert_allclose(np.concatenate(packed), g["f32_feats"], rtol=2e-4, atol=2e-4)


def test_next_generate_bf16_matches_hf():
    g, cfg, w, pix = _setup_next()
    toks, logits = L.generate(w, cfg, g["ids"], pix, 8, bf16=True, return_logits=True, image_sizes=g["image_sizes"].tolist(),
                              views_per_image=g["views"].tolist())
    assert toks.tolist() == g["bf16_tokens"].tolist()
    assert np.abs(logits - g["bf16_logits"]).max() <= 0.05 * np.abs(g["bf16_logits"]).max()


def test_loglikelihood_matches_hf_with_the_references_label_mask():
    """The reference's LLaVA.loglikelihood (src/models/_llava_hf.py:229-252) on HF's own model (tools/gen_golden.py
    gen_llava_loglik): the loss over every position from the UN-expanded prompt length on (image positions included) and the
    unshifted greedy comparison."""
    g = np.load(GOLDEN / "llava_loglik_tiny.npz")
    cfg = recipes.tiny_llava_cfg()
    w = recipes.llava_weights(cfg, 1234)
    n_ctx = int(g["n_ctx"])
    loss, eq, logits = L.loglikelihood(w, cfg, g["ids"], g["pix"], n_ctx, return_logits=True)
    assert abs(loss - float(g["f32_loss"])) <= 1e-5 and eq == bool(g["f32_max_equal"])
    np.testing.assert_allclose(logits, g["f32_logits"], rtol=3e-4, atol=3e-4)
    assert np.argmax(logits[1:], -1).tolist() == g["f32_greedy"].tolist()
    loss16, eq16 = L.loglikelihood(w, cfg, g["ids"], g["pix"], n_ctx, bf16=True)
    assert abs(loss16 - float(g["bf16_loss"])) <= 2e-3 * float(g["bf16_loss"]) and eq16 == bool(g["bf16_max_equal"])
    # the flag's True branch, text only (with an image the scored range holds image-token positions, whose greedy token would have
    # to be the image token itself): iterate ids[n:] <- argmax(logits[n:]) to a fixed point of the unshifted comparison
    ids = np.random.default_rng(3).integers(1, 400, 14)
    for _ in range(12):
        _, eq, lg = L.loglikelihood(w, cfg, ids, None, 8, return_logits=True)
        new = np.argmax(lg[1:], -1)
        if np.array_equal(new, ids[8:]):
            assert eq is True
            break
        assert eq is False
        ids[8:] = new
