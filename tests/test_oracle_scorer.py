"""Pin the numpy scorer oracle against golden vectors produced by the reference's own
encode_sentence_bert / semantic_similarity (tools/gen_golden.py, CPU fp32 branch)."""
import json
from pathlib import Path

import numpy as np
import pytest

from oracle import bert_np as B
from tests import recipes

GOLD = Path(__file__).parent / "golden"


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD / "scorer.npz"), json.loads((GOLD / "scorer.json").read_text())


@pytest.mark.parametrize("kind,n,L", [("tiny", 24, 12), ("minilm", 8, 16)])
def test_embed_and_paired_cosine(gold, kind, n, L):
    g, _ = gold
    c = recipes.bert_cfg(kind)
    w = recipes.bert_weights(c, 1234)
    ids_r, mask_r = recipes.label_tokens(n, L, c["vocab_size"], seed=21)
    ids_p, mask_p = recipes.label_tokens(n, L, c["vocab_size"], seed=22)
    zr = B.sentence_embed(w, c, ids_r, mask_r)
    zp = B.sentence_embed(w, c, ids_p, mask_p)
    np.testing.assert_allclose(zr, g[f"{kind}_ref_embeds"], atol=2e-6)
    np.testing.assert_allclose(zp, g[f"{kind}_pred_embeds"], atol=2e-6)
    cos = B.paired_cosine(zr, zp)
    np.testing.assert_allclose(cos, g[f"{kind}_semantic_similarity_none"], atol=2e-6)
    np.testing.assert_allclose(cos.mean(), g[f"{kind}_semantic_similarity_mean"], atol=2e-6)


def test_mean_average(gold):
    g, meta = gold
    got = B.mean_average(g["tiny_semantic_similarity_none"])
    for k, v in meta["tiny_mean_average"].items():
        assert abs(got[k] - v) < 1e-6


def test_topk_contains_paired():
    r = np.random.default_rng(0)
    z = r.standard_normal((20, 64)).astype(np.float32)
    z /= np.linalg.norm(z, axis=-1, keepdims=True)
    cl = z[:7]
    val, idx = B.cosine_topk(z, cl, 3)
    assert (idx[:7, 0] == np.arange(7)).all()
    np.testing.assert_allclose(val[:7, 0], 1.0, atol=1e-6)
    assert (np.diff(val, axis=1) <= 0).all()


def test_ragged_256_labels_at_minilm_size():
    """256 ragged labels (lengths 2..16) through the REFERENCE's encode_sentence_bert / semantic_similarity /
    mean_average_semantic_similarity at full MiniLM-L6 size (tools/gen_golden.py scorer_ragged)."""
    g = np.load(GOLD / "scorer_minilm256.npz")
    meta = json.loads((GOLD / "scorer_minilm256.json").read_text())
    c = recipes.bert_cfg("minilm")
    w = recipes.bert_weights(c, meta["weights_seed"])
    s_r, s_p = (int(x) for x in g["label_seeds"])
    ids_r, mask_r = recipes.label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=s_r)
    ids_p, mask_p = recipes.label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=s_p)
    assert len(set(mask_r.sum(1).tolist())) >= 12   # really ragged
    zr, zp = B.sentence_embed(w, c, ids_r, mask_r), B.sentence_embed(w, c, ids_p, mask_p)
    np.testing.assert_allclose(zr, g["ref_embeds"], atol=2e-6)
    np.testing.assert_allclose(zp, g["pred_embeds"], atol=2e-6)
    cos = B.paired_cosine(zr, zp)
    np.testing.assert_allclose(cos, g["semantic_similarity_none"], atol=2e-6)
    np.testing.assert_allclose(cos.mean(), g["semantic_similarity_mean"], atol=2e-6)
    got = B.mean_average(g["semantic_similarity_none"])
    for k, v in meta["mean_average"].items():
        assert abs(got[k] - v) < 1e-6


@pytest.mark.parametrize("kind", ["tiny", "base"])
def test_mpnet_embeddings_match_the_reference_hook_around_hf_mpnet(kind):
    """BASELINE.json configs[0] names all-mpnet-base-v2.  `bert_np.mpnet_forward` (relative-position bias buckets, position ids =
    column + 2, no token types) against the REFERENCE's encode_sentence_bert run around HF's MPNetModel on seeded weights
    (tools/gen_golden.py scorer_mpnet; tiny and full all-mpnet-base-v2 size, ragged labels padded with MPNet's pad id)."""
    g = np.load(GOLD / "scorer_mpnet.npz")
    meta = json.loads((GOLD / "scorer_mpnet.json").read_text())[kind]
    c = recipes.mpnet_cfg(kind)
    assert c == meta["cfg"]
    w = recipes.mpnet_weights(c, meta["weights_seed"])
    ids, mask = recipes.mpnet_label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=meta["label_seed"])
    np.testing.assert_allclose(B.mpnet_forward(w, c, ids[:2], mask[:2]), g[f"{kind}_hidden0"], atol=2e-5)
    np.testing.assert_allclose(B.sentence_embed(w, c, ids, mask), g[f"{kind}_embeds"], atol=2e-6)
    # the bucket function: one bucket per distance below 8, logarithmic bins up to 128, 16 buckets per direction
    b = B.mpnet_relative_bucket(np.arange(-200, 201))
    assert b[200] == 0 and list(b[201:209]) == [16 + i for i in range(1, 8)] + [24] and b[0] == 15 and b[-1] == 31
    assert list(b[192:200]) == [8, 7, 6, 5, 4, 3, 2, 1]
