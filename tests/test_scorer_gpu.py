"""Scorer parity on the GPU: fp32 HIP BERT encoder + cosine kernels vs the oracle and the golden
vectors the REFERENCE's encode_sentence_bert / semantic_similarity produced (tests/golden/scorer.npz).
Tolerance: 1e-4 on cosine (BASELINE.json north_star), we assert 2e-5."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import bert_np as B
from tests import recipes
from tests.util import to_np

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).parent / "golden"


@pytest.mark.parametrize("kind,n,L", [("minilm", 8, 16)])
def test_embed_matches_reference_golden(gpu, kind, n, L):
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    g = np.load(GOLD / "scorer.npz")
    c = recipes.bert_cfg(kind)
    sc = SentenceScorer(BertWeights(c, recipes.bert_weights(c, 1234), gpu))
    ids_r, mask_r = recipes.label_tokens(n, L, c["vocab_size"], seed=21)
    ids_p, mask_p = recipes.label_tokens(n, L, c["vocab_size"], seed=22)
    zr, zp = sc.embed(ids_r, mask_r), sc.embed(ids_p, mask_p)
    np.testing.assert_allclose(to_np(zr), g[f"{kind}_ref_embeds"], atol=2e-5)
    np.testing.assert_allclose(to_np(zp), g[f"{kind}_pred_embeds"], atol=2e-5)
    cos = to_np(sc.paired_cosine(zr, zp))
    np.testing.assert_allclose(cos, g[f"{kind}_semantic_similarity_none"], atol=2e-5)


def test_ragged_256_labels_match_reference_golden(gpu):
    """256 ragged labels (2..16 tokens) at full MiniLM-L6 size against the REFERENCE's own CPU-fp32 output
    (tests/golden/scorer_minilm256.npz): packed rows, a batch split in the middle (max_batch 100), paired cosine, the mean and
    the threshold masses of mean_average_semantic_similarity (/root/reference/src/data/metrics/_group.py:392-458, :537-544)."""
    import json

    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    g = np.load(GOLD / "scorer_minilm256.npz")
    meta = json.loads((GOLD / "scorer_minilm256.json").read_text())
    c = recipes.bert_cfg("minilm")
    s_r, s_p = (int(x) for x in g["label_seeds"])
    ids_r, mask_r = recipes.label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=s_r)
    ids_p, mask_p = recipes.label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=s_p)
    for max_batch in (100, 16384):
        sc = SentenceScorer(BertWeights(c, recipes.bert_weights(c, meta["weights_seed"]), gpu), max_batch=max_batch)
        zr, zp = sc.embed(ids_r, mask_r), sc.embed(ids_p, mask_p)
        np.testing.assert_allclose(to_np(zr), g["ref_embeds"], atol=2e-5)
        np.testing.assert_allclose(to_np(zp), g["pred_embeds"], atol=2e-5)
        cos = to_np(sc.paired_cosine(zr, zp))
        np.testing.assert_allclose(cos, g["semantic_similarity_none"], atol=2e-5)
        assert abs(float(cos.mean()) - float(g["semantic_similarity_mean"])) <= 2e-5
        got = B.mean_average(cos)   # threshold masses: a pure function of the cosines (the oracle's helper as the checker)
        for k, v in meta["mean_average"].items():
            assert abs(got[k] - v) <= 1.0 / meta["n"] + 1e-6, k   # a cosine within 2e-5 of a threshold may land on either side


def test_bf16x3_linears_match_the_exact_f32_mfma(gpu):
    """The encoder's linears run as three-piece bf16 splits of the fp32 operands (6 bf16 MFMAs per block, gemm_f32.hip); the
    exact f32-input MFMA kernel stays selectable.  Both must sit within the golden tolerance, and within 2e-6 of each other."""
    from lmms_owc_amd import _lib
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    g = np.load(GOLD / "scorer.npz")
    c = recipes.bert_cfg("minilm")
    sc = SentenceScorer(BertWeights(c, recipes.bert_weights(c, 1234), gpu))
    ids, mask = recipes.label_tokens(8, 16, c["vocab_size"], seed=21)
    lib = _lib.load()
    try:
        assert lib.owc_tuning_set(b"bert_bf16x3", 0) == 0
        exact = to_np(sc.embed(ids, mask))
        assert lib.owc_tuning_set(b"bert_bf16x3", 1) == 0
        split = to_np(sc.embed(ids, mask))
    finally:
        lib.owc_tuning_set(b"bert_bf16x3", 1)
    np.testing.assert_allclose(exact, g["minilm_ref_embeds"], atol=2e-5)
    np.testing.assert_allclose(split, g["minilm_ref_embeds"], atol=2e-5)
    np.testing.assert_allclose(split, exact, atol=2e-6)
    assert not np.array_equal(split, exact)  # the knob really switches kernels


def test_embed_matches_oracle_ragged(gpu):
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    c = recipes.bert_cfg("minilm")
    w = recipes.bert_weights(c, 77)
    sc = SentenceScorer(BertWeights(c, w, gpu), max_batch=100)
    for n, L in ((1, 2), (257, 9), (64, 33)):
        ids, mask = recipes.label_tokens(n, L, c["vocab_size"], seed=n)
        np.testing.assert_allclose(to_np(sc.embed(ids, mask)), B.sentence_embed(w, c, ids, mask), atol=2e-5)


@pytest.mark.parametrize("N,C,k", [(1, 3, 1), (100, 37, 5), (513, 397, 5), (64, 1000, 16), (300, 10450, 5)])  # last: config #5's ~10k-class set
def test_cosine_topk(gpu, N, C, k):
    from lmms_owc_amd.engine.scorer import SentenceScorer

    r = np.random.default_rng(N)
    z = r.standard_normal((N, 384)).astype(np.float32)
    z /= np.linalg.norm(z, axis=-1, keepdims=True)
    cl = r.standard_normal((C, 384)).astype(np.float32)
    cl /= np.linalg.norm(cl, axis=-1, keepdims=True)
    cl[0] = z[0]  # an exact hit
    label = r.integers(0, C, N)
    kk = min(k, C)
    tv, ti, paired = SentenceScorer.topk(torch.from_numpy(z).to(gpu), torch.from_numpy(cl).to(gpu), kk,
                                         torch.from_numpy(label.astype(np.int32)).to(gpu))
    wv, wi = B.cosine_topk(z, cl, kk)
    np.testing.assert_allclose(to_np(tv), wv, atol=2e-6)
    sim = z @ cl.T
    # indices must agree except where two classes tie within fp32 rounding
    got_i = to_np(ti).astype(int)
    np.testing.assert_allclose(np.take_along_axis(sim, got_i, 1), wv, atol=2e-6)
    assert (got_i == wi).mean() > 0.999
    np.testing.assert_allclose(to_np(paired), sim[np.arange(N), label], atol=2e-6)
    assert got_i[0, 0] == 0 and abs(to_np(tv)[0, 0] - 1.0) < 1e-5


def test_packed_rows_equal_padded_rows(gpu):
    """`owc_bert_embed_packed` (only mask == 1 tokens are rows) against `owc_bert_embed` (every padded position computed, the
    reference's shape of the computation, _text.py:193-198): a sequence's arithmetic is the same chain of operations in both (the
    two attention kernels differ only in how the compiler contracts their multiply-adds), so the embeddings agree to fp32
    rounding (<= 1e-6 on unit vectors) - ragged lengths 2..16, a batch split (max_batch), and masks with HOLES (position ids
    stay the padded columns); both also sit within 2e-5 of the numpy oracle."""
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    c = recipes.bert_cfg("minilm")
    w = recipes.bert_weights(c, 31)
    sc = SentenceScorer(BertWeights(c, w, gpu), max_batch=300)
    ids, mask = recipes.label_tokens(777, 16, c["vocab_size"], seed=5)
    a, b = sc.embed(ids, mask, packed=True), sc.embed(ids, mask, packed=False)
    assert float((a - b).abs().max()) <= 1e-6
    r = np.random.default_rng(9)
    holes = (r.random(mask.shape) < 0.8).astype(np.int64) * mask
    holes[:, 0] = 1
    a, b = sc.embed(ids, holes, packed=True), sc.embed(ids, holes, packed=False)
    assert float((a - b).abs().max()) <= 1e-6
    np.testing.assert_allclose(to_np(a[:64]), B.sentence_embed(w, c, ids[:64], holes[:64]), atol=2e-5)
    # device tensors are accepted as well (the layout is built from a host copy of the mask)
    d = sc.embed(torch.from_numpy(ids).to(gpu), torch.from_numpy(holes).to(gpu))
    assert torch.equal(d, a)


@pytest.mark.parametrize("kind", ["tiny", "base"])
def test_mpnet_embed_matches_reference_golden(gpu, kind):
    """all-mpnet-base-v2 (BASELINE.json configs[0]'s encoder) through owc_bert_embed / owc_bert_embed_packed: head_dim 64, position ids
    = column + 2, no token types, the relative-position bias table expanded over the key - query offsets at load.  Against the
    REFERENCE's encode_sentence_bert around HF's MPNetModel (tests/golden/scorer_mpnet.npz: tiny and the full 12 x 768 size), packed
    and padded rows, a batch split in the middle; <= 2e-5 like the MiniLM encoder."""
    import json

    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    g = np.load(GOLD / "scorer_mpnet.npz")
    meta = json.loads((GOLD / "scorer_mpnet.json").read_text())[kind]
    c = recipes.mpnet_cfg(kind)
    ids, mask = recipes.mpnet_label_tokens(meta["n"], meta["L"], c["vocab_size"], seed=meta["label_seed"])
    w = BertWeights(c, recipes.mpnet_weights(c, meta["weights_seed"]), gpu)
    assert w.mpnet and w.c.pos_offset == 2 and w.c.rel_span == c["max_position_embeddings"] - 2
    for max_batch in (10, 16384):
        sc = SentenceScorer(w, max_batch=max_batch)
        np.testing.assert_allclose(to_np(sc.embed(ids, mask)), g[f"{kind}_embeds"], atol=2e-5)
    np.testing.assert_allclose(to_np(sc.embed(ids, mask, packed=False)), g[f"{kind}_embeds"], atol=2e-5)


@pytest.mark.parametrize("L", [300, 384, 512])
def test_long_sequences_through_the_chunked_attention(gpu, L):
    """Round 6: the sentence encoder's attention kept K and V of a whole sequence in LDS - head_dim 64 (all-mpnet-base-v2,
    BASELINE.json configs[0]'s encoder) stopped at 317 tokens although the tokenizer truncates at 512 and the reference's
    `encode_sentence_bert` (src/data/pipelines/text/_text.py:175-202) simply encodes such a prediction.  The kernels now hold 256
    keys at a time and carry the online softmax across chunks in the same key order (bit-identical to the one-chunk form).  Base width
    (768 / 12 heads / 3072), 2 layers, sequences of 2..L tokens, packed and padded rows, against the numpy oracle; and MiniLM
    (head_dim 32) at the same lengths."""
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer

    c = dict(recipes.mpnet_cfg("base"), num_hidden_layers=2)
    ids, mask = recipes.mpnet_label_tokens(5, L, c["vocab_size"], seed=L)
    w = recipes.mpnet_weights(c, 31)
    want = B.sentence_embed(w, c, ids, mask)
    sc = SentenceScorer(BertWeights(c, w, gpu), max_batch=64)
    np.testing.assert_allclose(to_np(sc.embed(ids, mask)), want, atol=2e-5)
    np.testing.assert_allclose(to_np(sc.embed(ids, mask, packed=False)), want, atol=2e-5)
    c2 = recipes.bert_cfg("minilm")
    c2 = dict(c2, num_hidden_layers=2, max_position_embeddings=max(c2.get("max_position_embeddings", 512), 512))
    w2 = recipes.bert_weights(c2, 78)
    ids2, mask2 = recipes.label_tokens(4, L, c2["vocab_size"], seed=L + 1)
    sc2 = SentenceScorer(BertWeights(c2, w2, gpu), max_batch=64)
    np.testing.assert_allclose(to_np(sc2.embed(ids2, mask2)), B.sentence_embed(w2, c2, ids2, mask2), atol=2e-5)
