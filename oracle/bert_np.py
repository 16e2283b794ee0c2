"""numpy restatement of the sentence-embedding scorer (TEST INFRASTRUCTURE ONLY).

Follows `encode_sentence_bert` (/root/reference/src/data/pipelines/text/_text.py:143-208) and the
paired cosine of `semantic_similarity` / `mean_average_semantic_similarity`
(/root/reference/src/data/metrics/_group.py:488-544, :392-458).  The BertModel arithmetic itself lives
in transformers (pinned 4.47.0, not vendored): restated from modeling_bert.py (BertEmbeddings,
BertSelfAttention, BertSelfOutput, BertIntermediate, BertOutput).  Pinned against golden vectors
produced by running the reference's own functions in the build container (tools/gen_golden.py).
"""

from __future__ import annotations

import math

import numpy as np

from . import np_ops as ops


def _ln(x, w, b, eps):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + eps) * w + b


def bert_forward(w: dict, c: dict, ids: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """BertModel(input_ids, attention_mask).last_hidden_state  -> [n, L, H] (float32 maths)."""
    n, L = ids.shape
    H, NH = c["hidden_size"], c["num_attention_heads"]
    hd = H // NH
    eps = c["layer_norm_eps"]
    x = (w["embeddings.word_embeddings.weight"][ids]
         + w["embeddings.token_type_embeddings.weight"][0][None, None]
         + w["embeddings.position_embeddings.weight"][:L][None]).astype(np.float32)
    x = _ln(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], eps)
    neg = np.where(mask[:, None, None, :] > 0, 0.0, np.finfo(np.float32).min).astype(np.float32)
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        def lin(t, name):
            return t @ w[p + name + ".weight"].T + w[p + name + ".bias"]
        q = lin(x, "attention.self.query").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        k = lin(x, "attention.self.key").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        v = lin(x, "attention.self.value").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(hd) + neg
        a = ops.softmax(s, -1) @ v
        a = a.transpose(0, 2, 1, 3).reshape(n, L, H)
        x = _ln(lin(a, "attention.output.dense") + x, w[p + "attention.output.LayerNorm.weight"],
                w[p + "attention.output.LayerNorm.bias"], eps)
        f = ops.gelu_erf(lin(x, "intermediate.dense"))
        x = _ln(lin(f, "output.dense") + x, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], eps)
    return x.astype(np.float32)


def mpnet_relative_bucket(rel: np.ndarray, num_buckets: int = 32, max_distance: int = 128) -> np.ndarray:
    """MPNetEncoder.relative_position_bucket (transformers/models/mpnet/modeling_mpnet.py): `rel` = key column - query column.
    Half of the buckets per direction; distances below 8 get a bucket each, larger ones logarithmic bins up to 128."""
    n = -np.asarray(rel, np.int64)
    half = num_buckets // 2
    ret = (n < 0).astype(np.int64) * half
    n = np.abs(n)
    max_exact = half // 2
    # float32 like torch's `.float()`: log(n / 8) / log(16) * 8, truncated (n = 0 takes the `n < max_exact` branch below)
    large = max_exact + (np.log(np.maximum(n, 1).astype(np.float32) / np.float32(max_exact)) / np.float32(math.log(max_distance / max_exact))
                         * np.float32(half - max_exact)).astype(np.int64)
    large = np.minimum(large, half - 1)
    return ret + np.where(n < max_exact, n, large)


def mpnet_forward(w: dict, c: dict, ids: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """MPNetModel(input_ids, attention_mask).last_hidden_state -> [n, L, H] (float32): what sentence-transformers/all-mpnet-base-v2
    (BASELINE.json configs[0]'s encoder) runs.  Against BERT: no token-type embedding; position ids = (cumsum of non-pad) + 1,
    i.e. column + 2 for a right-padded row (pad id 1); ONE learned relative-position bias [32 buckets, heads], shared by all
    layers, added to the scaled scores; q / k / v / o linears and post-LayerNorm blocks as in BERT."""
    n, L = ids.shape
    H, NH = c["hidden_size"], c["num_attention_heads"]
    hd = H // NH
    eps = c["layer_norm_eps"]
    pad = 1
    nonpad = (ids != pad).astype(np.int64)
    pos = np.cumsum(nonpad, 1) * nonpad + pad
    x = (w["embeddings.word_embeddings.weight"][ids] + w["embeddings.position_embeddings.weight"][pos]).astype(np.float32)
    x = _ln(x, w["embeddings.LayerNorm.weight"], w["embeddings.LayerNorm.bias"], eps)
    neg = np.where(mask[:, None, None, :] > 0, 0.0, np.finfo(np.float32).min).astype(np.float32)
    col = np.arange(L)
    bucket = mpnet_relative_bucket(col[None, :] - col[:, None])                         # [query, key]
    bias = w["encoder.relative_attention_bias.weight"][bucket].transpose(2, 0, 1)[None]  # [1, heads, L, L]
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        def lin(t, name):
            return t @ w[p + name + ".weight"].T + w[p + name + ".bias"]
        q = lin(x, "attention.attn.q").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        k = lin(x, "attention.attn.k").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        v = lin(x, "attention.attn.v").reshape(n, L, NH, hd).transpose(0, 2, 1, 3)
        s = (q @ k.transpose(0, 1, 3, 2)) / math.sqrt(hd) + bias + neg
        a = ops.softmax(s, -1) @ v
        a = a.transpose(0, 2, 1, 3).reshape(n, L, H)
        x = _ln(lin(a, "attention.attn.o") + x, w[p + "attention.LayerNorm.weight"], w[p + "attention.LayerNorm.bias"], eps)
        f = ops.gelu_erf(lin(x, "intermediate.dense"))
        x = _ln(lin(f, "output.dense") + x, w[p + "output.LayerNorm.weight"], w[p + "output.LayerNorm.bias"], eps)
    return x.astype(np.float32)


def sentence_embed(w: dict, c: dict, ids: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """_text.py:175-189 mean pooling (clamp 1e-9) + :202 L2 normalisation."""
    h = mpnet_forward(w, c, ids, mask) if "encoder.relative_attention_bias.weight" in w else bert_forward(w, c, ids, mask)
    m = mask[:, :, None].astype(np.float32)
    pooled = (h * m).sum(1) / np.maximum(m.sum(1), 1e-9)
    return (pooled / np.linalg.norm(pooled, axis=-1, keepdims=True)).astype(np.float32)


def paired_cosine(refs_z: np.ndarray, preds_z: np.ndarray) -> np.ndarray:
    """torch.bmm(refs[N,1,D], preds[N,D,1]).squeeze()  (_group.py:537-544)."""
    return np.einsum("nd,nd->n", refs_z.astype(np.float32), preds_z.astype(np.float32))


def mean_average(cos: np.ndarray) -> dict:
    """_group.py:444-449 thresholds 0.5..0.9 and their mean."""
    out = {f"semantic_similarity@{t}": float((cos >= t).astype(np.float32).mean()) for t in (0.5, 0.6, 0.7, 0.8, 0.9)}
    out["semantic_similarity@avg"] = float(np.mean(np.array(list(out.values()), dtype=np.float32)))
    return out


def cosine_topk(preds_z: np.ndarray, classes_z: np.ndarray, k: int):
    """All-classes cosine + top-k (descending, lowest index on ties) — superset of the paired value."""
    sim = preds_z.astype(np.float32) @ classes_z.astype(np.float32).T
    order = np.lexsort((np.broadcast_to(np.arange(sim.shape[1]), sim.shape), -sim), axis=-1)[:, :k]
    return np.take_along_axis(sim, order, -1), order.astype(np.int32)
