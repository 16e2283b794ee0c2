"""Sentence-embedding scorer on libowc_hip.so (fp32): BERT/MiniLM encoder + cosine pairing / top-k.

Replaces the arithmetic behind the reference's `encode_sentence_bert`
(/root/reference/src/data/pipelines/text/_text.py:143-208) and `semantic_similarity`
(/root/reference/src/data/metrics/_group.py:488-544).  Tokenisation stays on the host.
"""

from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib, ops

F32, I32 = torch.float32, torch.int32

MINILM_L6 = dict(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                 intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)


def mpnet_relative_bucket(rel: torch.Tensor, num_buckets: int = 32, max_distance: int = 128) -> torch.Tensor:
    """MPNet's bucket of a relative position `rel` = key column - query column (HF modeling_mpnet.py, MPNetEncoder.
    relative_position_bucket - index bookkeeping, evaluated once per checkpoint with the float32 log HF itself uses): half of the
    buckets per direction, one bucket per distance below 8, logarithmic bins from 8 to 128, the last bin beyond."""
    import math

    n = -rel.long()
    half = num_buckets // 2
    ret = (n < 0).long() * half
    n = n.abs()
    max_exact = half // 2
    large = max_exact + (torch.log(n.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)).long()
    large = torch.minimum(large, torch.full_like(large, half - 1))
    return ret + torch.where(n < max_exact, n, large)


class BertWeights:
    """fp32 device weights of a BertModel (HF names without the `bert.` prefix) + the C struct."""

    def __init__(self, cfg: dict, sd, device):
        self.cfg = cfg
        self.device = torch.device(device)
        self._keep = []

        def get(name):
            t = sd[name]
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t))
            t = t.to(device=self.device, dtype=F32).contiguous()
            self._keep.append(t)
            return t

        n = cfg["num_hidden_layers"]
        # MPNet (sentence-transformers/all-mpnet-base-v2, BASELINE.json configs[0]) names its attention `attention.attn.{q,k,v,o}` +
        # `attention.LayerNorm` and carries ONE relative-position bias table for all layers; BERT (all-MiniLM-L6-v2, what the
        # reference's encode_sentence_bert loads, _text.py:161) `attention.self.{query,key,value}` + `attention.output.*`
        self.mpnet = "encoder.relative_attention_bias.weight" in sd
        qkv_names = [f"attention.attn.{k}" for k in "qkv"] if self.mpnet else [f"attention.self.{k}" for k in ("query", "key", "value")]
        o_name, ln1_name = ("attention.attn.o", "attention.LayerNorm") if self.mpnet else ("attention.output.dense", "attention.output.LayerNorm")
        layers = (_lib.BertLayer * n)()
        for i in range(n):
            p = f"encoder.layer.{i}."
            qkv_w = torch.cat([get(p + k + ".weight") for k in qkv_names], 0).contiguous()
            qkv_b = torch.cat([get(p + k + ".bias") for k in qkv_names], 0).contiguous()
            self._keep += [qkv_w, qkv_b]
            L = layers[i]
            L.qkv_w, L.qkv_b = qkv_w.data_ptr(), qkv_b.data_ptr()
            L.o_w, L.o_b = get(p + o_name + ".weight").data_ptr(), get(p + o_name + ".bias").data_ptr()
            L.ln1_w = get(p + ln1_name + ".weight").data_ptr()
            L.ln1_b = get(p + ln1_name + ".bias").data_ptr()
            L.fc1_w, L.fc1_b = get(p + "intermediate.dense.weight").data_ptr(), get(p + "intermediate.dense.bias").data_ptr()
            L.fc2_w, L.fc2_b = get(p + "output.dense.weight").data_ptr(), get(p + "output.dense.bias").data_ptr()
            L.ln2_w, L.ln2_b = get(p + "output.LayerNorm.weight").data_ptr(), get(p + "output.LayerNorm.bias").data_ptr()
        self._layers = layers
        w = _lib.BertWeights()
        w.n_layers, w.hidden, w.n_heads, w.inter = n, cfg["hidden_size"], cfg["num_attention_heads"], cfg["intermediate_size"]
        w.vocab, w.max_pos, w.ln_eps = cfg["vocab_size"], cfg["max_position_embeddings"], cfg["layer_norm_eps"]
        w.word_emb = get("embeddings.word_embeddings.weight").data_ptr()
        w.pos_emb = get("embeddings.position_embeddings.weight").data_ptr()
        if self.mpnet:
            w.type_emb, w.pos_offset = None, 2                  # no token types; position ids = column + padding_idx + 1
            span = min(int(cfg["max_position_embeddings"]) - 2, 512)
            table = get("encoder.relative_attention_bias.weight")                            # [buckets, heads]
            rel = torch.arange(-(span - 1), span, device=self.device)                         # key column - query column
            bias = table[mpnet_relative_bucket(rel, table.shape[0])].t().contiguous()        # [heads, 2 span - 1]
            self._keep.append(bias)
            w.rel_bias, w.rel_span = bias.data_ptr(), span
        else:
            w.type_emb = get("embeddings.token_type_embeddings.weight").data_ptr()
        w.emb_ln_w = get("embeddings.LayerNorm.weight").data_ptr()
        w.emb_ln_b = get("embeddings.LayerNorm.bias").data_ptr()
        w.layers = C.cast(layers, C.POINTER(_lib.BertLayer))
        self.c = w

    @classmethod
    def random(cls, cfg: dict, device, seed: int = 1234) -> "BertWeights":
        g = torch.Generator().manual_seed(seed)
        H, I = cfg["hidden_size"], cfg["intermediate_size"]
        sd = {"embeddings.word_embeddings.weight": torch.randn(cfg["vocab_size"], H, generator=g) * 0.3,
              "embeddings.position_embeddings.weight": torch.randn(cfg["max_position_embeddings"], H, generator=g) * 0.3,
              "embeddings.token_type_embeddings.weight": torch.randn(cfg["type_vocab_size"], H, generator=g) * 0.3,
              "embeddings.LayerNorm.weight": torch.ones(H), "embeddings.LayerNorm.bias": torch.zeros(H)}
        for i in range(cfg["num_hidden_layers"]):
            p = f"encoder.layer.{i}."
            for k in ("query", "key", "value"):
                sd[p + f"attention.self.{k}.weight"] = torch.randn(H, H, generator=g) * (2.0 / np.sqrt(H))
                sd[p + f"attention.self.{k}.bias"] = torch.randn(H, generator=g) * 0.05
            sd[p + "attention.output.dense.weight"] = torch.randn(H, H, generator=g) * (2.0 / np.sqrt(H))
            sd[p + "attention.output.dense.bias"] = torch.randn(H, generator=g) * 0.05
            sd[p + "attention.output.LayerNorm.weight"] = torch.ones(H)
            sd[p + "attention.output.LayerNorm.bias"] = torch.zeros(H)
            sd[p + "intermediate.dense.weight"] = torch.randn(I, H, generator=g) * (2.0 / np.sqrt(H))
            sd[p + "intermediate.dense.bias"] = torch.randn(I, generator=g) * 0.05
            sd[p + "output.dense.weight"] = torch.randn(H, I, generator=g) * (2.0 / np.sqrt(I))
            sd[p + "output.dense.bias"] = torch.randn(H, generator=g) * 0.05
            sd[p + "output.LayerNorm.weight"] = torch.ones(H)
            sd[p + "output.LayerNorm.bias"] = torch.zeros(H)
        return cls(cfg, sd, device)


class SentenceScorer:
    """Embeds token-id matrices and scores them by cosine similarity on the GPU."""

    def __init__(self, weights: BertWeights, max_batch: int = 8192):
        self.w = weights
        self.device = weights.device
        self.dev_index = self.device.index or 0
        self.max_batch = max_batch
        self._ws = None
        self._lib = _lib.load()
        self._ctx = _lib.ctx(self.dev_index)

    def embed(self, ids, mask, packed: bool = True) -> torch.Tensor:
        """ids/mask: int [n, L] (numpy or torch) -> L2-normalised fp32 embeddings [n, hidden] on the device.
        `packed` (default): only the tokens with mask == 1 become rows (`owc_bert_embed_packed`) - identical results, no work
        on padding; the layout (token list, position ids, sequence offsets) is integer bookkeeping done on the host, where
        the tokenizer's output lives anyway.  `packed=False` runs the padded `owc_bert_embed`."""
        if not packed:
            return self._embed_padded(ids, mask)
        ids_h = ids.cpu().numpy() if isinstance(ids, torch.Tensor) else np.asarray(ids)
        mask_h = (mask.cpu().numpy() if isinstance(mask, torch.Tensor) else np.asarray(mask)) != 0
        n, L = ids_h.shape
        H = self.w.cfg["hidden_size"]
        out = torch.empty((n, H), dtype=F32, device=self.device)
        lens = mask_h.sum(1).astype(np.int64)
        for i0 in range(0, n, self.max_batch):   # batches of whole sequences
            i1 = min(n, i0 + self.max_batch)
            m = mask_h[i0:i1]
            T = int(m.sum())
            if T == 0:
                out[i0:i1] = float("nan")   # no token at all: the reference divides 0 by a zero norm (_text.py:202)
                continue
            tok_ids = np.ascontiguousarray(ids_h[i0:i1][m], dtype=np.int32)
            tok_pos = np.ascontiguousarray(np.broadcast_to(np.arange(L, dtype=np.int32), m.shape)[m])
            seq_start = np.concatenate([[0], np.cumsum(lens[i0:i1])]).astype(np.int32)
            d_ids, d_pos, d_start = (_lib.h2d(a, self.device) for a in (tok_ids, tok_pos, seq_start))
            nb = self._lib.owc_bert_packed_workspace_bytes(C.byref(self.w.c), T)
            if self._ws is None or self._ws.numel() < nb:
                self._ws = None
                self._ws = torch.empty(int(nb), dtype=torch.uint8, device=self.device)
            rc = self._lib.owc_bert_embed_packed(self._ctx, C.byref(self.w.c), d_ids.data_ptr(), d_pos.data_ptr(), d_start.data_ptr(),
                                                 i1 - i0, T, int(lens[i0:i1].max()), out[i0:i1].data_ptr(), self._ws.data_ptr(),
                                                 self._ws.numel(), _lib.stream_ptr())
            _lib.check(rc, self.dev_index)
        return out

    def _embed_padded(self, ids, mask) -> torch.Tensor:
        if isinstance(ids, np.ndarray):
            ids = torch.from_numpy(np.ascontiguousarray(ids))
        if isinstance(mask, np.ndarray):
            mask = torch.from_numpy(np.ascontiguousarray(mask))
        ids = ids.to(device=self.device, dtype=I32).contiguous()
        mask = mask.to(device=self.device, dtype=I32).contiguous()
        n, L = ids.shape
        H = self.w.cfg["hidden_size"]
        out = torch.empty((n, H), dtype=F32, device=self.device)
        for i0 in range(0, n, self.max_batch):
            i1 = min(n, i0 + self.max_batch)
            nb = self._lib.owc_bert_workspace_bytes(C.byref(self.w.c), i1 - i0, L)
            if self._ws is None or self._ws.numel() < nb:
                self._ws = None
                self._ws = torch.empty(int(nb), dtype=torch.uint8, device=self.device)
            rc = self._lib.owc_bert_embed(self._ctx, C.byref(self.w.c), ids[i0:i1].data_ptr(), mask[i0:i1].data_ptr(),
                                          i1 - i0, L, out[i0:i1].data_ptr(), self._ws.data_ptr(), self._ws.numel(),
                                          _lib.stream_ptr())
            _lib.check(rc, self.dev_index)
        return out

    @staticmethod
    def paired_cosine(refs_z: torch.Tensor, preds_z: torch.Tensor) -> torch.Tensor:
        return ops.paired_dot(refs_z, preds_z)

    @staticmethod
    def topk(preds_z: torch.Tensor, classes_z: torch.Tensor, k: int, label: torch.Tensor | None = None):
        return ops.cosine_topk(preds_z, classes_z, k, label)
