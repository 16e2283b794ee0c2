"""Deterministic synthetic weights / inputs shared by the golden generator and the parity tests.

No checkpoint exists offline, so parity is pinned on seeded random weights of the
real architecture.  Everything here is plain numpy (PCG64 streams keyed by tensor name) so that the
generator (this container, with HF transformers + the reference importable) and the tests (GPU box,
neither available) rebuild bit-identical tensors.  Values are rounded to bfloat16 so the same
tensors are exact in fp32 and bf16 runs.
"""

from __future__ import annotations

import json
import zlib

import numpy as np

from oracle.np_ops import bf16_round
from oracle.qwen2vl_np import Cfg, TextCfg, VisionCfg


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([seed, zlib.crc32(name.encode())])


def tiny_cfg() -> Cfg:
    """Same structure as Qwen2-VL (GQA, head_dim 128, mrope [16,24,24], vision head_dim 80, patch 14, merge 2)."""
    return Cfg(
        vision=VisionCfg(depth=2, embed_dim=160, num_heads=2, mlp_ratio=4.0, hidden_size=256),
        text=TextCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                     intermediate_size=512, vocab_size=512, tie_word_embeddings=False),
        image_token_id=500,
    )


def qwen2vl_shapes(cfg: Cfg) -> dict[str, tuple]:
    v, t = cfg.vision, cfg.text
    E, F = v.embed_dim, int(v.embed_dim * v.mlp_ratio)
    d, hd = t.hidden_size, t.hidden_size // t.num_attention_heads
    s: dict[str, tuple] = {"model.visual.patch_embed.proj.weight": (E, v.in_channels, v.temporal_patch_size, v.patch_size, v.patch_size)}
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        s.update({p + "norm1.weight": (E,), p + "norm1.bias": (E,), p + "norm2.weight": (E,), p + "norm2.bias": (E,),
                  p + "attn.qkv.weight": (3 * E, E), p + "attn.qkv.bias": (3 * E,),
                  p + "attn.proj.weight": (E, E), p + "attn.proj.bias": (E,),
                  p + "mlp.fc1.weight": (F, E), p + "mlp.fc1.bias": (F,),
                  p + "mlp.fc2.weight": (E, F), p + "mlp.fc2.bias": (E,)})
    E4 = E * v.spatial_merge_size ** 2
    s.update({"model.visual.merger.ln_q.weight": (E,), "model.visual.merger.ln_q.bias": (E,),
              "model.visual.merger.mlp.0.weight": (E4, E4), "model.visual.merger.mlp.0.bias": (E4,),
              "model.visual.merger.mlp.2.weight": (v.hidden_size, E4), "model.visual.merger.mlp.2.bias": (v.hidden_size,)})
    s["model.language_model.embed_tokens.weight"] = (t.vocab_size, d)
    for i in range(t.num_hidden_layers):
        p = f"model.language_model.layers.{i}."
        s.update({p + "self_attn.q_proj.weight": (t.num_attention_heads * hd, d), p + "self_attn.q_proj.bias": (t.num_attention_heads * hd,),
                  p + "self_attn.k_proj.weight": (t.num_key_value_heads * hd, d), p + "self_attn.k_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.v_proj.weight": (t.num_key_value_heads * hd, d), p + "self_attn.v_proj.bias": (t.num_key_value_heads * hd,),
                  p + "self_attn.o_proj.weight": (d, t.num_attention_heads * hd),
                  p + "mlp.gate_proj.weight": (t.intermediate_size, d), p + "mlp.up_proj.weight": (t.intermediate_size, d),
                  p + "mlp.down_proj.weight": (d, t.intermediate_size),
                  p + "input_layernorm.weight": (d,), p + "post_attention_layernorm.weight": (d,)})
    s["model.language_model.norm.weight"] = (d,)
    if not t.tie_word_embeddings:
        s["lm_head.weight"] = (t.vocab_size, d)
    return s


_BIG = 1 << 24   # elements; no golden fixture holds a tensor this large, so the chunked generator below changes none of them


def _fill_big(name: str, shape: tuple, seed: int) -> np.ndarray:
    """Projection weights of the 7B / 34B / 72B-width slices (up to 242 M elements): row blocks drawn from independent
    float32 streams keyed by (seed, name, block) on a thread pool - same statistics as `_fill`, minutes faster."""
    import concurrent.futures as cf
    import os

    rows, fan_in = shape[0], int(np.prod(shape[1:]))
    out = np.empty((rows, fan_in), np.float32)
    step = 512
    scale = np.float32(1.0 / np.sqrt(fan_in))

    def block(r0):
        r = np.random.default_rng([seed, zlib.crc32(name.encode()), r0])
        x = r.standard_normal((min(step, rows - r0), fan_in), dtype=np.float32)
        out[r0:r0 + step] = bf16_round(x * scale)

    with cf.ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 8)) as ex:
        list(ex.map(block, range(0, rows, step)))
    return out.reshape(shape)


def _fill(name: str, shape: tuple, seed: int) -> np.ndarray:
    if int(np.prod(shape)) > _BIG and not (name.endswith("bias") or "norm" in name or "embed" in name):
        return _fill_big(name, shape, seed)
    r = _rng(seed, name)
    if name.endswith("bias"):
        x = 0.05 * r.standard_normal(shape)
    elif "norm" in name or "ln_q" in name or "LayerNorm" in name:
        x = 1.0 + 0.1 * r.standard_normal(shape)
    elif "embed" in name or "embeddings" in name:
        x = 0.5 * r.standard_normal(shape)
    else:
        fan_in = int(np.prod(shape[1:]))
        x = r.standard_normal(shape) / np.sqrt(fan_in)
    return bf16_round(x.astype(np.float32))


def qwen2vl_weights(cfg: Cfg, seed: int = 1234) -> dict[str, np.ndarray]:
    return {k: _fill(k, shp, seed) for k, shp in qwen2vl_shapes(cfg).items()}


def tiny_cfg25():
    """Qwen2.5-VL miniature with the real structure: RMSNorm + gated-MLP vision blocks (intermediate 420 is NOT a multiple of 16,
    like the real 3420), vision head_dim 80, 112-pixel windows, three blocks of which the middle one is a full-attention block."""
    from oracle.qwen25vl_np import Cfg25, Vision25Cfg

    return Cfg25(vision=Vision25Cfg(depth=3, embed_dim=160, num_heads=2, intermediate_size=420, hidden_size=256,
                                    window_size=112, fullatt_block_indexes=(1,)),
                 text=TextCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                              intermediate_size=512, vocab_size=512, tie_word_embeddings=False), image_token_id=500)


def qwen25vl_shapes(cfg) -> dict[str, tuple]:
    v = cfg.vision
    E, F = v.embed_dim, v.intermediate_size
    s = {k: shp for k, shp in qwen2vl_shapes(Cfg(vision=VisionCfg(depth=0, embed_dim=E, num_heads=v.num_heads, hidden_size=v.hidden_size),
                                                 text=cfg.text, image_token_id=cfg.image_token_id)).items()
         if "merger.ln_q.bias" not in k}
    for i in range(v.depth):
        p = f"model.visual.blocks.{i}."
        s.update({p + "norm1.weight": (E,), p + "norm2.weight": (E,), p + "attn.qkv.weight": (3 * E, E), p + "attn.qkv.bias": (3 * E,),
                  p + "attn.proj.weight": (E, E), p + "attn.proj.bias": (E,),
                  p + "mlp.gate_proj.weight": (F, E), p + "mlp.gate_proj.bias": (F,), p + "mlp.up_proj.weight": (F, E),
                  p + "mlp.up_proj.bias": (F,), p + "mlp.down_proj.weight": (E, F), p + "mlp.down_proj.bias": (E,)})
    return s


def qwen25vl_weights(cfg, seed: int = 1234) -> dict[str, np.ndarray]:
    return {k: _fill(k, shp, seed) for k, shp in qwen25vl_shapes(cfg).items()}


def pixel_values(grid_thw, seed: int = 7) -> np.ndarray:
    """Synthetic normalised patches [sum(t*h*w), 1176], bf16-representable."""
    n = int(sum(t * h * w for t, h, w in grid_thw))
    return bf16_round(_rng(seed, "pixel_values").standard_normal((n, 1176)).astype(np.float32))


def prompt_ids(cfg: Cfg, grid_thw, n_text_before=5, n_text_after=9, seed=11) -> np.ndarray:
    """[text..., <img tokens per image>..., text...] with ids below the special-token range."""
    r = _rng(seed, "prompt")
    merge2 = cfg.vision.spatial_merge_size ** 2
    hi = min(cfg.image_token_id, cfg.text.vocab_size) - 1
    parts = [r.integers(1, hi, n_text_before)]
    for t, h, w in grid_thw:
        parts.append(np.full(t * h * w // merge2, cfg.image_token_id))
        parts.append(r.integers(1, hi, 2))
    parts.append(r.integers(1, hi, n_text_after))
    return np.concatenate(parts).astype(np.int64)


# ---------------------------------------------------------------- BERT / MiniLM
def bert_cfg(kind: str = "tiny") -> dict:
    if kind == "tiny":
        return dict(vocab_size=120, hidden_size=64, num_hidden_layers=2, num_attention_heads=2,
                    intermediate_size=128, max_position_embeddings=64, type_vocab_size=2, layer_norm_eps=1e-12)
    # sentence-transformers/all-MiniLM-L6-v2 (src/data/pipelines/text/_text.py:161)
    return dict(vocab_size=30522, hidden_size=384, num_hidden_layers=6, num_attention_heads=12,
                intermediate_size=1536, max_position_embeddings=512, type_vocab_size=2, layer_norm_eps=1e-12)


def bert_shapes(c: dict) -> dict[str, tuple]:
    H, I = c["hidden_size"], c["intermediate_size"]
    s = {"embeddings.word_embeddings.weight": (c["vocab_size"], H),
         "embeddings.position_embeddings.weight": (c["max_position_embeddings"], H),
         "embeddings.token_type_embeddings.weight": (c["type_vocab_size"], H),
         "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,)}
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("query", "key", "value"):
            s[p + f"attention.self.{n}.weight"] = (H, H)
            s[p + f"attention.self.{n}.bias"] = (H,)
        s.update({p + "attention.output.dense.weight": (H, H), p + "attention.output.dense.bias": (H,),
                  p + "attention.output.LayerNorm.weight": (H,), p + "attention.output.LayerNorm.bias": (H,),
                  p + "intermediate.dense.weight": (I, H), p + "intermediate.dense.bias": (I,),
                  p + "output.dense.weight": (H, I), p + "output.dense.bias": (H,),
                  p + "output.LayerNorm.weight": (H,), p + "output.LayerNorm.bias": (H,)})
    return s


def bert_weights(c: dict, seed: int = 1234) -> dict[str, np.ndarray]:
    out = {}
    for k, shp in bert_shapes(c).items():
        r = _rng(seed, "bert." + k)
        if k.endswith("LayerNorm.weight"):
            x = 1.0 + 0.1 * r.standard_normal(shp)
        elif k.endswith("bias"):
            x = 0.05 * r.standard_normal(shp)
        elif "embeddings" in k:
            x = 0.3 * r.standard_normal(shp)
        else:
            x = 2.0 * r.standard_normal(shp) / np.sqrt(shp[1])
        out[k] = x.astype(np.float32)
    return out


def mpnet_cfg(kind: str = "tiny") -> dict:
    """MPNetConfig fields.  "base" = sentence-transformers/all-mpnet-base-v2 (BASELINE.json configs[0]'s wording; public config.json)."""
    if kind == "tiny":
        return dict(vocab_size=120, hidden_size=128, num_hidden_layers=2, num_attention_heads=2, intermediate_size=256,
                    max_position_embeddings=66, layer_norm_eps=1e-5, relative_attention_num_buckets=32)
    return dict(vocab_size=30527, hidden_size=768, num_hidden_layers=12, num_attention_heads=12, intermediate_size=3072,
                max_position_embeddings=514, layer_norm_eps=1e-5, relative_attention_num_buckets=32)


def mpnet_shapes(c: dict) -> dict[str, tuple]:
    H, I = c["hidden_size"], c["intermediate_size"]
    s = {"embeddings.word_embeddings.weight": (c["vocab_size"], H),
         "embeddings.position_embeddings.weight": (c["max_position_embeddings"], H),
         "embeddings.LayerNorm.weight": (H,), "embeddings.LayerNorm.bias": (H,),
         "encoder.relative_attention_bias.weight": (c["relative_attention_num_buckets"], c["num_attention_heads"])}
    for i in range(c["num_hidden_layers"]):
        p = f"encoder.layer.{i}."
        for n in ("q", "k", "v", "o"):
            s[p + f"attention.attn.{n}.weight"] = (H, H)
            s[p + f"attention.attn.{n}.bias"] = (H,)
        s.update({p + "attention.LayerNorm.weight": (H,), p + "attention.LayerNorm.bias": (H,),
                  p + "intermediate.dense.weight": (I, H), p + "intermediate.dense.bias": (I,),
                  p + "output.dense.weight": (H, I), p + "output.dense.bias": (H,),
                  p + "output.LayerNorm.weight": (H,), p + "output.LayerNorm.bias": (H,)})
    return s


def mpnet_weights(c: dict, seed: int = 1234) -> dict[str, np.ndarray]:
    out = {}
    for k, shp in mpnet_shapes(c).items():
        r = _rng(seed, "mpnet." + k)
        if k.endswith("LayerNorm.weight"):
            x = 1.0 + 0.1 * r.standard_normal(shp)
        elif k.endswith("bias"):
            x = 0.05 * r.standard_normal(shp)
        elif "relative_attention_bias" in k:
            x = 1.5 * r.standard_normal(shp)     # a bias that matters: of the order of the scaled scores
        elif "embeddings" in k:
            x = 0.3 * r.standard_normal(shp)
        else:
            x = 2.0 * r.standard_normal(shp) / np.sqrt(shp[1])
        out[k] = x.astype(np.float32)
    return out


def mpnet_label_tokens(n: int, L: int, vocab: int, seed: int):
    """Like `label_tokens` with MPNet's padding id (1): real ids in [2, vocab), pad = 1."""
    ids, mask = label_tokens(n, L, vocab - 1, seed)
    return np.where(mask > 0, ids + 1, 1).astype(np.int64), mask


def label_tokens(n: int, L: int, vocab: int, seed: int):
    """Synthetic tokenised labels: ids [n, L] (0 = pad), mask [n, L]; lengths uniform in [2, L]."""
    r = _rng(seed, "labels")
    lens = r.integers(2, L + 1, n)
    lens[0] = L  # "padding=True" pads to the longest member
    ids = r.integers(1, vocab, (n, L))
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int64)
    return (ids * mask).astype(np.int64), mask


# ---------------------------------------------------------------- LLaVA (CLIP ViT + projector + Llama decoder)
def tiny_llava_cfg():
    from oracle.llava_np import ClipCfg, LlavaCfg

    return LlavaCfg(vision=ClipCfg(hidden_size=128, intermediate_size=256, num_hidden_layers=3, num_attention_heads=2,
                                   image_size=56, patch_size=14, layer_norm_eps=1e-5),
                    text=TextCfg(hidden_size=256, num_hidden_layers=2, num_attention_heads=2, num_key_value_heads=1,
                                 intermediate_size=512, vocab_size=512, rms_norm_eps=1e-5, rope_theta=10000.0,
                                 tie_word_embeddings=False),
                    image_token_id=500, vision_feature_layer=-2)


TINY_PINPOINTS = ((56, 112), (112, 56), (112, 112), (168, 56), (56, 168))


def tiny_llava_next_cfg():
    cfg = tiny_llava_cfg()
    cfg.image_grid_pinpoints = TINY_PINPOINTS
    return cfg


def llava_shapes(cfg) -> dict[str, tuple]:
    v, t = cfg.vision, cfg.text
    E, F, d = v.hidden_size, v.intermediate_size, t.hidden_size
    hd = d // t.num_attention_heads
    g = v.image_size // v.patch_size
    VT = "model.vision_tower."
    s = {VT + "embeddings.class_embedding": (E,), VT + "embeddings.patch_embedding.weight": (E, 3, v.patch_size, v.patch_size),
         VT + "embeddings.position_embedding.weight": (g * g + 1, E), VT + "pre_layrnorm.weight": (E,), VT + "pre_layrnorm.bias": (E,)}
    for i in range(v.num_hidden_layers):
        p = f"{VT}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"] = (E, E)
            s[p + f"self_attn.{n}.bias"] = (E,)
        s.update({p + "layer_norm1.weight": (E,), p + "layer_norm1.bias": (E,), p + "layer_norm2.weight": (E,), p + "layer_norm2.bias": (E,),
                  p + "mlp.fc1.weight": (F, E), p + "mlp.fc1.bias": (F,), p + "mlp.fc2.weight": (E, F), p + "mlp.fc2.bias": (E,)})
    P = "model.multi_modal_projector."
    s.update({P + "linear_1.weight": (d, E), P + "linear_1.bias": (d,), P + "linear_2.weight": (d, d), P + "linear_2.bias": (d,)})
    if cfg.image_grid_pinpoints:
        s["model.image_newline"] = (d,)
    s["model.language_model.embed_tokens.weight"] = (t.vocab_size, d)
    for i in range(t.num_hidden_layers):
        p = f"model.language_model.layers.{i}."
        s.update({p + "self_attn.q_proj.weight": (t.num_attention_heads * hd, d), p + "self_attn.k_proj.weight": (t.num_key_value_heads * hd, d),
                  p + "self_attn.v_proj.weight": (t.num_key_value_heads * hd, d), p + "self_attn.o_proj.weight": (d, t.num_attention_heads * hd),
                  p + "mlp.gate_proj.weight": (t.intermediate_size, d), p + "mlp.up_proj.weight": (t.intermediate_size, d),
                  p + "mlp.down_proj.weight": (d, t.intermediate_size), p + "input_layernorm.weight": (d,),
                  p + "post_attention_layernorm.weight": (d,)})
    s["model.language_model.norm.weight"] = (d,)
    s["lm_head.weight"] = (t.vocab_size, d)
    return s


def llava_weights(cfg, seed: int = 1234) -> dict[str, np.ndarray]:
    out = {}
    for k, shp in llava_shapes(cfg).items():
        if k.endswith("class_embedding") or "position_embedding" in k or k.endswith("image_newline"):
            out[k] = bf16_round((0.3 * _rng(seed, k).standard_normal(shp)).astype(np.float32))
        else:
            out[k] = _fill(k, shp, seed)
    return out


def clip_pixels(n: int, size: int, seed: int = 31) -> np.ndarray:
    return bf16_round(_rng(seed, "clip_pixels").standard_normal((n, 3, size, size)).astype(np.float32))


# ---------------------------------------------------------------- eval_ranking.py inputs
def ranking_runs(root, n_docs: int = 30):
    """Synthetic `logs/schedule/{task}/{model}/*_samples_*.jsonl` runs; answers/targets are strings of token ids
    (IdTokenizer) so the same files feed the reference on CPU and the HIP scorer on the GPU box."""
    r = np.random.default_rng(77)
    vocab = bert_cfg("tiny")["vocab_size"]
    targets = [" ".join(str(int(t)) for t in r.integers(3, vocab, r.integers(2, 5))) for _ in range(n_docs)]
    for mi, model in enumerate(("model-a", "model-b", "model-c")):
        d = root / "toytask" / model
        d.mkdir(parents=True, exist_ok=True)
        rows = []
        for doc_id, tgt in enumerate(targets):
            toks = tgt.split()
            if r.random() < 0.35 + 0.2 * mi:   # model-c copies the target most often
                ans = tgt
            else:
                ans = " ".join(toks[:1] + [str(int(t)) for t in r.integers(3, vocab, r.integers(1, 6))])
            rows.append({"doc_id": doc_id, "filtered_resps": [ans], "target": tgt})
        (d / f"2026_samples_toytask.jsonl").write_text("\n".join(json.dumps(x) for x in rows) + "\n")
    return targets


# ---------------------------------------------------------------- toy task / text fixtures (tools/gen_golden_formats.py)
class HashTokenizer:
    """Stand-in for AutoTokenizer on free text (no tokenizer files offline): [CLS]=1, one id per whitespace word
    (3 + crc32(lower-cased word) % (vocab - 3)), [SEP]=2, pad 0, pad-to-longest.  Used on BOTH sides: injected into the
    reference (`_text.sentence_bert_processor`) by the golden generator and into the HIP scorer by the GPU tests."""

    def __init__(self, vocab_size: int, max_len: int = 32):
        self.vocab, self.max_len = vocab_size, max_len

    def encode(self, s: str) -> list[int]:
        words = s.lower().split()[: self.max_len - 2]
        return [1] + [3 + zlib.crc32(w.encode()) % (self.vocab - 3) for w in words] + [2]

    def __call__(self, text, padding=True, truncation=True, return_tensors="np"):
        rows = [self.encode(s) for s in text]
        L = max(len(r) for r in rows)
        ids = np.array([r + [0] * (L - len(r)) for r in rows], dtype=np.int64)
        mask = np.array([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=np.int64)
        if return_tensors == "pt":
            import torch

            class Enc(dict):
                def to(self, *a, **k):
                    return self

            return Enc(input_ids=torch.from_numpy(ids), attention_mask=torch.from_numpy(mask))
        return {"input_ids": ids, "attention_mask": mask}


def toy_nlp(text: str):
    """Rule-based stand-in for spaCy's parser: noun chunks = the pieces between ',', ' and ', ' with ', ' of ', ' in ';
    entities = maximal runs of Capitalised words.  Returns (noun_chunks, ents) as raw strings (case kept)."""
    import re

    chunks = [c.strip(" .!?") for c in re.split(r",| and | with | of | in ", text)]
    chunks = [c for c in chunks if c]
    ents = [m.group(0) for m in re.finditer(r"\b[A-Z][a-z]+(?: [A-Z][a-z]+)*\b", text)]
    return chunks, ents


TOY_CLASSES = ["golden_retriever", "sea_lion", "tabby_cat", "airplane", "grand_piano"]


def toy_docs(n: int = 11) -> list[dict]:
    return [{"visual": f"img{i}.jpg", "target": TOY_CLASSES[i % len(TOY_CLASSES)]} for i in range(n)]


def toy_answer(doc_id: int, target: str) -> str:
    """Deterministic stand-in answer of the stand-in model: exact / embedded / decorated / wrong, by document."""
    t = target.replace("_", " ")
    return [t, f"A photo of a {t}", f"The {t.title()} and a red ball", "something else entirely", f" {t.upper()}, $5 ",
            f"this photo shows the {t} in a garden", f"{t}."][doc_id % 7]


def toy_multi_round_answers(doc_id: int, target: str, doc_to_text, doc) -> tuple:
    """Stand-in for a wrapper's `generate_until_multi_round` on one request: rounds until the task's terminal signal; the last
    round answers like `toy_answer`, the earlier ones carry a round tag."""
    answers, r = [f"{toy_answer(doc_id, target)} [round 0]"], 1
    while True:
        out = doc_to_text(doc, round_idx=r, previous_round_results=list(answers), last_round_info=None)
        answers = list(out[3])
        if out[2]:
            break
        answers.append(f"{toy_answer(doc_id, target)} [round {r}]")
        r += 1
    answers[-1] = toy_answer(doc_id, target)
    return tuple(answers)


def toy_concept_items() -> list:
    """(ref, pred) pairs for concept_semantic_similarity: str / [str] refs, str / [.., str] preds, duplicates, prefix words
    ('the', 'a', 'its'), skip words ('this photo', 'image', 'it'), capitalised entities, a one-concept prediction."""
    return [("sea lion", "The sea lion and its ball"), (["tabby cat"], "a Tabby Cat in this photo, it sleeps"),
            ("airplane", ["ignored first round", "an image of the airplane with two wings"]), ("sea lion", "The sea lion and its ball"),
            ("grand piano", "piano"), ("golden retriever", "A dog, the Golden Retriever of Scotland and a stick"),
            (["airplane"], ["Boeing"]), ("tabby cat", "what type of object, some cat")]


# ---------------------------------------------------------------------------------- multi-round protocol (tools/gen_golden_wrappers.py)
def mr_answer_of(prompt_text: str) -> str:
    """The stand-in checkpoint of the multi-round protocol golden: a deterministic function of the rendered prompt.  Some answers
    carry the stop string and a tail that the `until` cut must remove."""
    import hashlib

    h = hashlib.sha1(prompt_text.encode()).hexdigest()
    words = ["maple", "otter", "quartz", "lantern", "violet", "harbor", "falcon", "cedar"]
    body = " ".join(words[int(c, 16) % 8] for c in h[:3])
    if int(h[3], 16) % 3 == 0:
        body += " STOP trailing words " + h[4:8]
    return body


def mr_image_of(seed: int, size=(40, 56)):
    from PIL import Image

    r = np.random.default_rng(seed)
    return Image.fromarray(r.integers(0, 256, (size[1], size[0], 3), dtype=np.uint8), "RGB")


def mr_docs_and_task():
    """Three docs and a multi-round `doc_to_text` (test input, the 5-tuple protocol of the reference's tasks): at most three rounds
    (two for doc 2), doc 1's second round brings a second image, doc 2 has two round-0 images, a doc ends early when its last
    answer contains 'cedar'."""
    docs = [{"id": i, "label": f"class{i}", "seed": 100 + i} for i in range(3)]

    def doc_to_visual(doc):   # (doc 2 has two images: the reference's round 0 shows the model only the first, _qwen2_vl.py:409-413 / :482-483)
        return [mr_image_of(doc["seed"])] + ([mr_image_of(doc["seed"] + 7, (56, 84))] if doc["id"] == 2 else [])

    def doc_to_text(doc, round_idx=0, previous_round_results=None, last_round_info=None):
        prev = list(previous_round_results or [])
        n_rounds = 3 if doc["id"] != 2 else 2
        terminal = round_idx >= n_rounds or any("cedar" in p for p in prev[-1:])
        visuals = [mr_image_of(doc["seed"] + 50)] if (doc["id"] == 1 and round_idx == 1) else []
        text = f"Round {round_idx} for {doc['label']}: given {' | '.join(prev) if prev else 'nothing'}, refine the answer."
        return visuals, text, terminal, prev, last_round_info

    return docs, doc_to_visual, doc_to_text


def mr_context(doc: dict) -> str:
    return f"<image>What type of object is in this photo? ({doc['label']})" if doc["id"] != 2 else "Describe the photo."


def su_docs_and_task():
    """Single-round protocol golden: six docs - one image each, except doc 3 (two images) and doc 4 (none); contexts with and
    without an `<image>` marker, of different lengths (the Collator sorts by them), one with surrounding blanks."""
    docs = [{"id": i, "label": f"class{i}", "seed": 200 + i} for i in range(6)]

    def doc_to_visual(doc):
        if doc["id"] == 4:
            return []
        return [mr_image_of(doc["seed"])] + ([mr_image_of(doc["seed"] + 9, (84, 56))] if doc["id"] == 3 else [])

    return docs, doc_to_visual


def su_context(doc: dict) -> str:
    i = doc["id"]
    return ["What type of object is in this photo?", "<image>What is this? Answer with one word.", "Name the object.  ",
            "<image> <image>\nWhat do the two photos have in common?", "No picture here: say hello.",
            "What type of object is in this photo? Let's think step by step."][i] + (f" ({doc['label']})" if i in (0, 5) else "")


def ll_logits(ids, vocab: int = 272) -> np.ndarray:
    """Stand-in decoder of the loglikelihood protocol golden: logits [S, vocab] as a deterministic function of the id prefix; for
    sequences with an even id sum position t prefers ids[t] (so the reference's UNSHIFTED greedy comparison is true for some requests)."""
    import zlib

    ids = np.asarray(ids, np.int32)
    out = np.empty((len(ids), vocab), np.float64)
    boost = int(ids.sum()) % 2 == 0
    for t in range(len(ids)):
        out[t] = np.random.default_rng(zlib.crc32(ids[: t + 1].tobytes())).normal(size=vocab) * 2.0
        if boost:
            out[t, ids[t]] += 12.0
    return out


def ll_continuation(doc: dict) -> str:
    return f" a {doc['label']} thing"


def ll_context(doc: dict) -> str:
    """(loglikelihood contexts carry no `<image>` marker: the reference prepends one per image unconditionally, _llava_hf.py:204-206)"""
    return su_context(doc).replace("<image>", "").lstrip()
