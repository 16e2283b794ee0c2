"""Generate tests/golden/*.npz|json by running the REFERENCE (and the HF code it wraps) in this container.

Runs only where /root/reference and transformers are importable (the build container).  Inputs and
weights come from tests/recipes.py (pure numpy, seeded) so the tests can rebuild them anywhere; only
inputs + expected outputs are written — never reference source.  Versions are recorded in each file.

  python tools/gen_golden.py
"""

from __future__ import annotations

import importlib.machinery
import importlib.util
import json
import sys
import zlib
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference")

from tests import recipes  # noqa: E402


def versions() -> dict:
    import datasets
    import transformers

    return {"transformers": transformers.__version__, "torch": torch.__version__, "datasets": datasets.__version__,
            "numpy": np.__version__, "reference_pins": "transformers==4.47.0 torch==2.5.1 datasets==3.1.0 (uv.lock)"}


# ------------------------------------------------------------------ HF Qwen2-VL (third-party arithmetic)
def hf_qwen(cfg, weights, dtype):
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    v, t = cfg.vision, cfg.text
    hcfg = Qwen2VLConfig(
        text_config=dict(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                         num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                         intermediate_size=t.intermediate_size, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                         max_position_embeddings=4096, tie_word_embeddings=t.tie_word_embeddings,
                         rope_parameters=dict(rope_type="default", rope_theta=t.rope_theta, mrope_section=list(t.mrope_section))),
        vision_config=dict(depth=v.depth, embed_dim=v.embed_dim, num_heads=v.num_heads, hidden_size=v.hidden_size,
                           mlp_ratio=int(v.mlp_ratio), patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size),
        image_token_id=cfg.image_token_id, video_token_id=cfg.image_token_id + 1,
        vision_start_token_id=cfg.image_token_id + 2, vision_end_token_id=cfg.image_token_id + 3,
        tie_word_embeddings=t.tie_word_embeddings)
    hcfg._attn_implementation = "eager"
    m = Qwen2VLForConditionalGeneration(hcfg)
    sd = {k: torch.from_numpy(a.copy()) for k, a in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return m.to(dtype).eval()


def gen_qwen():
    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, 1234)
    out = {}
    cases = {"a": [(1, 4, 4)], "b": [(1, 6, 4), (1, 4, 8)]}
    for name, grid in cases.items():
        pix = recipes.pixel_values(grid, seed=7)
        ids = recipes.prompt_ids(cfg, grid, seed=11)
        for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            m = hf_qwen(cfg, w, dtype)
            g = torch.tensor(grid)
            inp = torch.from_numpy(ids)[None]
            mm = (inp == cfg.image_token_id).int()
            with torch.no_grad():
                vis = m.model.visual(torch.from_numpy(pix).to(dtype), grid_thw=g).pooler_output
                gen = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype),
                                 image_grid_thw=g, mm_token_type_ids=mm, do_sample=False, num_beams=1, max_new_tokens=8,
                                 use_cache=True, eos_token_id=None, pad_token_id=0, output_logits=True,
                                 return_dict_in_generate=True)
                pos, delta = m.model.get_rope_index(inp, mm_token_type_ids=mm, image_grid_thw=g)
            out[f"{name}_{tag}_vit"] = torch.cat(list(vis), 0).float().numpy() if isinstance(vis, (list, tuple)) else vis.float().numpy()
            out[f"{name}_{tag}_tokens"] = gen.sequences[0, inp.shape[1]:].numpy()
            out[f"{name}_{tag}_logits"] = torch.stack([l[0] for l in gen.logits]).float().numpy()
            out[f"{name}_pos3"] = pos[:, 0].numpy()
            out[f"{name}_delta"] = np.array(int(delta[0, 0]))
        out[f"{name}_grid"] = np.array(grid)
        out[f"{name}_ids"] = ids
    # G5: rope index of the benchmark prompt shape (448x448 -> 256 image tokens, S = 286)
    big = Cfg448()
    out["p448_ids"], out["p448_pos3"], out["p448_delta"] = big
    np.savez_compressed(GOLD / "qwen2vl_tiny.npz", **out)
    (GOLD / "qwen2vl_tiny.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234,
                                                        "cases": {k: v for k, v in cases.items()}}, indent=1))
    print("qwen golden:", {k: v.shape for k, v in out.items()})


def gen_qwen_rep():
    """HF's own greedy generation of the tiny Qwen2-VL with `repetition_penalty` set on the model's GENERATION CONFIG (1.3) and the
    reference's argument list (/root/reference/src/models/_qwen2_vl.py:319-329: do_sample = temperature > 0, temperature, top_p,
    num_beams, max_new_tokens - no repetition_penalty argument), i.e. the situation of a checkpoint whose generation_config.json
    carries the field: tokens, RAW logits (`output_logits`) and PROCESSED scores (`output_scores`: after the penalty) of 12 steps."""
    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, 1234)
    out = {}
    # prompt seeds chosen (scan of 11..39 with HF fp32) so that the penalty DECIDES tokens: without it the seeded model loops
    # (seed 16: 394 394 394 ...; seed 18: 481 432 481 489 ...), with it it does not
    cases = {"s13": ([(1, 4, 4)], 13), "s16": ([(1, 4, 4)], 16), "s18": ([(1, 4, 4)], 18), "b12": ([(1, 6, 4), (1, 4, 8)], 12)}
    for name, (grid, pseed) in cases.items():
        pix = recipes.pixel_values(grid, seed=7)
        ids = recipes.prompt_ids(cfg, grid, seed=pseed)
        for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            m = hf_qwen(cfg, w, dtype)
            m.generation_config.repetition_penalty = 1.3
            g = torch.tensor(grid)
            inp = torch.from_numpy(ids)[None]
            mm = (inp == cfg.image_token_id).int()
            with torch.no_grad():
                gen = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype),
                                 image_grid_thw=g, mm_token_type_ids=mm, do_sample=False, temperature=0, top_p=None, num_beams=1,
                                 max_new_tokens=12, use_cache=True, eos_token_id=None, pad_token_id=0, output_logits=True,
                                 output_scores=True, return_dict_in_generate=True)
                plain = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype),
                                   image_grid_thw=g, mm_token_type_ids=mm, do_sample=False, temperature=0, top_p=None, num_beams=1,
                                   max_new_tokens=12, use_cache=True, eos_token_id=None, pad_token_id=0, repetition_penalty=1.0)
            out[f"{name}_{tag}_tokens"] = gen.sequences[0, inp.shape[1]:].numpy()
            out[f"{name}_{tag}_logits"] = torch.stack([l[0] for l in gen.logits]).float().numpy()
            out[f"{name}_{tag}_scores"] = torch.stack([l[0] for l in gen.scores]).float().numpy()
            out[f"{name}_{tag}_tokens_without_penalty"] = plain[0, inp.shape[1]:].numpy()
        out[f"{name}_grid"] = np.array(grid)
        out[f"{name}_ids"] = ids
    np.savez_compressed(GOLD / "qwen2vl_tiny_rep.npz", **out)
    (GOLD / "qwen2vl_tiny_rep.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234, "repetition_penalty": 1.3,
                                                            "set_on": "model.generation_config (not passed to generate)",
                                                            "cases": {k: {"grid": v[0], "prompt_seed": v[1]} for k, v in cases.items()}}, indent=1))
    for name in cases:
        print("qwen rep golden", name, out[f"{name}_f32_tokens"].tolist(), "without:", out[f"{name}_f32_tokens_without_penalty"].tolist())


def hf_qwen25(cfg, weights, dtype):
    from transformers import Qwen2_5_VLConfig, Qwen2_5_VLForConditionalGeneration

    v, t = cfg.vision, cfg.text
    hcfg = Qwen2_5_VLConfig(
        text_config=dict(hidden_size=t.hidden_size, num_hidden_layers=t.num_hidden_layers,
                         num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                         intermediate_size=t.intermediate_size, vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps,
                         max_position_embeddings=4096, tie_word_embeddings=t.tie_word_embeddings,
                         rope_parameters=dict(rope_type="default", rope_theta=t.rope_theta, mrope_section=list(t.mrope_section))),
        vision_config=dict(depth=v.depth, hidden_size=v.embed_dim, num_heads=v.num_heads, out_hidden_size=v.hidden_size,
                           intermediate_size=v.intermediate_size, patch_size=v.patch_size, spatial_merge_size=v.spatial_merge_size,
                           temporal_patch_size=v.temporal_patch_size, window_size=v.window_size,
                           fullatt_block_indexes=list(v.fullatt_block_indexes), hidden_act="silu"),
        image_token_id=cfg.image_token_id, video_token_id=cfg.image_token_id + 1,
        vision_start_token_id=cfg.image_token_id + 2, vision_end_token_id=cfg.image_token_id + 3,
        tie_word_embeddings=t.tie_word_embeddings)
    hcfg._attn_implementation = "eager"
    m = Qwen2_5_VLForConditionalGeneration(hcfg)
    sd = {k: torch.from_numpy(a.copy()) for k, a in weights.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("lm_head" in k or "inv_freq" in k for k in missing), (missing, unexpected)
    return m.to(dtype).eval()


def gen_qwen25():
    """Qwen2.5-VL (the reference's `Qwen2_5_VLForConditionalGeneration` branch, src/models/_qwen2_vl.py:106-115): vision tower with
    window attention (grids that leave ragged border windows, a window-multiple side, several images), per-step logits, tokens."""
    cfg = recipes.tiny_cfg25()
    w = recipes.qwen25vl_weights(cfg, 1234)
    out = {}
    cases = {"a": [(1, 12, 20)], "b": [(1, 6, 4), (1, 16, 8), (1, 10, 18)]}   # 112 px = 8 patches = 4 merged groups per window side
    for name, grid in cases.items():
        pix = recipes.pixel_values(grid, seed=7)
        ids = recipes.prompt_ids(cfg, grid, seed=11)
        for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
            m = hf_qwen25(cfg, w, dtype)
            g = torch.tensor(grid)
            inp = torch.from_numpy(ids)[None]
            mm = (inp == cfg.image_token_id).int()
            with torch.no_grad():
                vis = m.model.visual(torch.from_numpy(pix).to(dtype), grid_thw=g).pooler_output
                gen = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype),
                                 image_grid_thw=g, mm_token_type_ids=mm, do_sample=False, num_beams=1, max_new_tokens=8,
                                 use_cache=True, eos_token_id=None, pad_token_id=0, output_logits=True,
                                 return_dict_in_generate=True)
                pos, delta = m.model.get_rope_index(inp, mm_token_type_ids=mm, image_grid_thw=g)
            out[f"{name}_{tag}_vit"] = torch.cat(list(vis), 0).float().numpy() if isinstance(vis, (list, tuple)) else vis.float().numpy()
            out[f"{name}_{tag}_tokens"] = gen.sequences[0, inp.shape[1]:].numpy()
            out[f"{name}_{tag}_logits"] = torch.stack([l[0] for l in gen.logits]).float().numpy()
            out[f"{name}_pos3"] = pos[:, 0].numpy()
            out[f"{name}_delta"] = np.array(int(delta[0, 0]))
        out[f"{name}_grid"] = np.array(grid)
        out[f"{name}_ids"] = ids
    # the integer window bookkeeping at real geometry (448 x 448 and ragged sizes): window_index + cu_window_seqlens from HF
    from transformers import vision_utils as vu

    geo = {}
    for gname, grid in {"448": [[1, 32, 32]], "ragged": [[1, 36, 36], [1, 26, 48], [1, 2, 2], [1, 64, 64], [1, 14, 70]]}.items():
        wi, cu = vu.get_vision_window_index(torch.tensor(grid), spatial_merge_size=2, window_size=112, patch_size=14)
        geo[gname] = {"grid": grid, "window_index_crc": int(zlib.crc32(wi.numpy().astype(np.int64).tobytes())), "n": int(wi.numel()),
                      "cu_window_seqlens": cu.tolist()}
    np.savez_compressed(GOLD / "qwen25vl_tiny.npz", **out)
    (GOLD / "qwen25vl_tiny.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234, "cases": cases, "window_geometry": geo}, indent=1))
    print("qwen2.5 golden:", {k: v.shape for k, v in out.items()})


def Cfg448():
    """get_rope_index on a 286-token prompt holding one 32x32-patch image (16x16 merged tokens)."""
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    cfg = recipes.tiny_cfg()
    m = hf_qwen(cfg, recipes.qwen2vl_weights(cfg, 1234), torch.float32)
    r = np.random.default_rng(5)
    ids = np.concatenate([r.integers(1, 400, 14), np.full(256, cfg.image_token_id), r.integers(1, 400, 16)]).astype(np.int64)
    inp = torch.from_numpy(ids)[None]
    pos, delta = m.model.get_rope_index(inp, mm_token_type_ids=(inp == cfg.image_token_id).int(),
                                        image_grid_thw=torch.tensor([[1, 32, 32]]))
    return ids, pos[:, 0].numpy(), np.array(int(delta[0, 0]))


# ------------------------------------------------------------------ the reference's scorer
class _Anything:
    """Permissive stand-in for symbols of packages that are absent offline (never executed on the metric path)."""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()

    def __or__(self, other):
        return self

    __ror__ = __and__ = __rand__ = __or__


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything


def import_reference():
    for name in ("gdown", "pytablewriter", "dotenv", "tenacity", "wandb", "spacy", "sacrebleu", "sacrebleu.metrics"):
        if name in sys.modules:
            continue
        mod = _Stub(name)
        mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
        mod.__path__ = []
        sys.modules[name] = mod
    ten = sys.modules["tenacity"]
    ten.retry = lambda *a, **k: (lambda f: f)
    sys.modules["dotenv"].load_dotenv = lambda *a, **k: None
    sys.path.insert(0, str(REF))
    import datasets.arrow_dataset as ad

    col = getattr(ad, "Column", None)
    if col is not None and not hasattr(col, "unsqueeze"):  # datasets>=4 returns Column at _group.py:537
        col.unsqueeze = lambda s, d: torch.stack([torch.as_tensor(x) for x in s]).unsqueeze(d)
    if col is not None and not getattr(ad.Dataset, "_owc_v3_getitem", False):
        # datasets 3.1 (the reference's pin) returns the materialised column for `ds["name"]`: a stacked tensor under the
        # torch format when the rows stack, else a list; datasets >= 4 returns a lazy Column, which torch.mean etc. reject
        # (_group.py:318-330).  Restore the pinned behaviour.
        orig = ad.Dataset.__getitem__

        def getitem(self, key):
            out = orig(self, key)
            if isinstance(out, col):
                rows = list(out)
                if rows and all(isinstance(r, torch.Tensor) for r in rows) and len({tuple(r.shape) for r in rows}) == 1:
                    return torch.stack(rows)
                return rows
            return out

        ad.Dataset.__getitem__ = getitem
        ad.Dataset._owc_v3_getitem = True
    import src.data.metrics as metrics  # noqa: F401
    import src.data.pipelines.text._text as text_mod
    import src.utils as utils

    return metrics, text_mod, utils


def gen_scorer_ragged():
    """The reference's own encode_sentence_bert / semantic_similarity on 256 RAGGED labels at full MiniLM-L6 size (lengths uniform
    2..16, mixed `str` / `[str]` wrapping): the 8-label `minilm` case of gen_scorer() is one padded batch; this one is what a task's
    prediction column looks like.  Written to its own file (scorer_minilm256.npz) so the existing vectors stay byte-identical."""
    metrics, text_mod, _ = import_reference()
    c = recipes.bert_cfg("minilm")
    w = recipes.bert_weights(c, 1234)
    text_mod.sentence_bert_model = hf_bert(c, w)
    text_mod.sentence_bert_processor = IdTokenizer()
    n, L = 256, 16
    ids_r, mask_r = recipes.label_tokens(n, L, c["vocab_size"], seed=31)
    ids_p, mask_p = recipes.label_tokens(n, L, c["vocab_size"], seed=32)
    to_text = lambda ids, mask: [" ".join(str(int(t)) for t, m in zip(r, mk) if m) for r, mk in zip(ids, mask)]  # noqa: E731
    refs, preds = to_text(ids_r, mask_r), to_text(ids_p, mask_p)
    out = {}
    saved = torch.cuda.is_available
    torch.cuda.is_available = lambda: False   # the reference's CPU fp32 branch (_text.py:165-170)
    try:
        b = text_mod.encode_sentence_bert({"text": list(refs)}, input_column="text")
        out["ref_embeds"] = np.array(b["text_sentence_bert_embeds"], dtype=np.float32)
        b = text_mod.encode_sentence_bert({"text": list(preds)}, input_column="text")
        out["pred_embeds"] = np.array(b["text_sentence_bert_embeds"], dtype=np.float32)
        items = [(r, [p]) if i % 2 else ([r], p) for i, (r, p) in enumerate(zip(refs, preds))]
        ss = metrics.get_metric_info("semantic_similarity")
        out["semantic_similarity_none"] = np.array(ss.group_fn(ss.builder_fn(items), reduce="none"), dtype=np.float32)
        out["semantic_similarity_mean"] = np.array(ss.group_fn(ss.builder_fn(items), reduce="mean"), dtype=np.float32)
        ma = metrics.get_metric_info("mean_average_semantic_similarity")
        mean_average = ma.group_fn(ma.builder_fn(items), reduce="mean")   # dict: threshold -> mass, + their mean
    finally:
        torch.cuda.is_available = saved
    out["label_seeds"] = np.array([31, 32])
    np.savez_compressed(GOLD / "scorer_minilm256.npz", **out)
    (GOLD / "scorer_minilm256.json").write_text(json.dumps({"versions": versions(), "n": n, "L": L, "weights_seed": 1234,
                                                            "mean_average": mean_average,
                                                            "what": "reference encode_sentence_bert + semantic_similarity, CPU fp32"}, indent=1))
    print("scorer ragged golden:", {k: v.shape for k, v in out.items()})


def gen_scorer_mpnet():
    """BASELINE.json configs[0] names all-mpnet-base-v2: the reference's own encode_sentence_bert (CPU fp32 branch) around HF's
    MPNetModel on seeded weights - a tiny config and the full all-mpnet-base-v2 size - with ragged labels padded by MPNet's pad id (1)."""
    from transformers import MPNetConfig, MPNetModel

    _, text_mod, _ = import_reference()
    out, meta = {}, {"versions": versions()}
    for kind, n, L in (("tiny", 24, 12), ("base", 48, 16)):
        c = recipes.mpnet_cfg(kind)
        w = recipes.mpnet_weights(c, 1234)
        m = MPNetModel(MPNetConfig(**c), add_pooling_layer=False)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in w.items()}, strict=False)
        assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
        text_mod.sentence_bert_model = m.eval()
        text_mod.sentence_bert_processor = IdTokenizer(pad=1)
        ids, mask = recipes.mpnet_label_tokens(n, L, c["vocab_size"], seed=41)
        texts = [" ".join(str(int(t)) for t, mk in zip(r, mr) if mk) for r, mr in zip(ids, mask)]
        saved = torch.cuda.is_available
        torch.cuda.is_available = lambda: False   # the reference's CPU fp32 branch (_text.py:165-170)
        try:
            b = text_mod.encode_sentence_bert({"text": list(texts)}, input_column="text")
        finally:
            torch.cuda.is_available = saved
        out[f"{kind}_embeds"] = np.array(b["text_sentence_bert_embeds"], dtype=np.float32)
        with torch.no_grad():
            out[f"{kind}_hidden0"] = m(input_ids=torch.from_numpy(ids[:2]), attention_mask=torch.from_numpy(mask[:2]))[0].numpy()
        meta[kind] = {"n": n, "L": L, "label_seed": 41, "weights_seed": 1234, "cfg": c}
    text_mod.sentence_bert_model = text_mod.sentence_bert_processor = None
    np.savez_compressed(GOLD / "scorer_mpnet.npz", **out)
    (GOLD / "scorer_mpnet.json").write_text(json.dumps({**meta, "what": "reference encode_sentence_bert (CPU fp32) around HF MPNetModel, seeded weights"}, indent=1))
    print("scorer mpnet golden:", {k: v.shape for k, v in out.items()})


class IdTokenizer:
    """Stand-in for AutoTokenizer: a label is a string of space-separated token ids (no tokenizer files offline)."""

    def __init__(self, pad: int = 0):
        self.pad = pad

    def __call__(self, text, padding=True, truncation=True, return_tensors="pt"):
        rows = [[int(t) for t in s.split()] for s in text]
        L = max(len(r) for r in rows)
        ids = torch.tensor([r + [self.pad] * (L - len(r)) for r in rows])
        mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows])

        class Enc(dict):
            def to(self, *a, **k):
                return self

        return Enc(input_ids=ids, attention_mask=mask)


def hf_bert(c, w):
    from transformers import BertConfig, BertModel

    m = BertModel(BertConfig(**c), add_pooling_layer=False)
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in w.items()}, strict=False)
    assert not unexpected and not [k for k in missing if "position_ids" not in k], (missing, unexpected)
    return m.eval()


def gen_scorer():
    metrics, text_mod, utils = import_reference()
    out, meta = {}, {"versions": versions()}
    for kind, n, L in (("tiny", 24, 12), ("minilm", 8, 16)):
        c = recipes.bert_cfg(kind)
        w = recipes.bert_weights(c, 1234)
        text_mod.sentence_bert_model = hf_bert(c, w)
        text_mod.sentence_bert_processor = IdTokenizer()
        ids_r, mask_r = recipes.label_tokens(n, L, c["vocab_size"], seed=21)
        ids_p, mask_p = recipes.label_tokens(n, L, c["vocab_size"], seed=22)
        to_text = lambda ids, mask: [" ".join(str(int(t)) for t, m in zip(r, mk) if m) for r, mk in zip(ids, mask)]  # noqa: E731
        refs, preds = to_text(ids_r, mask_r), to_text(ids_p, mask_p)
        # the reference's own encode_sentence_bert (CPU fp32 branch, _text.py:165-170)
        saved = torch.cuda.is_available
        torch.cuda.is_available = lambda: False
        try:
            b = text_mod.encode_sentence_bert({"text": list(refs)}, input_column="text")
            out[f"{kind}_ref_embeds"] = np.array(b["text_sentence_bert_embeds"], dtype=np.float32)
            b = text_mod.encode_sentence_bert({"text": list(preds)}, input_column="text")
            out[f"{kind}_pred_embeds"] = np.array(b["text_sentence_bert_embeds"], dtype=np.float32)
            items = [(r, [p]) if i % 2 else ([r], p) for i, (r, p) in enumerate(zip(refs, preds))]
            ss = metrics.get_metric_info("semantic_similarity")
            out[f"{kind}_semantic_similarity_none"] = np.array(ss.group_fn(ss.builder_fn(items), reduce="none"), dtype=np.float32)
            out[f"{kind}_semantic_similarity_mean"] = np.array(ss.group_fn(ss.builder_fn(items), reduce="mean"), dtype=np.float32)
            if kind == "tiny":
                ma = metrics.get_metric_info("mean_average_semantic_similarity")
                meta["tiny_mean_average"] = ma.group_fn(ma.builder_fn(items), reduce="mean")
        finally:
            torch.cuda.is_available = saved
    # string metrics + host helpers (exact expected values from the reference's own functions)
    em = metrics.get_metric_info("exact_match").builder_fn
    ti = metrics.get_metric_info("textual_inclusion").builder_fn
    pairs = [("A dog", "a dog"), ("a, dog", "a dog"), ("$5 bill", "5 bill"), ("cat", "a photo of a cat"),
             ("Golden Retriever.", "golden retriever"), ("", "x"), ("sea  lion", "sea lion"), (" tabby cat ", "Tabby Cat")]
    meta["string_metrics"] = [
        {"pred": p, "ref": r,
         "exact_match": float(em(predictions=[p], references=[r], ignore_case=True, regexes_to_ignore=[",", "\\$"])["exact_match"]),
         "exact_match_plain": float(em(predictions=[p], references=[r])["exact_match"]),
         "textual_inclusion": float(ti(predictions=[p], references=[r])["textual_inclusion"])} for p, r in pairs]
    meta["create_iterator"] = {f"{w_}_{lim}": [[i for i, _ in utils.create_iterator(enumerate(range(10)), r, w_, lim)] for r in range(w_)]
                               for w_ in (1, 2, 4, 8) for lim in (None, 9)}
    meta["parse_string_args"] = {s: utils.parse_string_args(s) for s in
                                 ("", "a=1,b=true,c=False,d=0.5,e=hello", "max_pixels=802816,use_flash_attention_2=false", "x=1e-3,y=-2")}
    data = [("ctx b", {"max_new_tokens": 64, "until": ["\n"]}), ("ctx aaaa", {"max_new_tokens": 64, "until": ["\n"]}),
            ("c", {"max_new_tokens": 16}), ("ctx cc", {"max_new_tokens": 64, "until": ["\n"]})]
    col = utils.Collator(data, lambda x: (-len(x[0]), x[0]), grouping=True)
    batches = [list(b) for b in col.get_batched(n=2, batch_fn=None)]
    meta["collator"] = {"batches": [[x[0] for x in b] for b in batches],
                        "restored": col.get_original([x[0].upper() for b in batches for x in b])}
    mean = metrics.AGGREGATIONS["mean"].builder_fn if hasattr(metrics, "AGGREGATIONS") else None
    if mean:
        meta["mean"] = mean([0.0, 1.0, 1.0, 0.5])
    np.savez_compressed(GOLD / "scorer.npz", **out)
    (GOLD / "scorer.json").write_text(json.dumps(meta, indent=1))
    print("scorer golden:", {k: v.shape for k, v in out.items()})


def gen_llava():
    """HF LlavaForConditionalGeneration (CLIP tower + projector + Llama decoder) on a tiny config, fp32 and bf16."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration

    cfg = recipes.tiny_llava_cfg()
    w = recipes.llava_weights(cfg, 1234)
    v, t = cfg.vision, cfg.text
    hcfg = LlavaConfig(
        vision_config=CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                                       num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                                       image_size=v.image_size, patch_size=v.patch_size, hidden_act="quick_gelu",
                                       layer_norm_eps=v.layer_norm_eps, projection_dim=64),
        text_config=LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                                vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps, max_position_embeddings=1024,
                                rope_theta=t.rope_theta, tie_word_embeddings=False),
        image_token_id=cfg.image_token_id, vision_feature_layer=cfg.vision_feature_layer,
        vision_feature_select_strategy="default", projector_hidden_act="gelu", image_seq_length=16)
    hcfg._attn_implementation = "eager"
    out = {}
    r = np.random.default_rng(17)
    pix = recipes.clip_pixels(2, v.image_size)
    ids = np.concatenate([r.integers(1, 400, 6), np.full(16, cfg.image_token_id), r.integers(1, 400, 3),
                          np.full(16, cfg.image_token_id), r.integers(1, 400, 7)]).astype(np.int64)
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = LlavaForConditionalGeneration(hcfg)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in w.items()}, strict=False)
        assert not unexpected and all("post_layernorm" in k or "position_ids" in k for k in missing), (missing, unexpected)
        m = m.to(dtype).eval()
        inp = torch.from_numpy(ids)[None]
        with torch.no_grad():
            feats = m.get_image_features(torch.from_numpy(pix).to(dtype), vision_feature_layer=cfg.vision_feature_layer,
                                         vision_feature_select_strategy="default").pooler_output
            gen = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype),
                             do_sample=False, num_beams=1, max_new_tokens=8, use_cache=True, eos_token_id=None, pad_token_id=0,
                             output_logits=True, return_dict_in_generate=True)
        out[f"{tag}_feats"] = torch.cat(list(feats), 0).float().numpy()
        out[f"{tag}_tokens"] = gen.sequences[0, inp.shape[1]:].numpy()
        out[f"{tag}_logits"] = torch.stack([l[0] for l in gen.logits]).float().numpy()
    out["ids"] = ids
    np.savez_compressed(GOLD / "llava_tiny.npz", **out)
    (GOLD / "llava_tiny.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234}, indent=1))
    print("llava golden:", {k: v.shape for k, v in out.items()})


NEXT_SIZES = [(90, 160), (200, 100)]  # (h, w) of the two synthetic images: a wide one and a tall one


def gen_llava_next():
    """HF LlavaNextForConditionalGeneration (anyres tiling + unpad + image_newline) on a tiny config, fp32 and bf16;
    plus the feature ORDER pack_image_features produces for a sweep of image sizes (integer golden)."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaNextConfig, LlavaNextForConditionalGeneration
    from transformers.models.llava_next.modeling_llava_next import image_size_to_num_patches

    cfg = recipes.tiny_llava_next_cfg()
    w = recipes.llava_weights(cfg, 1234)
    v, t = cfg.vision, cfg.text
    pin = [list(p) for p in cfg.image_grid_pinpoints]
    hcfg = LlavaNextConfig(
        vision_config=CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                                       num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                                       image_size=v.image_size, patch_size=v.patch_size, hidden_act="quick_gelu",
                                       layer_norm_eps=v.layer_norm_eps, projection_dim=64),
        text_config=LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                                vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps, max_position_embeddings=1024,
                                rope_theta=t.rope_theta, tie_word_embeddings=False),
        image_token_id=cfg.image_token_id, vision_feature_layer=cfg.vision_feature_layer,
        vision_feature_select_strategy="default", projector_hidden_act="gelu", image_grid_pinpoints=pin, image_seq_length=16)
    hcfg._attn_implementation = "eager"
    out = {}
    S = v.image_size
    views = [image_size_to_num_patches(list(sz), pin, S) for sz in NEXT_SIZES]
    pix = recipes.clip_pixels(sum(views), S, seed=41)
    padded = np.zeros((len(views), max(views), 3, S, S), np.float32)
    v0 = 0
    for i, nv in enumerate(views):
        padded[i, :nv] = pix[v0:v0 + nv]
        v0 += nv
    r = np.random.default_rng(19)
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = LlavaNextForConditionalGeneration(hcfg)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in w.items()}, strict=False)
        assert not unexpected and all("post_layernorm" in k or "position_ids" in k for k in missing), (missing, unexpected)
        m = m.to(dtype).eval()
        with torch.no_grad():
            feats = m.get_image_features(torch.from_numpy(padded).to(dtype), torch.tensor(NEXT_SIZES),
                                         vision_feature_layer=cfg.vision_feature_layer,
                                         vision_feature_select_strategy="default").pooler_output
            n_tok = [int(f.shape[0]) for f in feats]
            if "ids" not in out:
                out["ids"] = np.concatenate([r.integers(1, 400, 5), np.full(n_tok[0], cfg.image_token_id), r.integers(1, 400, 2),
                                             np.full(n_tok[1], cfg.image_token_id), r.integers(1, 400, 6)]).astype(np.int64)
            inp = torch.from_numpy(out["ids"])[None]
            gen = m.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(padded).to(dtype),
                             image_sizes=torch.tensor(NEXT_SIZES), do_sample=False, num_beams=1, max_new_tokens=8, use_cache=True,
                             eos_token_id=None, pad_token_id=0, output_logits=True, return_dict_in_generate=True)
        out[f"{tag}_feats"] = torch.cat(list(feats), 0).float().numpy()
        out[f"{tag}_tokens"] = gen.sequences[0, inp.shape[1]:].numpy()
        out[f"{tag}_logits"] = torch.stack([l[0] for l in gen.logits]).float().numpy()
    out["views"] = np.array(views)
    out["n_tok"] = np.array(n_tok)
    out["image_sizes"] = np.array(NEXT_SIZES)
    # integer golden: order in which pack_image_features reads (view, patch) for many sizes, real geometry (g = 24)
    from transformers import LlavaNextProcessor  # noqa: F401  (import check only)

    real = LlavaNextConfig()
    real_pin = [[336, 672], [672, 336], [672, 672], [1008, 336], [336, 1008]]
    mm = LlavaNextForConditionalGeneration(hcfg).model  # only used for its pack_image_features method
    mm.config.vision_config.image_size, mm.config.vision_config.patch_size, mm.config.image_grid_pinpoints = 336, 14, real_pin
    order = {}
    for (h, w_) in [(480, 640), (640, 480), (336, 336), (500, 500), (1000, 300), (300, 1000), (333, 999), (768, 1024), (37, 1200), (1080, 1920)]:
        nv = image_size_to_num_patches([h, w_], real_pin, 336)
        f = (torch.arange(nv)[:, None] * 577 + 1 + torch.arange(576)[None]).float()[..., None]  # value = row in a view-major [nv*577] buffer
        packed, _ = mm.pack_image_features([f], torch.tensor([[h, w_]]), "default", image_newline=torch.tensor([-1.0]))
        idx = packed[0][:, 0].long().numpy()
        order[f"{h}x{w_}"] = {"views": int(nv), "n": int(len(idx)), "crc": int(zlib.crc32(idx.astype(np.int64).tobytes())),
                              "head": idx[576:576 + 30].tolist()}
    np.savez_compressed(GOLD / "llava_next_tiny.npz", **out)
    (GOLD / "llava_next_tiny.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234, "pack_order_real_geometry": order}, indent=1))
    print("llava-next golden:", {k: v.shape for k, v in out.items()}, views, n_tok)


def gradient_image(h=300, w=450):
    """Deterministic RGB test image (no dataset offline): three linear ramps."""
    yy, xx = np.mgrid[0:h, 0:w]
    return np.stack([(xx * 255 // (w - 1)), (yy * 255 // (h - 1)), ((xx + yy) * 255 // (h + w - 2))], -1).astype(np.uint8)


def gen_image():
    """G6: HF Qwen2VLImageProcessor (PIL backend) on a 450x300 image: smart_resize + bicubic + normalise + patchify."""
    from PIL import Image
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil

    proc = Qwen2VLImageProcessorPil(min_pixels=4 * 28 * 28, max_pixels=1024 * 28 * 28)  # the wrapper's defaults (_qwen2_vl.py:64-65)
    out = proc(images=[Image.fromarray(gradient_image(), "RGB")], return_tensors="np")
    pv = out["pixel_values"].astype(np.float32)
    np.savez_compressed(GOLD / "image_proc.npz", grid=out["image_grid_thw"], sample=pv[::37, ::29], row_sums=pv.sum(1),
                        first_rows=pv[:2], shape=np.array(pv.shape))
    (GOLD / "image_proc.json").write_text(json.dumps({"versions": versions(), "image": "gradient_image(300, 450)"}, indent=1))
    print("image golden:", pv.shape, out["image_grid_thw"].tolist())


def gen_ranking():
    """Runs the reference's own eval_ranking.main (criterion semantic_similarity, CPU fp32 BERT with seeded weights) on
    synthetic runs and records what it printed plus the per-game outcomes."""
    import contextlib
    import io
    import tempfile
    from argparse import Namespace

    metrics, text_mod, utils = import_reference()
    c = recipes.bert_cfg("tiny")
    text_mod.sentence_bert_model = hf_bert(c, recipes.bert_weights(c, 1234))
    text_mod.sentence_bert_processor = IdTokenizer()
    import src.data.pipelines.text as text_pkg

    text_pkg.encode_sentence_bert = text_mod.encode_sentence_bert
    spec = importlib.util.spec_from_file_location("ref_eval_ranking", str(REF / "eval_ranking.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    captured = {}
    orig_counter = mod.Counter

    def spy_counter(x=()):
        x = list(x)
        if x and all(v in (0, 0.5, 1, 0.0, 1.0) for v in x):
            captured["scores"] = [float(v) for v in x]
        return orig_counter(x)

    mod.Counter = spy_counter
    saved = torch.cuda.is_available
    torch.cuda.is_available = lambda: False
    out = {}
    try:
        with tempfile.TemporaryDirectory() as td:
            recipes.ranking_runs(Path(td))
            for tag, kw in {"default": {}, "no_zero_sum": {"disable_zero_sum": True, "k_factor": 32}}.items():
                args = Namespace(input=td, criterion="semantic_similarity", initial_rating=1000, k_factor=kw.get("k_factor", 16),
                                 num_rounds=10, num_samples=200, disable_zero_sum=kw.get("disable_zero_sum", False), seed=1234,
                                 log_level="WARNING")
                buf = io.StringIO()
                with contextlib.redirect_stdout(buf):
                    mod.main(args)
                out[tag] = {"stdout": buf.getvalue(), "scores": captured["scores"]}
    finally:
        torch.cuda.is_available = saved
    (GOLD / "ranking.json").write_text(json.dumps({"source": "eval_ranking.py main() of the reference, datasets " + __import__("datasets").__version__,
                                                    "versions": versions(), "cases": out}, indent=1))
    print("ranking golden:\n" + out["default"]["stdout"])


PROMPT_CASES = [
    [{"role": "user", "content": "<image>\nWhat type of object is in this image?"}],
    [{"role": "user", "content": "<image> <image>\nCompare."}, {"role": "assistant", "content": "Both are cats."},
     {"role": "user", "content": "Which breed?"}],
    [{"role": "user", "content": "no image here"}],
]


def gen_llava_loglik():
    """The arithmetic of the reference's LLaVA.loglikelihood (/root/reference/src/models/_llava_hf.py:229-252) run on HF's tiny LLaVA:
    labels = input_ids with the first `n_ctx` positions masked (n_ctx = length of the prompt tokenised WITHOUT image expansion, so
    image positions stay in the loss), outputs = model(**inputs, labels=labels) -> loss, and the UNSHIFTED comparison
    argmax(logits)[n_ctx:] == input_ids[n_ctx:]."""
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration

    cfg = recipes.tiny_llava_cfg()
    w = recipes.llava_weights(cfg, 1234)
    v, t = cfg.vision, cfg.text
    hcfg = LlavaConfig(
        vision_config=CLIPVisionConfig(hidden_size=v.hidden_size, intermediate_size=v.intermediate_size,
                                       num_hidden_layers=v.num_hidden_layers, num_attention_heads=v.num_attention_heads,
                                       image_size=v.image_size, patch_size=v.patch_size, hidden_act="quick_gelu",
                                       layer_norm_eps=v.layer_norm_eps, projection_dim=64),
        text_config=LlamaConfig(hidden_size=t.hidden_size, intermediate_size=t.intermediate_size, num_hidden_layers=t.num_hidden_layers,
                                num_attention_heads=t.num_attention_heads, num_key_value_heads=t.num_key_value_heads,
                                vocab_size=t.vocab_size, rms_norm_eps=t.rms_norm_eps, max_position_embeddings=1024,
                                rope_theta=t.rope_theta, tie_word_embeddings=False),
        image_token_id=cfg.image_token_id, vision_feature_layer=cfg.vision_feature_layer,
        vision_feature_select_strategy="default", projector_hidden_act="gelu", image_seq_length=16)
    hcfg._attn_implementation = "eager"
    r = np.random.default_rng(23)
    pix = recipes.clip_pixels(1, v.image_size)
    head, tail, cont = r.integers(1, 400, 5), r.integers(1, 400, 9), r.integers(1, 400, 6)
    ids = np.concatenate([head, np.full(16, cfg.image_token_id), tail, cont]).astype(np.int64)
    n_ctx = len(head) + 1 + len(tail)          # the un-expanded prompt: one <image> token
    out = {"ids": ids, "n_ctx": np.array(n_ctx), "pix": pix}
    for dtype, tag in ((torch.float32, "f32"), (torch.bfloat16, "bf16")):
        m = LlavaForConditionalGeneration(hcfg)
        missing, unexpected = m.load_state_dict({k: torch.from_numpy(a.copy()) for k, a in w.items()}, strict=False)
        assert not unexpected and all("post_layernorm" in k or "position_ids" in k for k in missing), (missing, unexpected)
        m = m.to(dtype).eval()
        inp = torch.from_numpy(ids)[None]
        labels = inp.clone()
        labels[:, :n_ctx] = -100
        with torch.inference_mode():
            o = m(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=torch.from_numpy(pix).to(dtype), labels=labels)
        greedy = o["logits"].argmax(dim=-1)[:, n_ctx:inp.shape[1]]
        out[f"{tag}_loss"] = np.array(float(o["loss"].item()))
        out[f"{tag}_greedy"] = greedy[0].numpy()
        out[f"{tag}_max_equal"] = np.array(bool((greedy == inp[:, n_ctx:]).all()))
        out[f"{tag}_logits"] = o["logits"][0, n_ctx - 1:].float().numpy()
    np.savez_compressed(GOLD / "llava_loglik_tiny.npz", **out)
    (GOLD / "llava_loglik_tiny.json").write_text(json.dumps({"versions": versions(), "weights_seed": 1234,
                                                             "source": "_llava_hf.py:229-252 on HF LlavaForConditionalGeneration"}, indent=1))
    print("llava loglik golden:", {k: (v.shape if v.ndim else v.item()) for k, v in out.items()})


def gen_llava_prompt():
    """Text the reference's fallback chat template renders (the template string is read from the reference at
    generation time, /root/reference/src/models/_llava_hf.py:23, and rendered the way HF apply_chat_template does)."""
    import ast

    from jinja2.sandbox import ImmutableSandboxedEnvironment

    src = Path("/root/reference/src/models/_llava_hf.py").read_text()
    node = next(n for n in ast.parse(src).body if isinstance(n, ast.Assign) and getattr(n.targets[0], "id", "") == "VICUNA_CHAT_TEMPLATE")
    template = ast.literal_eval(node.value)
    env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
    cases = []
    for msgs in PROMPT_CASES:
        for gen in (True, False):
            cases.append({"messages": msgs, "add_generation_prompt": gen, "eos_token": "</s>",
                          "text": env.from_string(template).render(messages=msgs, add_generation_prompt=gen, eos_token="</s>")})
    (GOLD / "llava_prompt.json").write_text(json.dumps({"source": "_llava_hf.py:23 rendered with jinja2 (trim_blocks, lstrip_blocks)", "cases": cases}, indent=1))
    print("llava prompt golden:", len(cases), "cases;", cases[0]["text"][:80])


def gen_llava_image():
    """HF CLIPImageProcessor (LLaVA-1.5) and LlavaNextImageProcessor (PIL backends) on the gradient image."""
    from PIL import Image
    from transformers.models.clip.image_processing_pil_clip import CLIPImageProcessorPil
    from transformers.models.llava_next.image_processing_pil_llava_next import LlavaNextImageProcessorPil

    out = {}
    kw = dict(size={"shortest_edge": 336}, crop_size={"height": 336, "width": 336}, resample=3)
    for tag, (h, w) in {"wide": (300, 450), "tall": (500, 220)}.items():
        im = Image.fromarray(gradient_image(h, w), "RGB")
        pv = CLIPImageProcessorPil(**kw)(images=[im], return_tensors="np")["pixel_values"][0].astype(np.float32)
        out[f"clip_{tag}_sample"], out[f"clip_{tag}_sums"] = pv[:, ::7, ::11], pv.sum(-1)
        o = LlavaNextImageProcessorPil(**kw)(images=[im], return_tensors="np")
        pv = o["pixel_values"][0].astype(np.float32)
        out[f"next_{tag}_sample"], out[f"next_{tag}_sums"], out[f"next_{tag}_size"] = pv[:, :, ::7, ::11], pv.sum(-1), np.asarray(o["image_sizes"][0])
    np.savez_compressed(GOLD / "llava_image_proc.npz", **out)
    print("llava image golden:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    GOLD.mkdir(parents=True, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["qwen", "scorer", "image", "llava", "llava_next", "llava_image", "llava_prompt", "ranking"]
    if "image" in which:
        gen_image()
    if "llava" in which:
        gen_llava()
    if "ranking" in which:
        gen_ranking()
    if "llava_prompt" in which:
        gen_llava_prompt()
    if "llava_loglik" in which:
        gen_llava_loglik()
    if "llava_image" in which:
        gen_llava_image()
    if "llava_next" in which:
        gen_llava_next()
    if "qwen" in which:
        gen_qwen()
    if "qwen_rep" in which:
        gen_qwen_rep()
    if "qwen25" in which:
        gen_qwen25()
    if "scorer" in which:
        gen_scorer()
    if "scorer_ragged" in which:
        gen_scorer_ragged()
    if "scorer_mpnet" in which:
        gen_scorer_mpnet()
