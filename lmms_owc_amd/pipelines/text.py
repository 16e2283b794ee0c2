"""`encode_sentence_bert` on the MI355X scorer — same hook contract as
/root/reference/src/data/pipelines/text/_text.py:143-208:

    encode_sentence_bert(batch: dict[str, list[str]], rank=None, *, input_column="text",
                         output_column=f"{input_column}_sentence_bert_embeds") -> dict

adds L2-normalised embeddings (list[list[float]]) and is `datasets.map(batched=True)` compatible.
Module-level lazy singletons like the reference (`_text.py:10-15`): `sentence_bert_model` is a
`SentenceScorer` (HIP encoder), `sentence_bert_processor` a tokenizer callable.  Tokenisation stays
on the host; all arithmetic runs in libowc_hip.so (fp32, reference CPU-branch numerics).
"""

from __future__ import annotations

import os
from pathlib import Path

import numpy as np
import torch

sentence_bert_model = None       # lmms_owc_amd.engine.scorer.SentenceScorer
sentence_bert_processor = None   # callable(list[str], padding=True, truncation=True, return_tensors="np"|"pt")

MODEL_NAME = "sentence-transformers/all-MiniLM-L6-v2"


def set_sentence_bert(scorer, tokenizer) -> None:
    """Inject an encoder + tokenizer (tests; a BERT- or MPNet-architecture sentence encoder whose weights `BertWeights` takes:
    all-MiniLM-L6-v2, the reference's default, or all-mpnet-base-v2, the encoder BASELINE.json configs[0] names)."""
    global sentence_bert_model, sentence_bert_processor
    sentence_bert_model, sentence_bert_processor = scorer, tokenizer


def _load_default(rank: int | None):
    """Lazy load of all-MiniLM-L6-v2 from the local HF cache / a directory given by OWC_SENTENCE_BERT_PATH."""
    global sentence_bert_model, sentence_bert_processor
    from transformers import AutoConfig, AutoTokenizer

    from ..engine.scorer import BertWeights, SentenceScorer

    path = os.environ.get("OWC_SENTENCE_BERT_PATH", MODEL_NAME)
    sentence_bert_processor = AutoTokenizer.from_pretrained(path)
    cfg = AutoConfig.from_pretrained(path).to_dict()
    sd = _load_state_dict(path)
    sd = {k.split(".", 1)[1] if k.startswith(("bert.", "mpnet.")) else k: v for k, v in sd.items()}
    device = torch.device("cuda", (rank or 0) % max(torch.cuda.device_count(), 1))
    keys = ("vocab_size", "hidden_size", "num_hidden_layers", "num_attention_heads", "intermediate_size",
            "max_position_embeddings", "type_vocab_size", "layer_norm_eps")
    # (BertWeights tells BERT from MPNet - e.g. OWC_SENTENCE_BERT_PATH=sentence-transformers/all-mpnet-base-v2 - by the checkpoint's names)
    sentence_bert_model = SentenceScorer(BertWeights({k: cfg[k] for k in keys if k in cfg}, sd, device))


def _load_state_dict(path: str) -> dict:
    from safetensors.torch import load_file

    p = Path(path)
    if not p.is_dir():
        from huggingface_hub import snapshot_download

        p = Path(snapshot_download(path, allow_patterns=["*.safetensors", "*.json", "*.txt"]))
    files = sorted(p.glob("*.safetensors"))
    if not files:
        raise FileNotFoundError(f"no safetensors weights under {p}")
    sd: dict = {}
    for f in files:
        sd.update(load_file(str(f)))
    return sd


def get_scorer(rank: int | None = None):
    if sentence_bert_model is None:
        if not torch.cuda.is_available():
            raise RuntimeError("the sentence encoder runs in libowc_hip.so and needs an MI355X (no CPU fallback)")
        _load_default(rank)
    return sentence_bert_model


def _tokenize(texts: list[str]) -> tuple[np.ndarray, np.ndarray]:
    enc = sentence_bert_processor(texts, padding=True, truncation=True, return_tensors="np")
    return np.asarray(enc["input_ids"]), np.asarray(enc["attention_mask"])


def embed_texts(texts: list[str], batch_size: int = 4096) -> torch.Tensor:
    """[len(texts), D] fp32 device tensor; padding is per batch (pad-to-longest), like the reference."""
    scorer = get_scorer()
    outs = []
    for i in range(0, len(texts), batch_size):
        ids, mask = _tokenize(list(texts[i:i + batch_size]))
        outs.append(scorer.embed(ids, mask))
    return torch.cat(outs, 0) if len(outs) > 1 else outs[0]


def embed_texts_unique(texts: list[str]) -> torch.Tensor:
    """Embeds each DISTINCT string once (class names repeat N/C times) and gathers rows back per sample."""
    uniq = sorted(set(texts), key=lambda s: (len(s), s))  # length-sorted -> tight padding per batch
    index = {s: i for i, s in enumerate(uniq)}
    z = embed_texts(uniq)
    idx = torch.tensor([index[s] for s in texts], dtype=torch.long, device=z.device)
    return z.index_select(0, idx).contiguous()


# ---- concept extraction (reference: concept_extraction_spacy, _text.py:18-140) ----------------------------
# Two plug points: `set_concept_nlp(fn)` replaces ONLY the parser (spaCy `en_core_web_lg` in the reference: noun chunks +
# named entities per text) and keeps the reference's own post-processing below; `set_concept_extractor(fn)` replaces the
# whole step.  The CPU NLP model itself is outside the accelerated path.
concept_extractor = None  # callable(list[str], skip_words: list[str]) -> list[list[str]]
concept_nlp = None        # callable(list[str]) -> list[(noun_chunk_texts: list[str], entity_texts: list[str])]

_PREFIX_TERMS = ("a", "an", "the", "his", "her", "its", "their")   # articles + possessive pronouns (_text.py:61-67)


def set_concept_extractor(fn) -> None:
    global concept_extractor
    concept_extractor = fn


def set_concept_nlp(fn) -> None:
    global concept_nlp
    concept_nlp = fn


def _strip_prefix(concept: str) -> str:
    for term in _PREFIX_TERMS:
        if concept.startswith(term + " "):
            return concept[len(term) + 1:]
    return concept


def postprocess_concepts(noun_chunks: list[str], ents: list[str], skip_words: list[str], remove_prefix_words: bool = True) -> list[str]:
    """What `concept_extraction_spacy` does with spaCy's spans (_text.py:56-92), quirks included: everything is lower-cased;
    a leading article / possessive is dropped; noun chunks in `skip_words` are skipped but duplicates among noun chunks are
    KEPT; noun chunks are only recorded when `remove_prefix_words` is set (the append sits inside that branch); entities
    are added when not already present."""
    concepts: list[str] = []
    for text in noun_chunks:
        concept = text.lower()
        if remove_prefix_words:
            concept = _strip_prefix(concept)
            if concept in skip_words:
                continue
            concepts.append(concept)
    for text in ents:
        concept = text.lower()
        if remove_prefix_words:
            concept = _strip_prefix(concept)
            if concept in skip_words:
                continue
        if concept not in concepts:
            concepts.append(concept)
    return concepts


def _spacy_nlp():
    try:
        import spacy

        nlp = spacy.load("en_core_web_lg")
    except Exception as e:  # spaCy / the model are not installed offline
        raise RuntimeError("concept_semantic_similarity needs a parser: install spaCy + en_core_web_lg, or call "
                           "lmms_owc_amd.pipelines.text.set_concept_nlp(fn) / set_concept_extractor(fn)") from e
    return lambda texts: [([c.text for c in doc.noun_chunks], [e.text for e in doc.ents]) for doc in nlp.pipe(texts, batch_size=len(texts))]


def extract_concepts(texts: list[str], skip_words: list[str]) -> list[list[str]]:
    global concept_nlp
    if concept_extractor is not None:
        return concept_extractor(texts, skip_words)
    if concept_nlp is None:
        concept_nlp = _spacy_nlp()
    return [postprocess_concepts(chunks, ents, skip_words, True) for chunks, ents in concept_nlp(list(texts))]


def encode_sentence_bert(batch: dict, rank: int | None = None, **kwargs) -> dict:
    input_column = kwargs.pop("input_column", "text")
    output_column = kwargs.pop("output_column", f"{input_column}_sentence_bert_embeds")
    get_scorer(rank)
    if input_column not in batch:
        raise ValueError(f"{input_column} missing in dataset")
    if not isinstance(batch[input_column], list):
        raise NotImplementedError
    ids, mask = _tokenize(batch[input_column])
    batch[output_column] = sentence_bert_model.embed(ids, mask).cpu().numpy().tolist()
    return batch
