"""Group-level (aggregation) metrics: mean, semantic_similarity, mean_average_semantic_similarity.

Same call contract as /root/reference/src/data/metrics/_group.py:380-389, :392-458, :488-544 —
`fn(items: list[(ref, pred)], reduce="mean"|"none")`, `ref` str or [str], `pred` str or [..., str] —
but the two sentence columns are embedded by the HIP encoder and paired on the GPU: class names are
embedded once per DISTINCT string instead of once per sample, and nothing round-trips through Arrow.
"""

from __future__ import annotations

from typing import Literal

from ._api import register_aggregation


@register_aggregation("mean", can_bootstrap=True)
def mean(arr: list) -> float:
    return sum(arr) / len(arr)


def _unwrap(items: list) -> tuple[list[str], list[str]]:
    refs, preds = zip(*items, strict=True)
    refs = [r[0] if isinstance(r, list) else r for r in refs]
    preds = [p[-1] if isinstance(p, list) else p for p in preds]
    return list(refs), list(preds)


def _paired_cosine(items: list):
    """cos_i = <embed(ref_i), embed(pred_i)> as a device fp32 tensor [N]."""
    from ..pipelines.text import embed_texts_unique, get_scorer

    refs, preds = _unwrap(items)
    scorer = get_scorer()
    ref_z = embed_texts_unique(refs)    # [N, D], distinct strings encoded once
    pred_z = embed_texts_unique(preds)
    return scorer.paired_cosine(ref_z, pred_z)


def _check_reduce(reduce: str, name: str) -> None:
    if reduce not in ("none", "mean"):
        raise ValueError(f'Unknown `reduce` value for `{name}` metric. Expected "none" or "mean", but got "{reduce}"')


@register_aggregation("semantic_similarity")
def semantic_similarity(items: list, reduce: Literal["none", "mean"] = "mean"):
    _check_reduce(reduce, "semantic_similarity")
    cos = _paired_cosine(items).cpu()
    return cos.mean().item() if reduce == "mean" else cos.tolist()


@register_aggregation("mean_average_semantic_similarity")
def mean_average_semantic_similarity(items: list, reduce: Literal["none", "mean"] = "mean"):
    _check_reduce(reduce, "mean_average_semantic_similarity")
    import torch

    cos = _paired_cosine(items).cpu()
    outputs: dict = {}
    if reduce == "mean":
        for thr in (0.5, 0.6, 0.7, 0.8, 0.9):
            outputs[f"semantic_similarity@{thr}"] = (cos >= thr).float().mean().item()
        outputs["semantic_similarity@avg"] = torch.tensor(list(outputs.values())).mean().item()
        return outputs
    for thr in (0.5, 0.6, 0.7, 0.8, 0.9):
        outputs[f"semantic_similarity@{thr}"] = (cos >= thr).int().tolist()
    outputs["semantic_similarity@avg"] = torch.tensor(list(outputs.values()), dtype=torch.float32).mean(dim=0).tolist()
    return outputs


SKIP_WORDS = [  # same groups as the reference (_group.py:207-236)
    "1", "2", "3", "4", "5", "6", "7", "8", "9", "10", "one", "two", "three", "four", "five", "six", "seven", "eight",
    "nine", "ten", "*", "a", "the", "image", "object", "photo", "type", "this photo", "it", "they", "them", "that",
    "this", "those", "which", "who", "whom", "whose", "where", "when", "what", "why", "how", "some",
]


@register_aggregation("concept_semantic_similarity")
def concept_semantic_similarity(items: list, reduce: Literal["none", "max", "mean", "median", "min"] = "max"):
    """Similarity between the ground truth and every concept (noun chunk) of the prediction, the whole prediction
    included; reduced per sample (max by default) and averaged.  Same contract as the reference
    (_group.py:176-334); concepts come from the pluggable extractor, the embedding of the UNIQUE strings and the
    pairing run on the GPU scorer."""
    import numpy as np
    import torch

    from ..pipelines.text import embed_texts_unique, extract_concepts, get_scorer

    if reduce not in ("none", "max", "mean", "median", "min"):
        raise ValueError(f'Unknown `reduce` value for `concept_semantic_similarity` metric. Expected "none", "max", '
                         f'"mean", "median", or "min", but got "{reduce}"')
    refs, preds = _unwrap(items)
    concepts = [c + [p] for c, p in zip(extract_concepts(preds, SKIP_WORDS), preds, strict=True)]
    flat_refs = [r for r, cs in zip(refs, concepts) for _ in cs]
    flat_concepts = [c for cs in concepts for c in cs]
    sims = get_scorer().paired_cosine(embed_texts_unique(flat_refs), embed_texts_unique(flat_concepts)).cpu().numpy()
    bounds = np.cumsum([0] + [len(cs) for cs in concepts])
    rows = [sims[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
    if reduce == "none":
        return list(zip(concepts, [r.tolist() for r in rows], strict=True))
    fn = {"max": np.max, "mean": np.mean, "median": lambda r: torch.from_numpy(r).median().item(), "min": np.min}[reduce]
    return float(np.mean([fn(r) for r in rows]))
