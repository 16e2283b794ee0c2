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
