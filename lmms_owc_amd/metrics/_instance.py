"""Per-instance metrics (mirror of /root/reference/src/data/metrics/_instance.py:315-360, :445-480):
exact_match, textual_inclusion (CPU string work) and the passthrough halves of the model-based metrics."""

from __future__ import annotations

import re
import string

import numpy as np

from ._api import register_metric


@register_metric(group_fn_name="mean", higher_is_better=True, output_types=["generate_until"], can_bootstrap=True)
def exact_match(predictions: list, references: list, regexes_to_ignore: list | None = None, ignore_case: bool = False,
                ignore_punctuation: bool = False, ignore_numbers: bool = False) -> dict:
    if regexes_to_ignore is not None:
        for s in regexes_to_ignore:
            predictions = np.array([re.sub(s, "", x) for x in predictions])
            references = np.array([re.sub(s, "", x) for x in references])
    else:
        predictions, references = np.asarray(predictions), np.asarray(references)
    if ignore_case:
        predictions, references = np.char.lower(predictions), np.char.lower(references)
    if ignore_punctuation:
        table = str.maketrans("", "", string.punctuation)
        predictions, references = np.char.translate(predictions, table=table), np.char.translate(references, table=table)
    if ignore_numbers:
        table = str.maketrans("", "", string.digits)
        predictions, references = np.char.translate(predictions, table=table), np.char.translate(references, table=table)
    return {"exact_match": np.mean(predictions == references)}


@register_metric(group_fn_name="mean", higher_is_better=True, output_types=["generate_until"], can_bootstrap=True)
def textual_inclusion(predictions: list, references: list) -> dict:
    scores = [ref.lower().strip() in pred.lower().strip() for ref, pred in zip(references, predictions, strict=True)]
    return {"textual_inclusion": np.mean(scores)}


@register_metric(group_fn_name="semantic_similarity", higher_is_better=True, output_types=["generate_until"])
def semantic_similarity(items: list) -> list:
    """Passthrough: the batched work happens in the aggregation of the same name."""
    return items


@register_metric(group_fn_name="mean_average_semantic_similarity", higher_is_better=True, output_types=["generate_until"])
def mean_average_semantic_similarity(items: list) -> list:
    return items


@register_metric(group_fn_name="concept_semantic_similarity", higher_is_better=True, output_types=["generate_until"])
def concept_semantic_similarity(items: list) -> list:
    return items
