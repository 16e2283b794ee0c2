"""Metric / aggregation registries (mirror of /root/reference/src/data/metrics/_api.py:101-109, :260-314)."""

from __future__ import annotations

import math
from collections.abc import Callable

from ..schema import AggregationInfo, MetricInfo

METRICS: dict[str, MetricInfo] = {}
AGGREGATIONS: dict[str, AggregationInfo] = {}


def register_aggregation(name: str | None = None, can_bootstrap: bool = False) -> Callable:
    def decorator(fn: Callable) -> Callable:
        key = name or fn.__name__.lower()
        AGGREGATIONS[key] = AggregationInfo(name=key, builder_fn=fn, can_bootstrap=can_bootstrap)
        return fn

    return decorator


def register_metric(name: str | None = None, group_fn_name: str | None = None, higher_is_better: bool | None = None,
                    output_types: list | None = None, can_bootstrap: bool = False) -> Callable:
    def decorator(fn: Callable) -> Callable:
        key = name or fn.__name__.lower()
        METRICS[key] = MetricInfo(name=key, higher_is_better=higher_is_better, builder_fn=fn,
                                  group_fn=AGGREGATIONS[group_fn_name].builder_fn, output_types=output_types or [],
                                  can_bootstrap=can_bootstrap)
        return fn

    return decorator


def get_metric_info(metric_id: str) -> MetricInfo:
    return METRICS[metric_id]


def get_metric_builder(metric_id: str) -> Callable:
    return METRICS[metric_id].builder_fn


def get_aggregation_builder(name: str) -> Callable:
    return AGGREGATIONS[name].builder_fn


def mean_stderr(arr: list) -> float:
    """Sample standard error of the mean (_api.py:117-137)."""
    mu = sum(arr) / len(arr)
    return math.sqrt(sum((x - mu) ** 2 for x in arr) / (len(arr) - 1)) / math.sqrt(len(arr))
