"""Registry value types (mirror of /root/reference/src/schema/_base.py:105-136, dataclasses instead of pydantic)."""

from __future__ import annotations

from collections.abc import Callable
from dataclasses import dataclass, field


@dataclass
class ModelInfo:
    name: str
    builder_fn: Callable


@dataclass
class AggregationInfo:
    name: str
    builder_fn: Callable
    can_bootstrap: bool = False


@dataclass
class MetricInfo:
    name: str
    builder_fn: Callable
    group_fn: Callable
    higher_is_better: bool | None = None
    output_types: list = field(default_factory=list)
    can_bootstrap: bool = False
