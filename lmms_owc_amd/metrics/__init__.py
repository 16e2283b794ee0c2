from . import _group  # noqa: F401  (aggregations must register before the metrics that name them)
from . import _instance  # noqa: F401
from ._api import (AGGREGATIONS, METRICS, get_aggregation_builder, get_metric_builder, get_metric_info, mean_stderr,
                   register_aggregation, register_metric)

__all__ = ["AGGREGATIONS", "METRICS", "get_aggregation_builder", "get_metric_builder", "get_metric_info", "mean_stderr",
           "register_aggregation", "register_metric"]
