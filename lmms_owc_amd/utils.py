"""Host helpers the hot path's callers use (behavioural mirror of /root/reference/src/utils):
create_iterator (_core_utils.py:53-69), parse_string_args (:161-194), hash_string (:136-144),
Collator (_models_utils.py:8-148), sanitize_list, make_table (_data_utils.py:395-475, simplified)."""

from __future__ import annotations

import hashlib
from collections import defaultdict
from collections.abc import Callable, Iterable, Iterator
from itertools import islice


def create_iterator(raw_iterator: Iterator, rank: int | None, world_size: int | None, limit: int | None = None) -> Iterator:
    """THE shard function: rank r takes items r, r+W, r+2W, ... below `limit`."""
    return islice(raw_iterator, rank, limit, world_size)


def hash_string(string: str) -> str:
    return hashlib.sha256(string.encode("utf-8")).hexdigest()


def get_git_commit_hash() -> str | None:
    """`git describe --always` of the checkout this package lives in, or None (_core_utils.py:87-108)."""
    import os
    import shutil
    import subprocess

    git = shutil.which("git")
    if not git:
        return None
    try:
        res = subprocess.run([git, "describe", "--always"], capture_output=True, text=True, check=True,
                             cwd=os.path.dirname(os.path.abspath(__file__)))
    except (subprocess.CalledProcessError, OSError):
        return None
    return res.stdout.strip()


def convert_non_serializable(obj):
    """`json.dumps(default=...)` of the reference's writers (_data_utils.py:89-102)."""
    import numpy as np

    if isinstance(obj, (np.int64, np.int32)):
        return int(obj)
    if isinstance(obj, set):
        return list(obj)
    return str(obj)


def _string_arg_to_type(arg: str):
    low = arg.lower()
    if low == "true":
        return True
    if low == "false":
        return False
    if arg.isnumeric():
        return int(arg)
    try:
        return float(arg)
    except ValueError:
        return arg


def parse_string_args(args_string: str) -> dict:
    """`k=v,k=v` -> typed dict (bool / int / float / str), same grammar as the reference."""
    args_string = args_string.strip()
    if not args_string:
        return {}
    out = {}
    for item in args_string.split(","):
        if not item:
            continue
        k, v = item.split("=")
        out[k] = _string_arg_to_type(v)
    return out


def sanitize_list(sub):
    if isinstance(sub, list):
        return [sanitize_list(x) for x in sub]
    if isinstance(sub, tuple):
        return tuple(sanitize_list(x) for x in sub)
    return str(sub)


class Collator:
    """Group by gen_kwargs, sort by a key, batch, and restore the original order afterwards."""

    def __init__(self, data_source: list, sort_fn: Callable, group_fn: Callable = lambda x: x[1], grouping: bool = False):
        self._sort_fn = sort_fn
        self._grouping = grouping
        self._order: list[int] = []
        self.size = len(data_source)
        indexed = tuple(enumerate(data_source))
        if grouping:
            groups: dict = defaultdict(list)
            for item in indexed:
                key = group_fn(item[1])
                try:
                    key = tuple((k, tuple(v) if isinstance(v, Iterable) and not isinstance(v, str) else v) for k, v in sorted(key.items()))
                except (AttributeError, TypeError):
                    pass
                groups[key].append(item)
            self._data = groups
        else:
            self._data = indexed

    def _reorder(self, items):
        items = sorted(items, key=lambda x: self._sort_fn(x[1]))
        self._order.extend(i for i, _ in items)
        return [x for _, x in items]

    @staticmethod
    def _split(items: list, n: int):
        for i in range(0, len(items), max(n, 1)):
            yield items[i:i + n]

    def get_batched(self, n: int = 1, batch_fn=None):
        if self._grouping:
            for values in self._data.values():
                yield from self._split(self._reorder(values), n)
        else:
            yield from self._split(self._reorder(self._data), n)

    def get_original(self, new_arr: list) -> list:
        res: list = [None] * self.size
        seen = [False] * self.size
        if len(new_arr) != len(self._order):
            raise ValueError("Not all elements were covered in the reordering.")
        for i, v in zip(self._order, new_arr):
            res[i], seen[i] = v, True
        if not all(seen):
            raise ValueError("Not all elements were covered in the reordering.")
        return res

    def __len__(self) -> int:
        return self.size


def make_table(results: dict) -> str:
    """Markdown table `task | metric | value | stderr` of a results dict."""
    lines = ["| Task | Metric | Value | Stderr |", "|---|---|---:|---:|"]
    for task, metrics in results.get("results", {}).items():
        for k, v in metrics.items():
            metric = k.split(",")[0]
            if metric.endswith("_stderr") or k == "alias" or k.startswith(" "):
                continue
            se = metrics.get(f"{metric}_stderr,{k.split(',')[1]}" if "," in k else f"{metric}_stderr", "N/A")
            val = f"{v:.4f}" if isinstance(v, (int, float)) else str(v)
            se = f"{se:.4f}" if isinstance(se, (int, float)) else str(se)
            lines.append(f"| {task} | {metric} | {val} | {se} |")
    return "\n".join(lines)
