"""Batched evaluation loop — the slice of /root/reference/src/engine/_engine.py (`simple_evaluate` :393-637,
`evaluate` :32-389) that open-world classification exercises:

  seed -> build tasks + model -> per task: strided shard (rank r owns docs r, r+W, ...) -> requests ->
  `model.generate_until(requests)` -> take_first filter -> per-doc `process_results` -> sample records
  (incl. the three sha256 hashes, `_engine.py:262-279`) -> gather to rank 0 -> aggregation
  (`calculate_aggregate_metric`, tasks/_base.py:742-774; stderr only for `mean`).

Multi-GPU: one process per GPU, no data-path collective; a single `gather_object` per task moves the
per-doc records (a few hundred bytes each) to rank 0 over RCCL/gloo.  Uneven shards need no padding here
(the reference pads by repeating the last request, `_engine.py:176-191`, and then discards the extras).
"""

from __future__ import annotations

import json
import random
from collections import defaultdict

import numpy as np
import torch

from .. import utils
from ..metrics import mean_stderr
from ..models import get_model
from ..tasks import ClassificationTask, load_task


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def simple_evaluate(model: str, model_args: str | dict = "", tasks: list[str] | None = None, batch_size: int = 1,
                    limit: int | float | None = None, gen_kwargs: str | dict | None = None, random_seed: int = 0,
                    numpy_random_seed: int = 1234, torch_random_seed: int = 1234, include_path: str | None = None,
                    data_root: str = "data", log_samples: bool = True, task_objects: dict | None = None,
                    model_object=None) -> dict | None:
    random.seed(random_seed)
    np.random.seed(numpy_random_seed)
    torch.manual_seed(torch_random_seed)
    task_dict: dict[str, ClassificationTask] = dict(task_objects or {})
    for name in tasks or []:
        t = load_task(name, data_root=data_root, include_path=include_path)
        task_dict[t.task_name] = t  # requests carry task.task_name (tasks/_manager.py:895-902)
    if isinstance(model_args, str):
        model_args = utils.parse_string_args(model_args)
    lm = model_object if model_object is not None else get_model(model, batch_size=batch_size, **model_args)
    lm.eval()
    torch.set_grad_enabled(False)
    if gen_kwargs:
        gk = utils.parse_string_args(gen_kwargs) if isinstance(gen_kwargs, str) else dict(gen_kwargs)
        for t in task_dict.values():
            t.generation_kwargs.update(gk)
    for t in task_dict.values():
        lm.task_dict[t.task_name] = t.dataset
    results = evaluate(lm, task_dict, limit=limit, log_samples=log_samples)
    if results is not None:
        results["config"] = {"model": model if isinstance(model, str) else type(lm).__name__, "model_args": model_args,
                             "batch_size": batch_size, "limit": limit, "gen_kwargs": gen_kwargs,
                             "random_seed": random_seed, "numpy_seed": numpy_random_seed, "torch_seed": torch_random_seed}
    return results


def evaluate(lm, task_dict: dict, limit: int | float | None = None, log_samples: bool = True) -> dict | None:
    rank, world = lm.rank, lm.world_size
    dist = _dist()
    results: dict = {"results": {}, "samples": {}, "n-samples": {}, "higher_is_better": {}}
    for task_name, task in task_dict.items():
        n_docs = len(task.docs)
        lim = None if limit is None else (int(n_docs * limit) if isinstance(limit, float) and limit < 1.0 else int(limit))
        task.build_all_requests(limit=lim, rank=rank, world_size=world)
        reqs = task.instances
        # requests of a task share one type (generate_until | generate_until_multi_round): dispatch like _engine.py:243-262
        resps = getattr(lm, task.OUTPUT_TYPE)(reqs) if reqs else []
        for r, x in zip(reqs, resps, strict=True):
            r.resps.append(x)
        if dist is not None:
            dist.barrier()
        task.apply_filters()
        samples, metric_items = [], defaultdict(list)
        for req in reqs:
            doc = req.doc
            metrics = task.process_results(doc, [req.filtered_resps["none"]])
            target = task.doc_to_target(doc)
            saved_doc = {k: v for k, v in doc.items() if isinstance(v, (str, int, float, bool, list, dict, type(None)))}
            if log_samples:
                example = {
                    "doc_id": req.doc_id, "doc": saved_doc, "target": target,
                    "arguments": [a for a in req.args if isinstance(a, (str, int, float, bool, list, dict, type(None)))],
                    "resps": [req.resps], "filtered_resps": [req.filtered_resps["none"]],
                    "doc_hash": utils.hash_string(json.dumps(saved_doc, indent=2, default=str, ensure_ascii=False)),
                    "prompt_hash": utils.hash_string(req.args[0]), "target_hash": utils.hash_string(str(target)),
                }
                example.update(metrics)
                samples.append(example)
            for m, v in metrics.items():
                metric_items[m].append((req.doc_id, v))
        # ---- the only exchange step: per-doc records to rank 0
        if dist is not None:
            gathered_s = [None] * world if rank == 0 else None
            gathered_m = [None] * world if rank == 0 else None
            dist.gather_object(samples, gathered_s, dst=0)
            dist.gather_object(dict(metric_items), gathered_m, dst=0)
            if rank == 0:
                samples = [s for part in gathered_s for s in part]
                merged = defaultdict(list)
                for part in gathered_m:
                    for m, v in part.items():
                        merged[m].extend(v)
                metric_items = merged
        if rank != 0:
            continue
        samples.sort(key=lambda s: s["doc_id"])
        agg, out = task.aggregation(), {}
        for m, items in metric_items.items():
            vals = [v for _, v in sorted(items, key=lambda t: t[0])]
            out[f"{m},none"] = agg[m](vals)
            out[f"{m}_stderr,none"] = mean_stderr(vals) if agg[m].__name__ == "mean" and len(vals) > 1 else "N/A"
        results["results"][task_name] = out
        results["samples"][task_name] = samples
        results["n-samples"][task_name] = {"original": n_docs, "effective": len(samples) if log_samples else None}
        results["higher_is_better"][task_name] = task.higher_is_better()
    if dist is not None:
        dist.barrier()
    return results if rank == 0 else None
