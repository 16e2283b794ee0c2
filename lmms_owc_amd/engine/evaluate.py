"""Batched evaluation loop — the slice of /root/reference/src/engine/_engine.py (`simple_evaluate` :393-637,
`evaluate` :32-389) that open-world classification exercises:

  seed -> build tasks + model -> per task: strided shard (rank r owns docs r, r+W, ...) -> requests ->
  `model.generate_until(requests)` -> [ONE fixed-width all_gather of the shard's answers] -> rank 0: take_first filter ->
  per-doc `process_results` -> sample records (incl. the three sha256 hashes, `_engine.py:262-279`) -> aggregation
  (`calculate_aggregate_metric`, tasks/_base.py:742-774; stderr only for `mean`).

The results dict carries the reference's keys (`results` incl. `alias`, `group_subtasks`, `configs`, `versions`, `n-shot`,
`higher_is_better`, `n-samples`, `config`, `git_hash`, `date`; pinned on the reference's own output in
tests/golden/engine_formats.json).

Multi-GPU: one process per GPU, no data-path collective.  The reference moves pickled per-doc dicts with two
`gather_object` calls per task (`_engine.py:298-315`) and scores every document on rank 0; here every rank SCORES ITS OWN
documents (take_first filter, `process_results`, the sample record with its hashes - round 4: on rank 0 alone that tail ran at
~10 k documents/s, i.e. 18 % on top of the GPU time of an 8-rank task, `tools/time_rank_tail.py`) and contributes ONE
FIXED-WIDTH int32 record per document,

    { doc_id, n, payload[width] }      payload = the UTF-8 bytes of the JSON-encoded [sample record, metric values],

all ranks pad their shard to the largest shard (sizes follow from the shard function; the record width is agreed by one
8-byte all_reduce(MAX)) and ONE `all_gather_into_tensor` (RCCL over xGMI; gloo in the CPU tests) brings them together.  Rank 0
only parses the records and lines them up in doc_id order, so an N-rank run writes byte-identical files to a 1-rank run (the
reference's N-rank file holds the same lines in rank-major order).  Uneven shards need no padding requests (the reference
re-runs the last request, `_engine.py:176-191`, and discards the extras).
"""

from __future__ import annotations

import json
import math
import random
from collections import defaultdict

import numpy as np
import torch

from .. import utils
from ..metrics import mean_stderr
from ..models import get_model
from ..tasks import ClassificationTask, load_task

_SERIALIZABLE = (str, int, float, bool, list, dict, type(None))
DOC_HASH_OF_NONE = utils.hash_string(json.dumps(None, indent=2, ensure_ascii=False))


def _dist():
    import torch.distributed as dist

    return dist if dist.is_available() and dist.is_initialized() else None


def simple_evaluate(model: str, model_args: str | dict = "", tasks: list[str] | None = None, batch_size: int = 1,
                    limit: int | float | None = None, gen_kwargs: str | dict | None = None, random_seed: int = 0,
                    numpy_random_seed: int = 1234, torch_random_seed: int = 1234, fewshot_random_seed: int = 1234,
                    include_path: str | None = None, data_root: str = "data", log_samples: bool = True,
                    task_objects: dict | None = None, model_object=None, bootstrap_iters: int = 100000,
                    use_cache: str | None = None, datetime_str: str | None = None, samples_as_lines: bool = False) -> dict | None:
    random.seed(random_seed)
    np.random.seed(numpy_random_seed)
    torch.manual_seed(torch_random_seed)
    task_dict: dict[str, ClassificationTask] = dict(task_objects or {})
    for name in tasks or []:
        t = load_task(name, data_root=data_root, include_path=include_path)
        task_dict[t.task_name] = t  # requests carry task.task_name (tasks/_manager.py:895-902)
    if not task_dict:
        raise ValueError("No tasks specified, or no tasks found. Please verify the task names.")
    model_args_str = model_args if isinstance(model_args, str) else ",".join(f"{k}={v}" for k, v in model_args.items())
    kwargs = utils.parse_string_args(model_args) if isinstance(model_args, str) else dict(model_args)
    lm = model_object if model_object is not None else get_model(model, batch_size=batch_size, **kwargs)
    lm.eval()
    torch.set_grad_enabled(False)
    gk = None
    if gen_kwargs:
        gk = utils.parse_string_args(gen_kwargs) if isinstance(gen_kwargs, str) else dict(gen_kwargs)
        for t in task_dict.values():   # cli settings win over the yaml's (_engine.py:540-541)
            t.generation_kwargs.update(gk)
    for t in task_dict.values():
        lm.task_dict[t.task_name] = t.dataset
    results = evaluate(lm, task_dict, limit=limit, log_samples=log_samples, bootstrap_iters=bootstrap_iters,
                       samples_as_lines=samples_as_lines)
    torch.set_grad_enabled(True)
    if results is None:
        return None
    results["config"] = {"model": model if isinstance(model, str) else type(lm).__name__, "model_args": model_args_str,
                         "batch_size": batch_size, "batch_sizes": [], "use_cache": use_cache, "limit": limit,
                         "bootstrap_iters": bootstrap_iters, "gen_kwargs": gk, "random_seed": random_seed,
                         "numpy_seed": numpy_random_seed, "torch_seed": torch_random_seed, "fewshot_seed": fewshot_random_seed}
    results["git_hash"] = utils.get_git_commit_hash()
    results["date"] = datetime_str
    return results


# ---------------------------------------------------------------- the one exchange step
def shard_sizes(n_docs: int, limit: int | None, world: int) -> list[int]:
    """Documents each rank owns under `create_iterator` (islice(docs, rank, limit, world)): known to every rank."""
    stop = n_docs if limit is None else min(n_docs, int(limit))
    return [len(range(r, stop, world)) for r in range(world)]


def _pack_bytes(blobs: list[bytes]) -> tuple[np.ndarray, np.ndarray]:
    """UTF-8 blobs -> int32 words, one zero-padded row per document."""
    width = max([1] + [(len(b) + 3) // 4 for b in blobs])
    mat = np.zeros((len(blobs), width), np.int32)
    for i, b in enumerate(blobs):
        mat[i].view(np.uint8)[: len(b)] = np.frombuffer(b, np.uint8)
    return mat, np.array([len(b) for b in blobs], np.int32)


def gather_records(lm, doc_ids: list[int], blobs: list[bytes], sizes: list[int], rank: int, world: int, dist) -> dict | None:
    """All ranks: contribute one JSON document (UTF-8 bytes) per owned document as a fixed-width record; rank 0 returns
    {doc_id: parsed object}."""
    mat, nb = _pack_bytes(blobs)
    # RCCL moves device memory, gloo (CPU tests) host memory
    device = torch.device(lm.device) if dist is not None and dist.get_backend() == "nccl" else torch.device("cpu")
    width = torch.tensor([mat.shape[1] if len(blobs) else 0], dtype=torch.int64, device=device)
    if dist is not None:
        dist.all_reduce(width, op=dist.ReduceOp.MAX)   # the widest record of any rank (8 bytes; a rank may own no document)
    W = max(int(width.item()), 1)
    rows = max(sizes)
    if rows == 0:   # no rank owns a document (limit 0 / empty task)
        return {} if rank == 0 else None
    # the record block is put together on the host with numpy (torch's CPU kernels would first spin up one OpenMP thread per core -
    # 0.8 s on a 256-core host) and crosses to the device, when the backend wants device memory, as ONE copy
    rec_h = np.zeros((rows, 2 + W), np.int32)
    k = len(doc_ids)
    if k:
        rec_h[:k, 0] = doc_ids
        rec_h[:k, 1] = nb
        rec_h[:k, 2:2 + mat.shape[1]] = mat
    rec = torch.from_numpy(rec_h).to(device)
    if dist is not None:
        allrec = torch.empty((world * rows, 2 + W), dtype=torch.int32, device=device)
        dist.all_gather_into_tensor(allrec, rec)
    else:
        allrec = rec
    if rank != 0:
        return None
    allrec = np.ascontiguousarray(allrec.cpu().numpy().reshape(world, rows, 2 + W))
    out = {}
    for r in range(world):   # one JSON parse per rank: its records joined into an array (a parse per record costs 5x as much)
        part = allrec[r, : sizes[r]]
        if not len(part):
            continue
        raw = part[:, 2:].view(np.uint8).reshape(len(part), 4 * W)
        objs = json.loads(b"[" + b",".join(raw[i, : part[i, 1]].tobytes() for i in range(len(part))) + b"]")
        out.update(zip(part[:, 0].tolist(), objs))
    return out


def _json_native(v) -> bool:
    """True when `json.loads(json.dumps(v))` gives back an equal object of the same types (tuples, non-str keys and anything
    `convert_non_serializable` would stringify do not)."""
    if v is None or isinstance(v, (str, bool, int)):
        return True
    if isinstance(v, float):
        return v == v and v not in (float("inf"), float("-inf"))
    if type(v) is list:
        return all(_json_native(x) for x in v)
    if type(v) is dict:
        return all(type(k) is str and _json_native(x) for k, x in v.items())
    return False


_PICKLED = "__owc_pickled_metrics__"


def _wire_metrics(m: dict) -> dict:
    """Metric values as they cross to rank 0: themselves when JSON carries them unchanged, else pickled (the reference moves
    pickles: `gather_object`, `_engine.py:298-315`) - rank 0 must aggregate the SAME objects a 1-rank run aggregates."""
    if _json_native(m):
        return m
    import base64
    import pickle

    return {_PICKLED: base64.b64encode(pickle.dumps(m)).decode("ascii")}


def _unwire_metrics(m: dict) -> dict:
    if len(m) == 1 and _PICKLED in m:
        import base64
        import pickle

        return pickle.loads(base64.b64decode(m[_PICKLED]))
    return m


def doc_record(task, req, log_samples: bool) -> tuple[dict | None, dict]:
    """One document's share of the post-processing (`_engine.py:244-292`): metric values and, with `log_samples`, the sample record."""
    doc = req.doc
    metrics = task.process_results(doc, [req.filtered_resps["none"]])
    if not log_samples:
        return None, metrics
    target = task.doc_to_target(doc)
    example = {
        # the reference's filter (`_engine.py:263`) plus: values that JSON cannot carry are dropped (a decoded PIL
        # image under `visual` would be written as its repr, memory address included; the reference's rows hold a path)
        "doc_id": req.doc_id,
        "doc": {k: v for k, v in doc.items() if "image" not in k and isinstance(v, _SERIALIZABLE)}, "target": target,
        "arguments": [a for a in req.args if isinstance(a, _SERIALIZABLE)],
        "resps": [req.resps], "filtered_resps": [req.filtered_resps["none"]],
        # the reference hashes `requests[0].doc`, which its TaskInstance never sets (tasks/_manager.py:881 `# doc=doc`):
        # every record carries sha256("null") (`_engine.py:267-274`); kept for file compatibility
        "doc_hash": DOC_HASH_OF_NONE,
        "prompt_hash": utils.hash_string(req.args[0]), "target_hash": utils.hash_string(str(target)),
    }
    example.update(metrics)
    return example, metrics


def evaluate(lm, task_dict: dict, limit: int | float | None = None, log_samples: bool = True,
             bootstrap_iters: int | None = 100000, samples_as_lines: bool = False) -> dict | None:
    """`samples_as_lines` (multi-rank runs with `log_samples`; what eval_model.py asks for): `results["samples"]` holds
    `tracker.SampleLine` records - the owning rank has already produced the samples-file line, so rank 0's share of a task's tail
    is a parse of the metric values only (tools/time_rank_tail.py)."""
    from .tracker import SampleLine, sample_line

    rank, world = lm.rank, lm.world_size
    dist = _dist()
    results: dict = {k: {} for k in ("results", "group_subtasks", "configs", "versions", "n-shot", "higher_is_better", "n-samples")}
    results["samples"] = {}
    for task_name, task in task_dict.items():
        n_docs = len(task.docs)
        if limit is not None:   # `_engine.py:125-126` (ceil; the converted value is reused for later tasks, as there)
            limit = int(math.ceil(n_docs * limit)) if limit < 1.0 else int(limit)
        task.build_all_requests(limit=limit, rank=rank, world_size=world)
        reqs = task.instances
        sizes = shard_sizes(n_docs, limit, world)
        assert len(reqs) == sizes[rank]
        # requests of a task share one type (generate_until | generate_until_multi_round): dispatch like _engine.py:180-198
        resps = getattr(lm, task.OUTPUT_TYPE)(reqs) if reqs else []
        for r, x in zip(reqs, resps, strict=True):
            r.resps.append(x)
        task.apply_filters()
        # every rank post-processes the documents it owns ...
        local = [doc_record(task, req, log_samples) for req in reqs]
        if world > 1:   # ... and rank 0 receives [sample record, metric values] per document, ordered by doc_id below
            dumps = lambda o: json.dumps(o, default=utils.convert_non_serializable, ensure_ascii=False)  # noqa: E731
            if log_samples and samples_as_lines:   # [the finished samples-file line, its two hashes, metric values]
                blobs = [dumps([sample_line(dict(e)), e["prompt_hash"], e["target_hash"], _wire_metrics(m)]).encode("utf-8") for e, m in local]
            else:
                blobs = [dumps([e, _wire_metrics(m)]).encode("utf-8") for e, m in local]
            got = gather_records(lm, [r.doc_id for r in reqs], blobs, sizes, rank, world, dist)
            if rank != 0:
                continue
            if log_samples and samples_as_lines:
                local = [(SampleLine(got[i][0], DOC_HASH_OF_NONE, got[i][1], got[i][2]), _unwire_metrics(got[i][3])) for i in sorted(got)]
            else:
                local = [(got[i][0], _unwire_metrics(got[i][1])) for i in sorted(got)]
        samples, metric_items = [], defaultdict(list)
        for example, metrics in local:
            if log_samples:
                samples.append(example)
            for m, v in metrics.items():
                metric_items[m].append(v)
        agg, out = task.aggregation(), {"alias": task_name}
        for m, vals in metric_items.items():
            out[f"{m},none"] = agg[m](vals)
            # stderr only for `mean` (`metric in can_bootstrap` compares a function with names, so nothing is ever bootstrapped),
            # only for more than one value and only with bootstrap_iters > 0 (tasks/_base.py:758-771, metrics/_api.py:235-257)
            has_stderr = agg[m].__name__ == "mean" and len(vals) > 1 and (bootstrap_iters or 0) > 0
            out[f"{m}_stderr,none"] = mean_stderr(vals) if has_stderr else "N/A"
        results["results"][task_name] = out
        results["group_subtasks"][task_name] = []
        results["configs"][task_name] = task.dump_config()
        results["versions"][task_name] = "Yaml"
        results["n-shot"][task_name] = 0
        results["higher_is_better"][task_name] = task.higher_is_better()
        results["n-samples"][task_name] = {"original": n_docs, "effective": min(limit if limit else n_docs, n_docs)}
        if log_samples:
            results["samples"][task_name] = samples
    if dist is not None:
        dist.barrier()
    # generation is over: whatever host work follows in this process (writing the files, an in-process scorer) may use every core
    # again - the plug-in pinned its launch thread and workers to the GPU's NUMA share (models/_base.py `pin_to_gpu_numa_node`)
    unpin = getattr(lm, "unpin_host_threads", None)
    if callable(unpin):
        unpin()
    if rank != 0:
        return None
    if not log_samples:
        results.pop("samples")
    return results
