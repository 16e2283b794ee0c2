#!/usr/bin/env python3
"""What rank 0 does AFTER the GPUs are done with a task, timed without GPUs (reference src/engine/_engine.py:293-322: gather ->
per-doc process_results -> sample records -> aggregation; src/engine/_tracker.py:220-341: the two files).

N gloo ranks run the real `evaluate()` + `EngineTracker` with a model stand-in whose `generate_until` returns its shard's answers
at once, so the wall time IS the tail: every rank scores its own documents (`process_results`, sample records with their sha256
hashes), fixed-width `all_gather_into_tensor` of N x docs_per_rank JSON records -> rank 0: parse, order by doc_id, aggregation,
results JSON + samples JSONL.  (Round 3 scored every document on rank 0: 10.4 k documents/s in the build container, i.e. 18 % on
top of an 8-rank task's GPU time - DESIGN.md section 6.)  Reported against the time the GPUs need for the same
documents (`--gpu-rate` images/s per rank), i.e. the share of a task the other ranks would stand idle for.

  python tools/time_rank_tail.py --ranks 8 --docs-per-rank 2048
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import tempfile
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(args) -> None:
    import numpy as np
    import torch.distributed as dist

    from lmms_owc_amd.engine import evaluate as ev
    from lmms_owc_amd.engine.tracker import EngineTracker
    from lmms_owc_amd.tasks import ClassificationTask

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    names = [f"class number {i}" for i in range(101)]

    class TailModel:
        device = "cpu"
        task_dict: dict = {}

        def __init__(self):
            self.rank, self.world_size = rank, world

        def eval(self):
            return self

        def generate_until(self, requests):
            out = []
            for q in requests:
                q.args[1].pop("until", None)
                out.append(names[q.doc_id % 101] if q.doc_id % 2 else "a photo of something else")
            return out

    n = args.docs_per_rank * world
    docs = [{"visual": f"img{i}.jpg", "target": names[i % 101]} for i in range(n)]
    metrics = [{"metric": "exact_match", "aggregation": "mean", "ignore_case": True, "regexes_to_ignore": [",", "\\$"]},
               {"metric": "textual_inclusion", "aggregation": "mean"}]
    task = ClassificationTask("tail", docs, metric_list=metrics)
    task.doc_to_visual = lambda doc: []
    lm = TailModel()
    lm.task_dict["tail"] = task.dataset
    out_dir = Path(args.out)
    stamps = {}
    orig_gather = ev.gather_records

    def timed_gather(*a, **k):
        t0 = time.perf_counter()
        out = orig_gather(*a, **k)
        stamps["gather_and_parse_s"] = time.perf_counter() - t0
        return out

    ev.gather_records = timed_gather
    import torch

    stamps["collective_s"] = 0.0
    for name in ("all_reduce", "all_gather_into_tensor"):   # the collectives themselves: gloo over loopback TCP on host tensors
        def timed(*a, _f=getattr(dist, name), **k):         # here, RCCL on device tensors in a real run - reported separately
            t0 = time.perf_counter()
            out = _f(*a, **k)
            stamps["collective_s"] += time.perf_counter() - t0
            return out

        setattr(dist, name, timed)

    # the first collective of each kind sets up the backend's connections (seconds with gloo's full mesh, once per run with RCCL
    # too): not part of a task's tail
    w = torch.zeros(1, dtype=torch.int64)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    g = torch.empty(world * 1024, dtype=torch.int32)
    dist.all_gather_into_tensor(g, torch.zeros(1024, dtype=torch.int32))
    dist.barrier()
    t0 = time.perf_counter()
    if args.profile and rank == 0:
        import cProfile
        import pstats

        pr = cProfile.Profile()
        res = pr.runcall(ev.evaluate, lm, {"tail": task}, limit=None, log_samples=True)
        pstats.Stats(pr, stream=sys.stderr).sort_stats("cumulative").print_stats(18)
    else:
        res = ev.evaluate(lm, {"tail": task}, limit=None, log_samples=True)
    t_eval = time.perf_counter() - t0
    if res is not None:
        tracker = EngineTracker(output_path=str(out_dir))
        tracker.log_experiment_args(model_source="stub", model_args="")
        samples = res.pop("samples")
        t1 = time.perf_counter()
        tracker.save_results_aggregated(results=res, samples=samples, datetime_str="2026-01-02T03:04:05")
        tracker.save_results_samples(task_name="tail", samples=samples["tail"])
        t_files = time.perf_counter() - t1
        print(json.dumps({"ranks": world, "documents": n, "evaluate_s": t_eval, **stamps, "files_s": t_files,
                          "tail_s": t_eval + t_files, "tail_without_collective_s": t_eval + t_files - stamps["collective_s"],
                          "documents_per_s_without_collective": n / (t_eval + t_files - stamps["collective_s"]),
                          "gpu_seconds_for_the_same_documents": n / (world * args.gpu_rate),
                          "tail_without_collective_over_gpu_time": (t_eval + t_files - stamps["collective_s"]) / (n / (world * args.gpu_rate))}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--docs-per-rank", type=int, default=2048)
    ap.add_argument("--gpu-rate", type=float, default=240.0)
    ap.add_argument("--profile", action="store_true", help="cProfile of rank 0's evaluate() on stderr")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for rk in range(args.ranks):
            env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(args.ranks), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            procs.append(subprocess.Popen([sys.executable, __file__, *sys.argv[1:], "--child", "--out", td], env=env,
                                          stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=True))   # (stderr inherited)
        out, _ = procs[0].communicate()
        for p in procs[1:]:
            p.wait()
        if any(p.returncode for p in procs):
            raise SystemExit("a rank failed")
    print([ln for ln in out.splitlines() if ln.startswith("{")][-1])


if __name__ == "__main__":
    main()
