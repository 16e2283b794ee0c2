#!/usr/bin/env python3
"""Online evaluation CLI — flag-compatible subset of /root/reference/eval_model.py:379-586 for the
open-world classification path.  One process per GPU:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 eval_model.py \
        --model qwen2-vl-7b --tasks caltech101 --batch_size 64 --output_path logs/schedule/caltech101/qwen2-vl-7b

(the reference's `accelerate launch --num_processes=N -m eval_model …`, scripts/schedule_batch.sh:109-112).
Flags that configure out-of-scope subsystems (W&B, Hub push, request cache, few-shot) are accepted and ignored."""

from __future__ import annotations

import argparse
import datetime
import logging
import os
import sys

os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

from lmms_owc_amd import utils
from lmms_owc_amd.engine.evaluate import simple_evaluate
from lmms_owc_amd.engine.tracker import EngineTracker

log = logging.getLogger("eval_model")


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser(formatter_class=argparse.RawTextHelpFormatter)
    p.add_argument("--config", default="", help="YAML list of argument overrides run sequentially")
    p.add_argument("--model", default="qwen2-vl-7b")
    p.add_argument("--tasks", default=None, help="comma-separated task names")
    p.add_argument("--model_args", default="", help="k=v,k=v passed to the model builder")
    p.add_argument("--num_fewshot", type=int, default=None)
    p.add_argument("--batch_size", "-b", type=str, default="1")
    p.add_argument("--max_batch_size", type=int, default=None)
    p.add_argument("--device", type=str, default=None)
    p.add_argument("--output_path", default=None, type=str)
    p.add_argument("--limit", type=float, default=None)
    p.add_argument("--use_cache", "-c", type=str, default=None)
    p.add_argument("--cache_requests", type=str, default=None, choices=["true", "refresh", "delete"])
    p.add_argument("--check_integrity", action="store_true")
    p.add_argument("--write_out", "-w", action="store_true", default=False)
    p.add_argument("--log_samples", action="store_true", default=False)
    p.add_argument("--wandb_log_samples", action="store_true", default=False)
    p.add_argument("--log_samples_suffix", type=str, default="model_outputs")
    p.add_argument("--system_instruction", type=str, default=None)
    p.add_argument("--apply_chat_template", action="store_true", default=False)
    p.add_argument("--fewshot_as_multiturn", action="store_true", default=False)
    p.add_argument("--show_config", action="store_true", default=False)
    p.add_argument("--include_path", type=str, default=None)
    p.add_argument("--gen_kwargs", default="")
    p.add_argument("--verbosity", "--log_level", dest="verbosity", type=str, default="INFO")
    p.add_argument("--wandb_args", default="")
    p.add_argument("--timezone", default="Asia/Singapore")
    p.add_argument("--hf_hub_log_args", type=str, default="")
    p.add_argument("--predict_only", "-x", action="store_true", default=False)
    p.add_argument("--seed", type=str, default="0,1234,1234,1234")
    p.add_argument("--trust_remote_code", action="store_true")
    p.add_argument("--data_root", type=str, default="data")
    # reference eval_model.py:582.  Its only reader (src/engine/_engine.py:221-228) picks a media-free doc iterator when the
    # flag is absent, and line 230 then overwrites that choice with `task.doc_iterator(...)` unconditionally - the flag changes
    # nothing in the reference, so it is accepted here and changes nothing either (the post-processing loop always iterates
    # the task's documents, engine/evaluate.py)
    p.add_argument("--process_with_media", action="store_true", help="accepted for command-line compatibility (no effect, as in the reference)")
    return p.parse_args(argv)


def run_single(args: argparse.Namespace, date_id: str) -> dict | None:
    if args.num_fewshot not in (None, 0):
        raise ValueError("open-world classification runs are 0-shot")
    if not args.tasks:
        raise SystemExit("--tasks is required")
    seeds = [int(s) if s != "None" else None for s in args.seed.split(",")]
    seeds = (seeds * 4)[:4] if len(seeds) == 1 else seeds
    limit = args.limit if args.limit is None or args.limit < 1.0 else int(args.limit)
    tracker = EngineTracker(output_path=args.output_path)
    tracker.log_experiment_args(model_source=args.model, model_args=args.model_args, system_instruction=args.system_instruction,
                                chat_template=None, fewshot_as_multiturn=args.fewshot_as_multiturn)
    results = simple_evaluate(model=args.model, model_args=args.model_args, tasks=args.tasks.split(","),
                              batch_size=int(args.batch_size), limit=limit, gen_kwargs=args.gen_kwargs,
                              random_seed=seeds[0] or 0, numpy_random_seed=seeds[1] or 1234, torch_random_seed=seeds[2] or 1234,
                              fewshot_random_seed=seeds[3] or 1234, include_path=args.include_path, data_root=args.data_root,
                              log_samples=args.log_samples, use_cache=args.use_cache, datetime_str=date_id,
                              samples_as_lines=True)   # the samples only travel on to the tracker
    if results is not None:
        samples = results.pop("samples") if args.log_samples else None   # eval_model.py:217-233
        tracker.save_results_aggregated(results=results, samples=samples, datetime_str=date_id)
        if args.log_samples:
            for task_name in results["configs"]:
                tracker.save_results_samples(task_name=task_name, samples=samples[task_name])
        print(utils.make_table(results))
    return results


def main(argv=None) -> None:
    args = parse_args(argv)
    logging.basicConfig(level=args.verbosity)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist

        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=datetime.timedelta(seconds=60000))
        stamp = [datetime.datetime.now().isoformat() if dist.get_rank() == 0 else None]
        dist.broadcast_object_list(stamp, src=0)  # one start timestamp for every rank (eval_model.py:334-336)
        date_id = stamp[0]
    else:
        date_id = datetime.datetime.now().isoformat()
    configs = [args]
    if args.config:
        import yaml

        overrides = yaml.safe_load(open(args.config))
        overrides = overrides if isinstance(overrides, list) else [overrides]
        configs = []
        for ov in overrides:
            ns = argparse.Namespace(**vars(args))
            for k, v in ov.items():
                setattr(ns, k, v)
            configs.append(ns)
    for cfg in configs:
        try:
            run_single(cfg, date_id)
        except Exception:
            if args.verbosity == "DEBUG":
                raise
            log.exception("Error during evaluation; continuing with the next configuration")  # eval_model.py:351-361
    if world > 1:
        import torch.distributed as dist

        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1:])
