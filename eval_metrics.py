#!/usr/bin/env python3
"""Offline scorer — same command line and file conventions as /root/reference/eval_metrics.py:19-204:

    python eval_metrics.py -i 'logs/schedule/**' -m semantic_similarity,textual_inclusion

Globs `*_samples_*.jsonl` (path convention `logs/schedule/{task}/{model}/…`, :59-60), reads
`filtered_resps` / `target`, runs the registered metrics (the sentence-embedding ones on the MI355X scorer),
writes the per-sample columns back into the JSONL in place (:119-123) and prints `{model:<29}: {value:.3f}`."""

from __future__ import annotations

import logging
import random
from argparse import ArgumentParser, Namespace
from pathlib import Path

import numpy as np
import pandas as pd
import torch

from lmms_owc_amd.metrics import get_metric_info

log = logging.getLogger("eval_metrics")

SAVE_INTERMEDIATE = ["concept_semantic_similarity", "mean_average_semantic_similarity", "semantic_similarity"]


def main(args: Namespace) -> dict:
    logging.basicConfig(level=args.log_level)
    if args.seed:
        random.seed(args.seed)
        np.random.seed(args.seed)
        torch.manual_seed(args.seed)
    input_paths = sorted(Path().glob(args.input)) if "*" in args.input else [Path(args.input)]
    files = sorted(str(f) for p in input_paths for f in (p.glob("**/*_samples_*.jsonl") if p.is_dir() else [p]))
    log.info("Found %d `jsonl` files to process", len(files))
    tasks_outputs: dict = {}
    for input_file in files:
        task_name, model_name = Path(input_file).parent.parent.name, Path(input_file).parent.name
        df = pd.read_json(input_file, lines=True)
        predictions, references = df["filtered_resps"].tolist(), df["target"].tolist()
        if isinstance(predictions[0], list) and isinstance(predictions[0][0], list):  # multi-round: inner list
            predictions = [p[0] for p in predictions]
        items = list(zip(references, predictions, strict=True))
        outputs: dict = {"_num_samples": len(items)}
        for metric_name in args.metrics.split(","):
            info = get_metric_info(metric_name)
            if info.name == "textual_inclusion":
                output = info.builder_fn([p[-1] for p in predictions], [str(r) for r in references])
            elif info.name in SAVE_INTERMEDIATE:
                output = info.group_fn(info.builder_fn([(str(r), p) for r, p in items]), reduce="none")
                extra = {}
                if info.name == "concept_semantic_similarity":  # (concepts, similarities) per sample (:94-104)
                    concepts, sims = zip(*output, strict=True)
                    output = [float(np.max(row)) for row in sims]
                    extra["last_resp_concepts"] = list(concepts)
                    extra["last_resp_concepts_similarities"] = list(sims)
                if info.name == "mean_average_semantic_similarity":
                    avg = output.pop("semantic_similarity@avg")
                    extra.update(output)
                    output = avg
                df[info.name] = output
                for k, v in extra.items():
                    df[k] = v
                df.to_json(input_file, lines=True, orient="records")
                output = np.mean(output)
            else:
                output = info.group_fn(info.builder_fn(items))
            if isinstance(output, dict):
                outputs.update(output)
            else:
                outputs[metric_name] = output
        prev = tasks_outputs.setdefault(task_name, {}).get(model_name)
        if prev is None or outputs["_num_samples"] > prev["_num_samples"]:  # keep the larger run (:140-153)
            tasks_outputs[task_name][model_name] = outputs
    for task_name, task_outputs in tasks_outputs.items():
        names = sorted({k for o in task_outputs.values() for k in o if not k.startswith("_")})
        for metric_name in names:
            text = f"{metric_name.capitalize().replace('_', ' ')} on {task_name}:\n"
            for model_name, o in task_outputs.items():
                text += f"{model_name:<29}: {o[metric_name]:.3f}\n"
            print(text)
    return tasks_outputs


if __name__ == "__main__":
    parser = ArgumentParser()
    parser.add_argument("-i", "--input", required=True, type=str, help="Path to the folder/file containing the samples to process")
    parser.add_argument("-m", "--metrics", required=True, type=str, help="List of comma-separated metrics to evaluate on the data")
    parser.add_argument("--seed", type=int, default=1234, help="Random seed for reproducibility (default: 1234)")
    parser.add_argument("--log-level", type=str, default="INFO", help="Logging level (default: INFO)")
    main(parser.parse_args())
