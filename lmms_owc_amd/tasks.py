"""The slice of the reference's task layer that the open-world classification loop touches:

* `TaskInstance` (src/data/tasks/_base.py:29-55) — request tuple
  `(context, gen_kwargs, doc_to_visual, doc_id, task, split)` (src/data/tasks/_manager.py:895-902);
* `build_all_requests` sharding through `create_iterator` (src/data/tasks/_base.py:363-368);
* `process_results`, generate_until branch (src/data/tasks/_manager.py:938-942, :1030-1090): strip the
  response, call `fn(references=[gold], predictions=[pred], **kw)` and fall back to storing `[gold, [pred]]`
  for passthrough metrics;
* the `take_first` filter (src/data/filters/_selection.py:36-52);
* `output_type: generate_until_multi_round` (the `*_llamav_o1` configs): request tuple
  `(context, gen_kwargs, doc_to_visual, doc_to_text, doc_id, task, split)` (_manager.py:903-915) where
  `doc_to_text(doc, round_idx=, previous_round_results=, last_round_info=)` follows
  `_caltech101_utils.doc_to_text_multi_round` (:29-72); `process_results` scores the LAST round (_manager.py:1033-1036).

Task definitions are small YAML files (prompt, generation kwargs, metric list) mirroring
src/data/tasks/_classification/*/base.yaml + assets/_default_template_yaml; documents come from a JSONL
manifest, a `datasets.load_from_disk` directory, or a seeded synthetic generator (no dataset exists offline).
"""

from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from pathlib import Path

import numpy as np
import yaml

from . import utils
from .metrics import get_metric_info

CONFIG_DIR = Path(__file__).resolve().parent / "task_configs"


@dataclass
class TaskInstance:
    request_type: str
    arguments: tuple
    idx: int
    doc: dict | None = None
    metadata: dict = field(default_factory=dict)
    resps: list = field(default_factory=list)
    filtered_resps: dict = field(default_factory=dict)
    task_name: str | None = None
    doc_id: int | None = None

    @property
    def args(self) -> tuple:
        return self.arguments if isinstance(self.arguments, tuple) else (self.arguments,)


DEFAULT_METRICS = [
    {"metric": "exact_match", "aggregation": "mean", "higher_is_better": True, "ignore_case": True,
     "ignore_punctuation": False, "regexes_to_ignore": [",", "\\$"]},
    {"metric": "semantic_similarity", "aggregation": "semantic_similarity", "higher_is_better": True},
    {"metric": "textual_inclusion", "aggregation": "mean", "higher_is_better": True},
]


class ClassificationTask:
    """`output_type: generate_until`, `test_split: test` open-world classification task."""

    OUTPUT_TYPE = "generate_until"

    def __init__(self, name: str, docs: list[dict], prompt: str = "What type of object is in this photo?",
                 pre_prompt: str = "", post_prompt: str = "", generation_kwargs: dict | None = None,
                 metric_list: list[dict] | None = None, split: str = "test", output_type: str = "generate_until",
                 prompts: list[str] | None = None) -> None:
        if output_type not in ("generate_until", "generate_until_multi_round"):
            raise ValueError(f"unsupported output_type {output_type!r} (generate_until, generate_until_multi_round)")
        self.OUTPUT_TYPE = output_type
        self.task_name = name
        self.dataset_path = f"data/{name}"   # informational (results file); load_task overwrites it with the real path
        self.split = split
        self.docs = docs
        self.prompt, self.pre_prompt, self.post_prompt = prompt, pre_prompt, post_prompt
        self.prompts = prompts
        if output_type == "generate_until_multi_round" and (not isinstance(prompts, list) or len(prompts) < 2):
            raise ValueError("`multi_round` expects at least two questions")
        self.generation_kwargs = dict(generation_kwargs or {"max_new_tokens": 64, "do_sample": False})
        if "temperature" in self.generation_kwargs:
            self.generation_kwargs["temperature"] = float(self.generation_kwargs["temperature"])
        if "until" not in self.generation_kwargs:   # TaskConfig.__post_init__, src/data/tasks/_config.py:199-205
            self.generation_kwargs["until"] = ["\n\n"]   # [fewshot_delimiter]
        self.metric_list = metric_list or DEFAULT_METRICS
        self._metric_fn, self._metric_kwargs, self._agg, self._higher = {}, {}, {}, {}
        for m in self.metric_list:
            m = dict(m)
            name_ = m.pop("metric")
            info = get_metric_info(name_)
            self._metric_fn[name_] = info.builder_fn
            self._agg[name_] = info.group_fn
            self._higher[name_] = m.pop("higher_is_better", info.higher_is_better)
            m.pop("aggregation", None)
            self._metric_kwargs[name_] = m
        self.instances: list[TaskInstance] = []

    # ---- document accessors (…/_caltech101_utils.py:75-94)
    @property
    def dataset(self) -> dict:
        return {self.split: self.docs}

    def doc_to_text(self, doc: dict) -> str:
        if self.OUTPUT_TYPE == "generate_until_multi_round":
            return self.doc_to_text_multi_round(doc)
        return f"{self.pre_prompt}{self.prompt}{self.post_prompt}"

    def doc_to_text_multi_round(self, doc: dict, round_idx: int | None = None, previous_round_results: list | None = None,
                                last_round_info: dict | None = None):
        """Round protocol of the reference: without a round index the first question; otherwise
        `(visual, text, should_terminate, previous_round_results, last_round_info)` - the image only travels in round 0
        (later rounds return visual None and rely on the message history in `last_round_info`)."""
        previous_round_results = [] if previous_round_results is None else previous_round_results
        if round_idx is None:
            return f"{self.pre_prompt}{self.prompts[0]}{self.post_prompt}"
        if round_idx < len(self.prompts):
            return None, f"{self.pre_prompt}{self.prompts[round_idx]}{self.post_prompt}", False, previous_round_results, last_round_info
        return None, None, True, previous_round_results, last_round_info

    @staticmethod
    def doc_to_target(doc: dict) -> str:
        return str(doc["target"]).replace("_", " ")

    @staticmethod
    def doc_to_visual(doc: dict) -> list:
        v = doc["visual"]
        if isinstance(v, (str, Path)):
            from PIL import Image

            return [Image.open(v).convert("RGB")]
        return [v.convert("RGB")]

    # ---- request construction with the reference's strided shard
    def build_all_requests(self, *, limit: int | None = None, rank: int = 0, world_size: int = 1) -> None:
        self.instances = []
        it = utils.create_iterator(enumerate(self.docs), rank, world_size, limit)
        for doc_id, doc in it:
            gk = copy.deepcopy(self.generation_kwargs)   # _manager.py:897, :906
            if self.OUTPUT_TYPE == "generate_until_multi_round":
                args = (self.doc_to_text(doc), gk, self.doc_to_visual, self.doc_to_text_multi_round, doc_id,
                        self.task_name, self.split)
            else:
                args = (self.doc_to_text(doc), gk, self.doc_to_visual, doc_id, self.task_name, self.split)
            self.instances.append(TaskInstance(self.OUTPUT_TYPE, args, idx=0, doc=doc, task_name=self.task_name, doc_id=doc_id))

    def apply_filters(self) -> None:
        for inst in self.instances:  # default ensemble ("none", [take_first])
            inst.filtered_resps["none"] = inst.resps[0]

    def process_results(self, doc: dict, results: list) -> dict:
        if isinstance(results, list) and results and isinstance(results[0], list):
            results = results[0]
        if self.OUTPUT_TYPE == "generate_until_multi_round":  # one tuple of per-round answers per response: score the last
            result = [r[-1].strip() for r in results]
        else:
            result = [r.strip() for r in results]
        gold = [self.doc_to_target(doc)]
        out = {}
        for metric, fn in self._metric_fn.items():
            try:
                score = fn(references=gold, predictions=result, **self._metric_kwargs[metric])
            except TypeError:  # passthrough metric: keep the pair for the batched aggregation
                score = fn([gold, result])
            if isinstance(score, dict):
                score = score[metric]
            out[metric] = score
        return out

    def aggregation(self) -> dict:
        return dict(self._agg)

    def dump_config(self) -> dict:
        """`configs[task]` of the results file: the keys `TaskConfig.to_dict` writes for a classification task
        (src/data/tasks/_config.py; key set and order pinned in tests/golden/engine_formats.json)."""
        # the YAML's own `model_specific_kwargs.default` when the task came from one (key order included); else the keys the
        # reference's task YAMLs of that type carry: pre_prompt / prompt / post_prompt, a multi-round task `prompts` in place of `prompt`
        msk = getattr(self, "_msk_yaml", None)
        if msk is None:
            msk = {"pre_prompt": self.pre_prompt, "prompt": self.prompt, "post_prompt": self.post_prompt}
            if self.prompts is not None:
                msk = {"pre_prompt": self.pre_prompt, "prompts": list(self.prompts), "post_prompt": self.post_prompt}
        msk = copy.deepcopy(msk)
        return {
            "task": self.task_name, "dataset_path": self.dataset_path, "dataset_kwargs": {}, "test_split": self.split,
            "full_docs": False, "process_results_use_image": False, "doc_to_visual": repr(self.doc_to_visual),
            "doc_to_text": repr(self.doc_to_text), "doc_to_target": repr(self.doc_to_target), "description": "",
            "target_delimiter": " ", "fewshot_delimiter": "\n\n", "num_fewshot": 0, "metric_list": copy.deepcopy(self.metric_list),
            "output_type": self.OUTPUT_TYPE, "generation_kwargs": copy.deepcopy(self.generation_kwargs), "repeats": 1,
            "should_decontaminate": False, "metadata": [{"version": 0.0}], "model_specific_kwargs": {"default": dict(msk), **msk},
        }

    def higher_is_better(self) -> dict:
        return dict(self._higher)


def load_task(name: str, *, data_root: str | Path = "data", include_path: str | Path | None = None) -> ClassificationTask:
    """`name` is a YAML task config (task_configs/ or --include_path) or `synthetic:<n>:<H>x<W>[:<classes>]`."""
    if name.startswith("synthetic"):
        return synthetic_task(name)  # "synthetic-mr:..." = the multi-round (llamav_o1-style) variant
    for base in ([Path(include_path)] if include_path else []) + [CONFIG_DIR]:
        f = base / f"{name}.yaml"
        if f.exists():
            cfg = yaml.safe_load(f.read_text())
            break
    else:
        raise KeyError(f"unknown task '{name}' (no {name}.yaml under {CONFIG_DIR} or --include_path)")
    msk = (cfg.get("model_specific_kwargs") or {}).get("default", {})
    docs = load_docs(Path(cfg.get("dataset_path", Path(data_root) / name)), cfg.get("test_split", "test"))
    output_type = cfg.get("output_type", "generate_until")
    dataset_path = str(cfg.get("dataset_path", Path(data_root) / name))
    # default question of the reference's doc_to_text when a config has no `prompt` key (_caltech101_utils.py:23)
    task = ClassificationTask(cfg.get("task", name), docs, prompt=msk.get("prompt", "What's in the image?"),
                              pre_prompt=msk.get("pre_prompt", ""), post_prompt=msk.get("post_prompt", ""),
                              generation_kwargs=cfg.get("generation_kwargs"), metric_list=cfg.get("metric_list"),
                              split=cfg.get("test_split", "test"), output_type=output_type,
                              prompts=msk.get("prompts") if output_type == "generate_until_multi_round" else None)
    if msk:
        task._msk_yaml = dict(msk)
    task.dataset_path = dataset_path
    return task


def load_docs(path: Path, split: str) -> list[dict]:
    manifest = path / f"{split}.jsonl"
    if manifest.exists():
        docs = [json.loads(line) for line in manifest.read_text().splitlines() if line.strip()]
        for d in docs:
            if isinstance(d["visual"], str) and not Path(d["visual"]).is_absolute():
                d["visual"] = str(path / d["visual"])
        return docs
    if (path / "dataset_dict.json").exists() or (path / split).exists():
        import datasets

        ds = datasets.load_from_disk(str(path))
        ds = ds[split] if isinstance(ds, datasets.DatasetDict) else ds
        return [dict(r) for r in ds]
    raise FileNotFoundError(f"no {split}.jsonl manifest or saved `datasets` directory under {path}")


def synthetic_task(spec: str) -> ClassificationTask:
    """`synthetic:<n>:<H>x<W>:<classes>` — seeded uint8 images + class names (benchmarks / tests)."""
    from PIL import Image

    parts = spec.split(":")
    n = int(parts[1]) if len(parts) > 1 else 16
    h, w = (int(x) for x in parts[2].split("x")) if len(parts) > 2 else (448, 448)
    c = int(parts[3]) if len(parts) > 3 else 10
    rng = np.random.default_rng(1234)
    names = [f"class_{i}" for i in range(c)]
    docs = []
    for i in range(n):
        arr = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        docs.append({"visual": Image.fromarray(arr, "RGB"), "target": names[i % c]})
    if parts[0] == "synthetic-mr":
        return ClassificationTask("synthetic", docs, generation_kwargs={"max_new_tokens": 6, "do_sample": False},
                                  output_type="generate_until_multi_round",
                                  prompts=["What type of object in this photo? Generate a summary of the picture.",
                                           "Generate a detailed caption for the image.", "Generate the final answer based on reasoning steps."])
    return ClassificationTask("synthetic", docs, generation_kwargs={"max_new_tokens": 8, "do_sample": False})
