"""On-disk outputs in the reference's formats (mirror of /root/reference/src/engine/_tracker.py:25-123, :220-262,
:297-341): `{date}_results.json` and `{date}_samples_{task}.jsonl` under `{output_path}/{model_name_sanitized}` — the
files `eval_metrics.py` consumes.  Key sets, key order and value formats are pinned on the reference's own tracker output
(tests/golden/engine_formats.json).  Hub pushing is out of scope."""

from __future__ import annotations

import json
import re
import time
from pathlib import Path

from .. import utils


def model_name_from_args(model_args: str) -> str:
    """GeneralConfigTracker._get_model_name (_tracker.py:56-80): first of these keys present in `--model_args`."""
    for prefix in ("peft=", "delta=", "pretrained=", "model=", "path=", "engine="):   # order matters
        if prefix in model_args:
            return model_args.split(prefix)[1].split(",")[0]
    return ""   # registry models carry no `pretrained=`: sample files land directly in --output_path


def sanitize_model_name(model_name: str) -> str:
    parts = model_name.split("/")   # utils.sanitize_model_name (_core_utils.py:265-280): keep org/name of a hub id
    return re.sub(r"[\"<>:/\|\\?\*\[\]]+", "__", "/".join(parts[-2:]) if len(parts) > 1 else parts[-1])


class EngineTracker:
    def __init__(self, output_path: str | None = None, **_ignored) -> None:
        self.output_path = output_path
        self.date_id = None
        # GeneralConfigTracker fields, in the order `asdict` writes them
        self.general = {"model_source": None, "model_name": None, "model_name_sanitized": None, "system_instruction": None,
                        "system_instruction_sha": None, "fewshot_as_multiturn": None, "chat_template": None,
                        "chat_template_sha": None, "start_time": time.perf_counter(), "end_time": None,
                        "total_evaluation_time_seconds": None}

    @property
    def model_name_sanitized(self) -> str:
        return self.general["model_name_sanitized"] or ""

    def log_experiment_args(self, model_source: str = "", model_args: str | dict = "", system_instruction: str | None = None,
                            chat_template: str | None = None, fewshot_as_multiturn: bool = False) -> None:
        if not isinstance(model_args, str):
            model_args = ",".join(f"{k}={v}" for k, v in model_args.items())
        g = self.general
        g["model_source"] = model_source
        g["model_name"] = model_name_from_args(model_args)
        g["model_name_sanitized"] = sanitize_model_name(g["model_name"])
        g["system_instruction"] = system_instruction
        g["system_instruction_sha"] = utils.hash_string(system_instruction) if system_instruction else None
        g["chat_template"] = chat_template
        g["chat_template_sha"] = utils.hash_string(chat_template) if chat_template else None
        g["fewshot_as_multiturn"] = fewshot_as_multiturn

    def _dir(self) -> Path:
        p = Path(self.output_path) / self.model_name_sanitized
        p.mkdir(parents=True, exist_ok=True)
        return p

    def save_results_aggregated(self, results: dict, samples: dict | None = None, datetime_str: str | None = None) -> Path | None:
        g = self.general
        g["end_time"] = time.perf_counter()
        g["total_evaluation_time_seconds"] = str(g["end_time"] - g["start_time"])
        if not self.output_path:
            return None
        task_hashes = {}
        if samples:
            task_hashes = {t: utils.hash_string("".join(s["doc_hash"] + s["prompt_hash"] + s["target_hash"] for s in ss))
                           for t, ss in samples.items()}
        results.update({"task_hashes": task_hashes})
        results.update(g)
        self.date_id = (datetime_str or "").replace(":", "-")
        f = self._dir() / f"{self.date_id}_results.json"
        f.open("w", encoding="utf-8").write(json.dumps(results, indent=2, default=utils.convert_non_serializable, ensure_ascii=False))
        return f

    def save_results_samples(self, task_name: str, samples: list[dict]) -> Path | None:
        if not self.output_path:
            return None
        f = self._dir() / f"{self.date_id}_samples_{task_name}.jsonl"
        with f.open("a", encoding="utf-8") as fh:
            for sample in samples:
                fh.write((sample.line if isinstance(sample, SampleLine) else sample_line(sample)) + "\n")
        return f


def sample_line(sample: dict) -> str:
    """One line of `<date>_samples_<task>.jsonl` (reference src/engine/_tracker.py:310-335) from a sample record of `evaluate`."""
    # `for key, value in enumerate(sample["arguments"][1])` over the gen_kwargs DICT (_tracker.py:318-322): the file
    # records {position: KEY NAME} of the request's generation kwargs, not their values
    arguments = dict(enumerate(sample["arguments"][1]))
    sample["input"] = sample["arguments"][0]
    sample["resps"] = utils.sanitize_list(sample["resps"])
    sample["filtered_resps"] = utils.sanitize_list(sample["filtered_resps"])
    sample["arguments"] = arguments
    sample["target"] = str(sample["target"])
    return json.dumps(sample, default=utils.convert_non_serializable, ensure_ascii=False)


class SampleLine(dict):
    """A sample record that already IS its samples-file line (`evaluate(..., samples_as_lines=True)` in a multi-rank run: the rank
    that owns the document serialises it, rank 0 neither parses nor re-encodes 50 000 records).  As a dict it carries what
    `save_results_aggregated` reads - the three hashes; `.record()` parses the line for anything else."""

    def __init__(self, line: str, doc_hash: str, prompt_hash: str, target_hash: str) -> None:
        super().__init__(doc_hash=doc_hash, prompt_hash=prompt_hash, target_hash=target_hash)
        self.line = line

    def record(self) -> dict:
        return json.loads(self.line)
