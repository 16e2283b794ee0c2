"""On-disk outputs in the reference's formats (mirror of /root/reference/src/engine/_tracker.py:220-262,
:297-341): `{date}_results.json` and `{date}_samples_{task}.jsonl` under
`{output_path}/{model_name_sanitized}` — the files `eval_metrics.py` consumes.  Hub pushing is out of scope."""

from __future__ import annotations

import json
import re
import time
from datetime import datetime
from pathlib import Path

from .. import utils


class EngineTracker:
    def __init__(self, output_path: str | None = None, **_ignored) -> None:
        self.output_path = output_path
        self.start_time = time.perf_counter()
        self.date_id = datetime.now().isoformat().replace(":", "-")
        self.model_name_sanitized = ""

    def log_experiment_args(self, model_args: str = "", **_) -> None:
        args = utils.parse_string_args(model_args) if isinstance(model_args, str) else dict(model_args)
        name = str(args.get("pretrained", ""))  # registry models carry no `pretrained=` -> "" (_tracker.py:82-87)
        self.model_name_sanitized = re.sub(r"[\"<>:/\|\\?\*\[\]]+", "__", name)

    def _dir(self) -> Path:
        p = Path(self.output_path) / self.model_name_sanitized
        p.mkdir(parents=True, exist_ok=True)
        return p

    def save_results_aggregated(self, results: dict, samples: dict | None = None, datetime_str: str | None = None) -> Path | None:
        if not self.output_path:
            return None
        if datetime_str:
            self.date_id = datetime_str.replace(":", "-")
        out = {k: v for k, v in results.items() if k != "samples"}
        if samples:
            out["task_hashes"] = {t: utils.hash_string("".join(s["doc_hash"] + s["prompt_hash"] + s["target_hash"] for s in ss))
                                  for t, ss in samples.items()}
        out["total_evaluation_time_seconds"] = str(time.perf_counter() - self.start_time)
        f = self._dir() / f"{self.date_id}_results.json"
        f.write_text(json.dumps(out, indent=2, default=str, ensure_ascii=False), encoding="utf-8")
        return f

    def save_results_samples(self, task_name: str, samples: list[dict]) -> Path | None:
        if not self.output_path:
            return None
        f = self._dir() / f"{self.date_id}_samples_{task_name}.jsonl"
        with f.open("a", encoding="utf-8") as fh:
            for sample in samples:
                sample = dict(sample)
                args = sample["arguments"]
                sample["input"] = args[0]
                sample["arguments"] = {str(i): v for i, v in enumerate(args[1])} if len(args) > 1 and isinstance(args[1], (list, tuple)) else \
                    ({str(i): v for i, v in enumerate(args[1].items())} if len(args) > 1 and isinstance(args[1], dict) else {})
                sample["resps"] = utils.sanitize_list(sample["resps"])
                sample["filtered_resps"] = utils.sanitize_list(sample["filtered_resps"])
                sample["target"] = str(sample["target"])
                fh.write(json.dumps(sample, default=str, ensure_ascii=False) + "\n")
        return f
