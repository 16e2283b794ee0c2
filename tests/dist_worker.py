"""Worker of test_evaluate_two_ranks_equals_one: runs the evaluate loop under gloo with a stub model
(no GPU, CPU string metrics only) and lets rank 0 write the gathered result."""
import json
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch.distributed as dist  # noqa: E402

from lmms_owc_amd.engine.evaluate import evaluate  # noqa: E402
from lmms_owc_amd.tasks import ClassificationTask  # noqa: E402


class StubModel:
    """Deterministic stand-in for a Model plug-in: answer depends only on the document."""

    def __init__(self):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.task_dict = {}

    def generate_until(self, requests):
        out = []
        for r in requests:
            ctx, gk, d2v, doc_id, task, split = r.args
            doc = self.task_dict[task][split][doc_id]
            out.append(f" {doc['target'].replace('_', ' ')} " if doc_id % 3 else "something else")
        return out


def main():
    world = int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo")
    docs = [{"visual": f"img{i}.jpg", "target": f"class_{i % 4}"} for i in range(11)]
    metrics = [{"metric": "exact_match", "aggregation": "mean", "ignore_case": True, "regexes_to_ignore": [",", "\\$"]},
               {"metric": "textual_inclusion", "aggregation": "mean"}]
    task = ClassificationTask("toy", docs, metric_list=metrics)
    task.doc_to_visual = lambda doc: []
    lm = StubModel()
    lm.task_dict["toy"] = task.dataset
    res = evaluate(lm, {"toy": task}, limit=int(os.environ.get("OWC_TEST_LIMIT", "9")))
    if res is not None:
        Path(sys.argv[1]).write_text(json.dumps({"results": {k: (v if isinstance(v, str) else float(v)) for k, v in res["results"]["toy"].items()},
                                                 "samples": res["samples"]["toy"]}, default=float))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
