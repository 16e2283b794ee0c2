"""Worker of the world-size-2 tests in tests/test_host_logic.py: runs simple_evaluate + the tracker under gloo with a stub
model (no GPU, CPU string metrics only); rank 0 writes the result files.  OWC_TEST_LINES=1 asks for `samples_as_lines` (the
owning rank serialises the samples-file line, as eval_model.py does)."""
import os
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

from lmms_owc_amd.engine.evaluate import simple_evaluate  # noqa: E402
from lmms_owc_amd.engine.tracker import EngineTracker  # noqa: E402
from lmms_owc_amd.tasks import ClassificationTask  # noqa: E402


class StubModel:
    """Deterministic stand-in for a Model plug-in: the answer depends only on the document."""

    device = "cpu"

    def __init__(self):
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.task_dict = {}

    def eval(self):
        return self

    def generate_until(self, requests):
        out = []
        for r in requests:
            ctx, gk, d2v, doc_id, task, split = r.args
            gk.pop("until", None)
            doc = self.task_dict[task][split][doc_id]
            out.append(f" {doc['target'].replace('_', ' ')} é" if doc_id % 3 else "something else, entirely longer than the others")
        return out


def main():
    world = int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("OWC_TEST_BACKEND", "gloo")   # "nccl" (= RCCL): tests/test_rccl_gpu.py, also with ONE rank
    if backend == "nccl":
        import torch

        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
        StubModel.device = f"cuda:{torch.cuda.current_device()}"
    elif world > 1:
        dist.init_process_group("gloo")
    docs = [{"visual": f"img{i}.jpg", "target": f"class_{i % 4}"} for i in range(11)]
    metrics = [{"metric": "exact_match", "aggregation": "mean", "ignore_case": True, "regexes_to_ignore": [",", "\\$"]},
               {"metric": "textual_inclusion", "aggregation": "mean"}]
    task = ClassificationTask("toy", docs, metric_list=metrics)
    task.doc_to_visual = lambda doc: []
    lm = StubModel()
    out_dir = Path(sys.argv[1])
    tracker = EngineTracker(output_path=str(out_dir))
    tracker.log_experiment_args(model_source="stub", model_args="")
    date = "2026-01-02T03:04:05"
    res = simple_evaluate(model="stub", task_objects={"toy": task}, limit=int(os.environ.get("OWC_TEST_LIMIT", "9")), model_object=lm,
                          datetime_str=date, samples_as_lines=os.environ.get("OWC_TEST_LINES") == "1")
    assert (res is None) == (lm.rank != 0)
    if res is not None:
        samples = res.pop("samples")
        tracker.general["start_time"] = 0.0   # the three time fields are the only run-dependent bytes of the results file
        tracker.save_results_aggregated(results=res, samples=samples, datetime_str=date)
        for name in res["configs"]:
            tracker.save_results_samples(task_name=name, samples=samples[name])
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
