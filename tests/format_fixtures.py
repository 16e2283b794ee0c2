"""Shared by tests/test_formats.py (CPU, scorer = numpy oracle double) and tests/test_plugin_gpu.py (GPU, HIP scorer):
the toy task + stand-in model of tools/gen_golden_formats.py rebuilt on THIS repo's engine, and comparison helpers
against the reference-run fixtures tests/golden/{engine_formats,eval_metrics,concept_similarity}.json."""
import json
from pathlib import Path

import numpy as np

from tests import recipes

GOLD = Path(__file__).parent / "golden"
TOY_METRICS = [
    {"metric": "concept_semantic_similarity", "aggregation": "concept_semantic_similarity", "higher_is_better": True},
    {"metric": "exact_match", "aggregation": "mean", "higher_is_better": True, "ignore_case": True, "ignore_punctuation": False,
     "regexes_to_ignore": [",", "\\$"]},
    {"metric": "semantic_similarity", "aggregation": "semantic_similarity", "higher_is_better": True},
    {"metric": "textual_inclusion", "aggregation": "mean", "higher_is_better": True},
]
DATE = "2026-01-02T03:04:05"
VOLATILE_RESULT_KEYS = {"git_hash", "start_time", "end_time", "total_evaluation_time_seconds"}
VOLATILE_CONFIG_KEYS = {"doc_to_visual", "doc_to_text", "doc_to_target"}   # function reprs with addresses


class StandInModel:
    """The deterministic stand-in the golden generator plugged into the reference (same answers, same `until` pop)."""

    rank, world_size = 0, 1
    chat_template = None

    def __init__(self):
        self.task_dict = {}

    def eval(self):
        return self

    def generate_until(self, requests):
        out = []
        for r in requests:
            ctx, gen_kwargs, d2v, doc_id, task, split = r.args
            gen_kwargs.pop("until", None)
            out.append(recipes.toy_answer(doc_id, self.task_dict[task][split][doc_id]["target"]))
        return out


    def generate_until_multi_round(self, requests):
        out = []
        for r in requests:
            ctx, gen_kwargs, d2v, d2t, doc_id, task, split = r.args
            gen_kwargs.pop("until", None)
            doc = self.task_dict[task][split][doc_id]
            out.append(recipes.toy_multi_round_answers(doc_id, doc["target"], d2t, doc))
        return out


TOY_MR_PROMPTS = ["What type of object is in this photo? Summarise.", "Caption the image.", "Reason about the caption.", "Give the final answer."]


def toy_task_mr():
    from lmms_owc_amd.tasks import ClassificationTask

    t = ClassificationTask("toymr", recipes.toy_docs(), output_type="generate_until_multi_round", prompts=list(TOY_MR_PROMPTS),
                           generation_kwargs={"max_new_tokens": 64, "do_sample": False}, metric_list=[dict(m) for m in TOY_METRICS])
    t.dataset_path = "data/toy"
    t.doc_to_visual = lambda doc: []
    return t


def toy_task():
    from lmms_owc_amd.tasks import ClassificationTask

    t = ClassificationTask("toytask", recipes.toy_docs(), generation_kwargs={"max_new_tokens": 64, "do_sample": False},
                           metric_list=[dict(m) for m in TOY_METRICS])
    t.dataset_path = "data/toy"
    t.doc_to_visual = lambda doc: []
    return t


def install_text_pipeline(scorer) -> None:
    """Tiny seeded BERT + HashTokenizer + the rule-based parser: what the generator injected into the reference."""
    from lmms_owc_amd.pipelines import text

    text.set_sentence_bert(scorer, recipes.HashTokenizer(recipes.bert_cfg("tiny")["vocab_size"]))
    text.set_concept_extractor(None)
    text.set_concept_nlp(lambda texts: [recipes.toy_nlp(t) for t in texts])


def run_engine(out_dir: Path, multi_round: bool = False) -> tuple[dict, dict]:
    """simple_evaluate + tracker exactly as eval_model.py drives them; returns (results, {file name: text})."""
    from lmms_owc_amd.engine.evaluate import simple_evaluate
    from lmms_owc_amd.engine.tracker import EngineTracker

    tracker = EngineTracker(output_path=str(out_dir))
    tracker.log_experiment_args(model_source="stand-in", model_args="", system_instruction=None, chat_template=None,
                                fewshot_as_multiturn=False)
    tasks = {"toymr": toy_task_mr()} if multi_round else {"toytask": toy_task()}
    res = simple_evaluate(model="stand-in", model_args="", task_objects=tasks, batch_size=1, limit=7,
                          model_object=StandInModel(), datetime_str=DATE)
    samples = res.pop("samples")
    tracker.save_results_aggregated(results=res, samples=samples, datetime_str=DATE)
    for task_name in res["configs"]:
        tracker.save_results_samples(task_name=task_name, samples=samples[task_name])
    return res, {p.name: p.read_text() for p in sorted(out_dir.rglob("*")) if p.is_file()}


def assert_same_json(got, want, tol: float, path: str = "") -> None:
    """Same types, same dict KEY ORDER, same strings / ints / bools; floats within `tol`."""
    if isinstance(want, dict):
        assert isinstance(got, dict) and list(got) == list(want), f"{path}: keys {list(got) if isinstance(got, dict) else got} != {list(want)}"
        for k in want:
            assert_same_json(got[k], want[k], tol, f"{path}.{k}")
    elif isinstance(want, list):
        assert isinstance(got, list) and len(got) == len(want), f"{path}: {got} != {want}"
        for i, (g, w) in enumerate(zip(got, want)):
            assert_same_json(g, w, tol, f"{path}[{i}]")
    elif isinstance(want, float) or (isinstance(got, float) and isinstance(want, (int, float)) and not isinstance(want, bool)):
        assert isinstance(got, (int, float)) and abs(float(got) - float(want)) <= tol, f"{path}: {got} != {want}"
    else:
        assert type(got) is type(want) and got == want, f"{path}: {got!r} != {want!r}"


def check_engine_files(files: dict, tol: float, multi_round: bool = False) -> None:
    gold = json.loads((GOLD / "engine_formats.json").read_text())
    gold, task = (gold["multi_round"]["files"], "toymr") if multi_round else (gold["files"], "toytask")
    assert sorted(files) == sorted(gold)
    sname = next(n for n in gold if "_samples_" in n)
    got_lines, want_lines = files[sname].splitlines(), gold[sname].splitlines()
    assert len(got_lines) == len(want_lines)
    for g, w in zip(got_lines, want_lines):
        assert g == w   # the samples file holds no scorer output: it must be BYTE-identical to the reference's
    rname = next(n for n in gold if n.endswith("_results.json"))
    got, want = json.loads(files[rname]), json.loads(gold[rname])
    assert list(got) == list(want), (list(got), list(want))
    for k in VOLATILE_RESULT_KEYS:
        got.pop(k), want.pop(k)
    for cfg in (got["configs"][task], want["configs"][task]):
        for k in VOLATILE_CONFIG_KEYS:
            cfg.pop(k)
    assert_same_json(got, want, tol, "results.json")
    # text-level layout of the results file: indent=2, same line count
    assert files[rname].count("\n") == gold[rname].count("\n") and files[rname].startswith('{\n  "results": {')


def check_eval_metrics(root: Path, capsys, tol: float, multi_round: bool = False) -> None:
    """Runs THIS repo's eval_metrics.main on the reference's `before` file; the rewritten JSONL and the printed table must
    equal the reference's `after` / stdout (columns, order, int-vs-float formatting; floats within tol)."""
    import os
    from argparse import Namespace

    import eval_metrics

    gold = json.loads((GOLD / "eval_metrics.json").read_text())
    task = "toymr" if multi_round else "toytask"
    if multi_round:
        gold = gold["multi_round"]
    d = root / "logs" / "schedule" / task / "stand-in"
    d.mkdir(parents=True)
    f = d / f"2026-01-02T03-04-05_samples_{task}.jsonl"
    f.write_text(gold["before"])
    cwd = os.getcwd()
    os.chdir(root)
    try:
        capsys.readouterr()
        eval_metrics.main(Namespace(input="logs/schedule", metrics=gold["metrics"], seed=1234, log_level="WARNING"))
        out = capsys.readouterr().out
    finally:
        os.chdir(cwd)
    assert out == gold["stdout"]
    got_lines, want_lines = f.read_text().splitlines(), gold["after"].splitlines()
    assert len(got_lines) == len(want_lines)
    for g, w in zip(got_lines, want_lines):
        assert_same_json(json.loads(g), json.loads(w), tol, "after")
        # pandas' compact separators / 10-digit floats: same textual shape (numbers aside)
        assert g.count(",") == w.count(",") and g.count(":") == w.count(":")


def check_concept_similarity(tol: float) -> None:
    from lmms_owc_amd.metrics import get_metric_info

    gold = json.loads((GOLD / "concept_similarity.json").read_text())
    items = [(r, p) for r, p in gold["items"]]
    assert [[r, p] for r, p in recipes.toy_concept_items()] == gold["items"]
    info = get_metric_info("concept_semantic_similarity")
    none = info.group_fn(info.builder_fn(items), reduce="none")
    assert [list(c) for c, _ in none] == gold["concepts"]          # the reference's post-processing of the parser's spans
    for (_, s), want in zip(none, gold["similarities"]):
        assert np.abs(np.array(s) - np.array(want)).max() <= tol
    for red in ("max", "mean", "median", "min"):
        assert abs(info.group_fn(info.builder_fn(items), reduce=red) - gold[red]) <= tol, red
