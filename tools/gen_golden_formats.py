#!/usr/bin/env python3
"""Golden fixtures for the drop-in FILE FORMATS and the offline scorer CLI, produced by RUNNING THE REFERENCE in this
container (SURVEY.md §8c G9 + rows a19 / a20 / f1).  Writes

  tests/golden/engine_formats.json   the reference's `simple_evaluate` -> `evaluate` -> `EngineTracker` on a toy classification
                                     task with a deterministic stand-in model: results dict, `*_results.json` text,
                                     `*_samples_<task>.jsonl` text (src/engine/_engine.py:32-637, _tracker.py:220-341)
  tests/golden/eval_metrics.json     the reference's `eval_metrics.main` on that samples file: the file before / after
                                     (added columns, float formatting) and the printed table (eval_metrics.py:19-171)
  tests/golden/concept_similarity.json  the reference's `concept_semantic_similarity` (src/data/metrics/_group.py:176-334)
                                     with an injected rule-based noun chunker in place of spaCy's en_core_web_lg

Only third-party pieces that are absent offline are replaced: stub modules for gdown / wandb / spacy / ... (never on the
executed path), `src.models` (its wrappers need torchvision / llava / qwen_vl_utils; the stand-in model below honours
the same plug-in contract), the MiniLM checkpoint (seeded BERT weights of tests/recipes.py) and its tokenizer
(tests/recipes.HashTokenizer).  Everything else that runs is the reference's own code.

    python tools/gen_golden_formats.py        # needs /root/reference; the fixtures travel, the reference does not
"""

from __future__ import annotations

import contextlib
import importlib
import importlib.machinery
import importlib.util
import io
import json
import os
import sys
import tempfile
import types
from argparse import Namespace
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import recipes  # noqa: E402
from tools import gen_golden as G  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference")

TOY_YAML = """task: "toytask"
model_specific_kwargs:
  default:
    pre_prompt: ""
    prompt: "What type of object is in this photo?"
    post_prompt: ""
generation_kwargs:
  max_new_tokens: 64
  do_sample: False
dataset_path: data/toy
dataset_kwargs:
  load_from_disk: true
doc_to_visual: !function toy_utils.doc_to_visual
doc_to_text: !function toy_utils.doc_to_text
doc_to_target: !function toy_utils.doc_to_target
output_type: generate_until
test_split: test
metric_list:
  - metric: concept_semantic_similarity
    aggregation: concept_semantic_similarity
    higher_is_better: true
  - metric: exact_match
    aggregation: mean
    higher_is_better: true
    ignore_case: true
    ignore_punctuation: false
    regexes_to_ignore:
      - ","
      - "\\\\$"
  - metric: semantic_similarity
    aggregation: semantic_similarity
    higher_is_better: true
  - metric: textual_inclusion
    aggregation: mean
    higher_is_better: true
metadata:
  - version: 0.0
"""

TOY_MR_PROMPTS = ["What type of object is in this photo? Summarise.", "Caption the image.", "Reason about the caption.", "Give the final answer."]
TOY_MR_YAML = TOY_YAML.replace('task: "toytask"', 'task: "toymr"').replace(
    '    prompt: "What type of object is in this photo?"\n', "    prompts:\n" + "".join(f'      - "{q}"\n' for q in TOY_MR_PROMPTS)).replace(
    "doc_to_text: !function toy_utils.doc_to_text\n", "doc_to_text: !function toy_utils.doc_to_text_multi_round\n").replace(
    "output_type: generate_until\n", "output_type: generate_until_multi_round\n")

TOY_UTILS = '''
def doc_to_visual(doc):
    return []


def doc_to_text_multi_round(doc, model_specific_kwargs=None, round_idx=None, previous_round_results=None, last_round_info=None):
    k = model_specific_kwargs or {}
    prompts = k["prompts"]
    if round_idx is None:
        return k.get("pre_prompt", "") + prompts[0] + k.get("post_prompt", "")
    if round_idx < len(prompts):
        return None, k.get("pre_prompt", "") + prompts[round_idx] + k.get("post_prompt", ""), False, previous_round_results or [], last_round_info
    return None, None, True, previous_round_results or [], last_round_info


def doc_to_text(doc, model_specific_kwargs=None):
    k = model_specific_kwargs or {}
    return k.get("pre_prompt", "") + k.get("prompt", "") + k.get("post_prompt", "")


def doc_to_target(doc):
    return doc["target"].replace("_", " ")
'''


class FakeSpacy:
    """spaCy stand-in for `concept_extraction_spacy` (_text.py:18-140): `.pipe(texts, batch_size=)` yields docs whose
    `.noun_chunks` / `.ents` come from tests/recipes.toy_nlp (the rule-based chunker the GPU test plugs in as well)."""

    class _Span:
        def __init__(self, text):
            self.text = text

    class _Doc:
        def __init__(self, chunks, ents):
            self.noun_chunks = [FakeSpacy._Span(c) for c in chunks]
            self.ents = [FakeSpacy._Span(e) for e in ents]

    def pipe(self, texts, batch_size=None):
        return [self._Doc(*recipes.toy_nlp(t)) for t in texts]

    def __call__(self, text):
        return self._Doc(*recipes.toy_nlp(text))


def setup_reference():
    metrics, text_mod, utils = G.import_reference()
    m = types.ModuleType("src.models")
    m.__spec__ = importlib.machinery.ModuleSpec("src.models", None)

    class Model:  # the ABC's role only: the engine type-hints it
        pass

    class StandInModel(Model):
        """Plug-in contract of src/models/_base.py + _qwen2_vl.py:143-348 without the arithmetic: deterministic answers,
        `until` popped from the request's gen_kwargs like the reference wrapper does (_qwen2_vl.py:211-219)."""

        rank, world_size = 0, 1
        chat_template = None

        def __init__(self, **kw):
            self.task_dict = {}
            self.kw = kw

        def eval(self):
            return self

        def generate_until(self, requests):
            out = []
            for r in requests:
                ctx, gen_kwargs, d2v, doc_id, task, split = r.args
                gen_kwargs.pop("until", None)
                doc = self.task_dict[task][split][doc_id]
                out.append(recipes.toy_answer(doc_id, doc["target"]))
            return out

        def generate_until_multi_round(self, requests):
            """The wrapper's side of the protocol (_qwen2_vl.py:434-603) without a model: rounds until the task's terminal signal, the
            last round answers like `generate_until`, the earlier ones with a round tag."""
            out = []
            for r in requests:
                ctx, gen_kwargs, d2v, d2t, doc_id, task, split = r.args
                gen_kwargs.pop("until", None)
                doc = self.task_dict[task][split][doc_id]
                out.append(recipes.toy_multi_round_answers(doc_id, doc["target"], d2t, doc))
            return out

    m.Model, m.get_model = Model, (lambda name, **kw: StandInModel(**kw))
    sys.modules["src.models"] = m
    c = recipes.bert_cfg("tiny")
    text_mod.sentence_bert_model = G.hf_bert(c, recipes.bert_weights(c, 1234))
    text_mod.sentence_bert_processor = recipes.HashTokenizer(c["vocab_size"])
    text_mod.spacy_model = FakeSpacy()
    return metrics, text_mod, utils


def run_engine(tmp: Path, task: str = "toytask", yaml_text: str = TOY_YAML) -> dict:
    import datasets

    import src.engine as E
    from src.data.tasks import TaskManager

    docs = recipes.toy_docs()
    if not (tmp / "data" / "toy").exists():
        datasets.DatasetDict({"test": datasets.Dataset.from_list(docs)}).save_to_disk(str(tmp / "data" / "toy"))
    tdir = tmp / "tasks" / task
    tdir.mkdir(parents=True)
    (tdir / f"{task}.yaml").write_text(yaml_text)
    (tdir / "toy_utils.py").write_text(TOY_UTILS)
    out_dir = tmp / "logs" / "schedule" / task / "stand-in"
    tracker = E.EngineTracker(output_path=str(out_dir))
    tm = TaskManager(include_path=str(tmp / "tasks" / task), include_defaults=False, model_name="stand-in")
    date = "2026-01-02T03:04:05"
    res = E.simple_evaluate(model_name="stand-in", model_args="", tasks=[task], batch_size=1, limit=7, bootstrap_iters=100000,
                            log_samples=True, engine_tracker=tracker, task_manager=tm, datetime_str=date,
                            cli_args=Namespace(process_with_media=False, output_path=str(out_dir)))
    samples = res.pop("samples")
    tracker.save_results_aggregated(results=res, samples=samples, datetime_str=date)
    for task_name in res["configs"]:
        tracker.save_results_samples(task_name=task_name, samples=samples[task_name])
    files = {p.name: p.read_text() for p in sorted(out_dir.rglob("*")) if p.is_file()}
    return {"results": json.loads(json.dumps(res, default=str)), "files": files, "out_dir_rel": str(out_dir.relative_to(tmp))}


def run_eval_metrics(tmp: Path, samples_name: str, task: str = "toytask") -> dict:
    spec = importlib.util.spec_from_file_location("ref_eval_metrics", REF / "eval_metrics.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    f = tmp / "logs" / "schedule" / task / "stand-in" / samples_name
    before = f.read_text()
    buf = io.StringIO()
    metrics = "semantic_similarity,textual_inclusion,mean_average_semantic_similarity,concept_semantic_similarity"
    with contextlib.redirect_stdout(buf):
        mod.main(Namespace(input=f"logs/schedule/{task}", metrics=metrics, seed=1234, log_level="WARNING"))
    return {"metrics": metrics, "before": before, "after": f.read_text(), "stdout": buf.getvalue()}


def run_concept(metrics) -> dict:
    info = metrics.get_metric_info("concept_semantic_similarity")
    items = recipes.toy_concept_items()
    out = {"items": [[r, p] for r, p in items]}
    none = info.group_fn(info.builder_fn(items), reduce="none")
    out["concepts"] = [list(c) for c, _ in none]
    out["similarities"] = [list(map(float, s)) for _, s in none]
    for red in ("max", "mean", "median", "min"):
        out[red] = float(info.group_fn(info.builder_fn(items), reduce=red))
    return out


def main():
    metrics, text_mod, utils = setup_reference()
    saved = torch.cuda.is_available
    torch.cuda.is_available = lambda: False   # the reference's CPU fp32 scoring branch (_text.py:165-170)
    cwd = os.getcwd()
    try:
        with tempfile.TemporaryDirectory() as td:
            tmp = Path(td)
            os.chdir(tmp)
            eng = run_engine(tmp)
            sname = next(n for n in eng["files"] if "_samples_" in n)
            em = run_eval_metrics(tmp, sname)
            # the same through a MULTI-ROUND task (output_type generate_until_multi_round: 7-tuple requests, last-round scoring
            # _manager.py:1033-1036, nested `resps`, eval_metrics.py:67-68's unwrap)
            eng["multi_round"] = run_engine(tmp, "toymr", TOY_MR_YAML)
            em["multi_round"] = run_eval_metrics(tmp, next(n for n in eng["multi_round"]["files"] if "_samples_" in n), "toymr")
        conc = run_concept(metrics)
    finally:
        os.chdir(cwd)
        torch.cuda.is_available = saved
    meta = {"versions": G.versions()}
    (GOLD / "engine_formats.json").write_text(json.dumps({**meta, **eng}, indent=1))
    (GOLD / "eval_metrics.json").write_text(json.dumps({**meta, **em}, indent=1))
    (GOLD / "concept_similarity.json").write_text(json.dumps({**meta, **conc}, indent=1))
    print("wrote engine_formats.json, eval_metrics.json, concept_similarity.json")
    print(em["stdout"])


if __name__ == "__main__":
    main()
