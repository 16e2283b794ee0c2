"""Drop-in file formats, the offline scorer CLI and concept similarity against fixtures produced by RUNNING THE REFERENCE
(tools/gen_golden_formats.py -> tests/golden/engine_formats.json, eval_metrics.json, concept_similarity.json), on CPU:
the host logic under test is this repo's engine / tracker / eval_metrics.py / metrics; the sentence encoder is replaced by
a TEST DOUBLE backed by the numpy oracle (the product has no CPU scorer - tests/test_plugin_gpu.py repeats every check
with the HIP scorer on the GPU)."""
import numpy as np
import pytest
import torch

from oracle import bert_np
from tests import format_fixtures as F
from tests import recipes


class OracleScorer:
    """Test double with SentenceScorer's interface; arithmetic = oracle/bert_np.py (fp32 numpy)."""

    def __init__(self):
        self.cfg = recipes.bert_cfg("tiny")
        self.w = recipes.bert_weights(self.cfg, 1234)

    def embed(self, ids, mask) -> torch.Tensor:
        return torch.from_numpy(bert_np.sentence_embed(self.w, self.cfg, np.asarray(ids), np.asarray(mask)).astype(np.float32))

    @staticmethod
    def paired_cosine(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        return (a * b).sum(-1)


@pytest.fixture()
def pipeline():
    from lmms_owc_amd.pipelines import text

    saved = (text.sentence_bert_model, text.sentence_bert_processor, text.concept_extractor, text.concept_nlp)
    F.install_text_pipeline(OracleScorer())
    yield
    text.sentence_bert_model, text.sentence_bert_processor, text.concept_extractor, text.concept_nlp = saved


def test_results_and_samples_files_match_reference_tracker(pipeline, tmp_path):
    _, files = F.run_engine(tmp_path / "out")
    F.check_engine_files(files, tol=2e-6)


def test_eval_metrics_cli_matches_reference(pipeline, tmp_path, capsys):
    F.check_eval_metrics(tmp_path, capsys, tol=2e-6)


def test_multi_round_task_files_match_reference_tracker(pipeline, tmp_path):
    """The same through a `generate_until_multi_round` task (the `*_llamav_o1` configs' type): 7-tuple requests, the tuple of round
    answers nested in `resps`, scoring on the LAST round (`_manager.py:1033-1036`) - results dict and samples file of the reference's
    own engine + tracker run (tools/gen_golden_formats.py)."""
    _, files = F.run_engine(tmp_path / "out", multi_round=True)
    F.check_engine_files(files, tol=2e-6, multi_round=True)


def test_eval_metrics_cli_on_a_multi_round_samples_file_matches_reference(pipeline, tmp_path, capsys):
    """eval_metrics.py:67-68 unwraps the nested multi-round responses before scoring."""
    F.check_eval_metrics(tmp_path, capsys, tol=2e-6, multi_round=True)


def test_concept_semantic_similarity_matches_reference(pipeline):
    F.check_concept_similarity(tol=2e-6)


def test_concept_postprocessing_quirks():
    from lmms_owc_amd.pipelines.text import postprocess_concepts

    skip = ["image", "it", "this photo"]
    # duplicates among noun chunks are kept, entities are de-duplicated, prefixes stripped once, skip words dropped
    assert postprocess_concepts(["The cat", "the cat", "an image", "its tail"], ["Cat", "Paris"], skip) == ["cat", "cat", "tail", "paris"]
    # without remove_prefix_words the reference records no noun chunk at all (the append sits inside that branch)
    assert postprocess_concepts(["The cat"], ["Paris"], skip, remove_prefix_words=False) == ["paris"]


def test_fractional_limit_is_ceiled_like_the_reference(pipeline):
    """`--limit 0.1` of 11 documents = ceil(1.1) = 2 documents (`_engine.py:125-126`), not floor."""
    from lmms_owc_amd.engine.evaluate import simple_evaluate

    res = simple_evaluate(model="stand-in", task_objects={"toytask": F.toy_task()}, limit=0.1, model_object=F.StandInModel())
    assert res["n-samples"]["toytask"] == {"original": 11, "effective": 2} and len(res["samples"]["toytask"]) == 2


def test_decoded_image_in_doc_never_reaches_the_samples_file(pipeline, tmp_path):
    """Docs of this repo may carry the decoded image under `visual` (the reference's rows hold a path there): the sample
    record keeps JSON-serialisable values only, so two runs write identical files and no object repr / address leaks."""
    import json

    from PIL import Image

    from lmms_owc_amd.engine.evaluate import simple_evaluate
    from lmms_owc_amd.engine.tracker import EngineTracker

    texts = []
    for run in range(2):
        task = F.toy_task()
        for d in task.docs:
            d["visual"] = Image.new("RGB", (8, 8))   # a fresh object (fresh address) per run
        res = simple_evaluate(model="stand-in", task_objects={"toytask": task}, limit=3, model_object=F.StandInModel(), datetime_str=F.DATE)
        tr = EngineTracker(output_path=str(tmp_path / f"run{run}"))
        tr.save_results_samples(task_name="toytask", samples=res["samples"]["toytask"])
        (f,) = (tmp_path / f"run{run}").rglob("*_samples_*.jsonl")
        texts.append(f.read_text())
    assert texts[0] == texts[1] and "PIL" not in texts[0] and " at 0x" not in texts[0]
    for line in texts[0].splitlines():
        assert "visual" not in json.loads(line)["doc"] and "target" in json.loads(line)["doc"]
