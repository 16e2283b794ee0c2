"""End-to-end drop-in path on the GPU: model registry -> batched generate_until -> evaluate loop ->
samples JSONL -> eval_metrics.py offline scorer (sentence encoder on the HIP scorer)."""
import json
import re
from argparse import Namespace

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class HashTokenizer:
    """Word -> id by crc32 (no tokenizer files offline); pad-to-longest like AutoTokenizer(padding=True)."""

    def __call__(self, text, padding=True, truncation=True, return_tensors="np"):
        import zlib

        rows = [[101] + [1000 + zlib.crc32(w.encode()) % 20000 for w in re.findall(r"\w+", t.lower())][:30] + [102] for t in text]
        L = max(len(r) for r in rows)
        ids = np.array([r + [0] * (L - len(r)) for r in rows], dtype=np.int64)
        mask = np.array([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=np.int64)
        return {"input_ids": ids, "attention_mask": mask}


@pytest.fixture(scope="module")
def scorer(gpu):
    from lmms_owc_amd.engine.scorer import MINILM_L6, BertWeights, SentenceScorer
    from lmms_owc_amd.pipelines import text

    sc = SentenceScorer(BertWeights.random(MINILM_L6, gpu, seed=3))
    text.set_sentence_bert(sc, HashTokenizer())
    return sc


def test_generate_until_batch_invariant_and_ordered(gpu):
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    task = load_task("synthetic:7:56x84:3")
    task.build_all_requests(limit=None, rank=0, world_size=1)
    outs = []
    for bs in (1, 4):
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs)
        lm.task_dict[task.task_name] = task.dataset
        outs.append(lm.generate_until(task.instances))
    assert len(outs[0]) == 7 and outs[0] == outs[1]
    assert lm.generate_until([]) == []  # empty shard
    with pytest.raises(ValueError):
        get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", bogus=1)


def test_generate_until_with_straggler_hand_over_equals_without(gpu):
    """`generate_until` over several engine passes with an EOS that many answers hit early and some never: the stragglers of a
    pass finish inside the following passes (Qwen2VLEngine.generate `carry`), and every answer equals the one the same plug-in
    gives with the hand-over switched off."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    outs = {}
    for tag in ("carry", "plain"):
        task = load_task("synthetic:90:56x84:3")
        task.generation_kwargs.update({"max_new_tokens": 24})
        task.build_all_requests(limit=None, rank=0, world_size=1)
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=1, engine_batch=24)
        lm.task_dict[task.task_name] = task.dataset
        if tag == "carry":   # first run: find a token that ends many answers early, use it as EOS in both runs
            lm._no_carry = True
            free = lm.generate_until_tokens(task.instances)[0]
            vals, counts = np.unique(free[:, 2:12], return_counts=True)
            eos = int(vals[np.argmax(counts)])
            lm._no_carry = False
            task.build_all_requests(limit=None, rank=0, world_size=1)
        else:
            lm._no_carry = True
        lm._tokenizer.eos_token_id = eos
        outs[tag] = lm.generate_until(task.instances)
        if tag == "carry":
            assert lm.last_timing["chunks"] >= 3
    assert outs["carry"] == outs["plain"] and len(outs["carry"]) == 90
    assert len({len(a) for a in outs["carry"]}) > 3      # really ragged


def test_repetition_penalty_through_the_plug_in_over_several_passes(gpu):
    """Round 6: the plug-in's engine carries the checkpoint's `repetition_penalty` (what `Qwen2VL.load_model` reads from
    generation_config.json; here set on the synthetic model).  Answers with the penalty differ from the plain greedy ones, do not depend
    on the batch size or on how the documents are cut into engine passes (the straggler hand-over is switched off under a penalty: a
    pass runs its sequences to the end), and a second run reproduces them."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    outs = {}
    for tag, bs, eb, pen in (("plain", 4, 24, 1.0), ("a", 1, 24, 1.3), ("b", 4, 7, 1.3), ("c", 4, 24, 1.3)):
        task = load_task("synthetic:40:56x84:3")
        task.generation_kwargs.update({"max_new_tokens": 12})
        task.build_all_requests(limit=None, rank=0, world_size=1)
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs, engine_batch=eb)
        lm._model.repetition_penalty = pen
        lm.task_dict[task.task_name] = task.dataset
        outs[tag] = lm.generate_until(task.instances)
    assert outs["a"] == outs["b"] == outs["c"] and len(outs["a"]) == 40
    assert outs["a"] != outs["plain"]


def test_sampled_generate_until_through_gen_kwargs(gpu):
    """`--gen_kwargs temperature=0.8,top_p=0.9` (reference src/models/_qwen2_vl.py:308-329: do_sample = temperature > 0) reaches the
    on-device sampler: the answers differ from the greedy ones, are reproducible under the same torch seed, do not depend on the
    batch size (one random stream per document) and change with the seed; beams raise a clear error."""
    import torch

    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    outs = {}
    for tag, bs, seed, gk in (("greedy", 4, 1234, {}), ("a", 1, 1234, {"temperature": 0.8, "top_p": 0.9}),
                              ("b", 4, 1234, {"temperature": 0.8, "top_p": 0.9}), ("c", 4, 99, {"temperature": 0.8, "top_p": 0.9})):
        torch.manual_seed(seed)
        task = load_task("synthetic:9:56x84:3")
        task.generation_kwargs.update(gk)
        task.build_all_requests(limit=None, rank=0, world_size=1)
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs)
        lm.task_dict[task.task_name] = task.dataset
        outs[tag] = lm.generate_until(task.instances)
    assert outs["a"] == outs["b"] and outs["a"] != outs["greedy"] and outs["c"] != outs["a"]
    task = load_task("synthetic:3:56x84:3")
    task.generation_kwargs.update({"num_beams": 4, "temperature": 0.7})
    task.build_all_requests(limit=None, rank=0, world_size=1)
    lm.task_dict[task.task_name] = task.dataset
    with pytest.raises(NotImplementedError, match="beam SAMPLING"):
        lm.generate_until(task.instances)


@pytest.mark.parametrize("model_type", ["qwen2-vl", "llava"])
def test_num_beams_reaches_the_beam_search(gpu, model_type):
    """`--gen_kwargs num_beams=3` (reference src/models/_qwen2_vl.py:308-329, _llava_hf.py:365-376 hand it to HF's generate) runs
    `generate_beam`: the answers equal the engine-level call on the same prompts, do not depend on the plug-in's batch size, and the
    one-beam requests of the same task still take the greedy path."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    outs, calls = {}, []
    for tag, bs, gk in (("greedy", 4, {}), ("b1", 1, {"num_beams": 3}), ("b4", 4, {"num_beams": 3})):
        task = load_task("synthetic:7:56x84:3")
        task.generation_kwargs.update(gk)
        task.build_all_requests(limit=None, rank=0, world_size=1)
        lm = get_model("custom-model", model_type=model_type, model_name_or_path="synthetic:tiny", batch_size=bs)
        lm.task_dict[task.task_name] = task.dataset
        eng = lm._model
        orig = eng.generate_beam

        def spy(*a, _orig=orig, _tag=tag, **kw):
            calls.append((_tag, len(a[0]), a[4]))
            return _orig(*a, **kw)

        eng.generate_beam = spy
        outs[tag] = lm.generate_until(task.instances)
    assert outs["b1"] == outs["b4"] and len(outs["b4"]) == 7
    assert [c for c in calls if c[0] == "greedy"] == [] and sum(c[1] for c in calls if c[0] == "b4") == 7
    assert all(c[2] == 3 for c in calls)


def test_evaluate_then_offline_metrics(gpu, scorer, tmp_path, capsys):
    import eval_metrics
    from lmms_owc_amd.engine.evaluate import simple_evaluate
    from lmms_owc_amd.engine.tracker import EngineTracker
    from lmms_owc_amd.pipelines.text import encode_sentence_bert

    res = simple_evaluate(model="custom-model", model_args="model_type=qwen2-vl,model_name_or_path=synthetic:tiny",
                          tasks=["synthetic:6:84x56:3"], batch_size=4, limit=5)
    r = res["results"]["synthetic"]
    assert set(r) >= {"exact_match,none", "semantic_similarity,none", "textual_inclusion,none", "exact_match_stderr,none"}
    assert -1.0 <= r["semantic_similarity,none"] <= 1.0 and r["semantic_similarity_stderr,none"] == "N/A"
    samples = res["samples"]["synthetic"]
    assert [s["doc_id"] for s in samples] == [0, 1, 2, 3, 4]
    out_dir = tmp_path / "logs" / "schedule" / "synthetic" / "tiny-model"
    tr = EngineTracker(output_path=str(out_dir))
    tr.log_experiment_args(model_source="custom-model", model_args="")
    tr.save_results_aggregated({k: v for k, v in res.items() if k != "samples"}, res["samples"], "2026-01-01T00:00:00")
    f = tr.save_results_samples("synthetic", samples)
    rec = json.loads(f.read_text().splitlines()[0])
    assert list(rec)[:6] == ["doc_id", "doc", "target", "arguments", "resps", "filtered_resps"] and "input" in rec
    # offline re-scoring rewrites the JSONL in place with per-sample columns and prints the table
    out = eval_metrics.main(Namespace(input=str(tmp_path / "logs" / "schedule"), metrics="semantic_similarity,textual_inclusion,mean_average_semantic_similarity",
                                      seed=1234, log_level="WARNING"))
    rec = json.loads(f.read_text().splitlines()[0])
    assert "semantic_similarity" in rec and "semantic_similarity@0.5" in rec
    per_sample = [json.loads(l)["semantic_similarity"] for l in f.read_text().splitlines()]
    assert abs(np.mean(per_sample) - out["synthetic"]["tiny-model"]["semantic_similarity"]) < 1e-6
    assert "tiny-model" in capsys.readouterr().out
    # the hook contract of encode_sentence_bert: adds a list[list[float]] column of unit vectors
    b = encode_sentence_bert({"reference": ["sea lion", "a dog"]}, input_column="reference")
    z = np.array(b["reference_sentence_bert_embeds"])
    assert z.shape == (2, 384) and np.allclose(np.linalg.norm(z, axis=1), 1.0, atol=1e-5)
    # paired cosine of identical strings is 1
    from lmms_owc_amd.metrics import get_metric_info

    ss = get_metric_info("semantic_similarity")
    vals = ss.group_fn(ss.builder_fn([("sea lion", ["x", "sea lion"]), (["golden retriever"], "golden retriever")]), reduce="none")
    assert np.allclose(vals, 1.0, atol=1e-5)


def test_eval_model_cli_end_to_end(gpu, scorer, tmp_path, capsys):
    """eval_model.py with the reference's flag names: writes {date}_results.json + {date}_samples_{task}.jsonl."""
    import eval_model

    out = tmp_path / "logs" / "schedule" / "synthetic" / "qwen2-vl-tiny"
    eval_model.main(["--model", "custom-model", "--model_args", "model_type=qwen2-vl,model_name_or_path=synthetic:tiny",
                     "--tasks", "synthetic:5:56x56:2", "--batch_size", "3", "--limit", "4", "--log_samples",
                     "--output_path", str(out), "--verbosity", "DEBUG", "--gen_kwargs", "max_new_tokens=4"])
    # `model_name_or_path=synthetic:tiny` contains "path=": like the reference's GeneralConfigTracker._get_model_name
    # (_tracker.py:56-80) the run lands in the sub-directory of the sanitised name
    sub = out / "synthetic__tiny"
    files = sorted(p.name for p in sub.iterdir())
    assert any(f.endswith("_results.json") for f in files) and any("_samples_synthetic.jsonl" in f for f in files)
    res = json.loads(next(sub.glob("*_results.json")).read_text())
    assert "synthetic" in res["results"] and "total_evaluation_time_seconds" in res and res["model_name"] == "synthetic:tiny"
    assert len(next(sub.glob("*_samples_*.jsonl")).read_text().splitlines()) == 4
    assert "| synthetic |" in capsys.readouterr().out


def test_concept_semantic_similarity(gpu, scorer):
    """Contract of the reference's concept_semantic_similarity with a plugged-in extractor."""
    from lmms_owc_amd.metrics import get_metric_info
    from lmms_owc_amd.pipelines import text

    def bigrams(batch, skip):  # stand-in for spaCy noun chunks: every sliding word pair
        out = []
        for t in batch:
            w = re.findall(r"[a-z]+", t.lower())
            out.append([f"{a} {b}" for a, b in zip(w, w[1:]) if f"{a} {b}" not in skip])
        return out

    text.set_concept_extractor(bigrams)
    info = get_metric_info("concept_semantic_similarity")
    items = [("sea lion", ["it is a sea lion on a rock"]), (["golden retriever"], "a photo of a golden retriever dog")]
    rows = info.group_fn(info.builder_fn(items), reduce="none")
    assert rows[0][0][-1] == "it is a sea lion on a rock" and len(rows[0][0]) == len(rows[0][1])
    assert any(abs(v - 1.0) < 1e-5 for v in rows[0][1])  # the chunk "sea lion" matches the reference exactly
    mx = info.group_fn(info.builder_fn(items), reduce="max")
    mn = info.group_fn(info.builder_fn(items), reduce="min")
    assert mn <= info.group_fn(info.builder_fn(items), reduce="mean") <= mx <= 1.0 + 1e-5
    assert abs(mx - np.mean([max(r[1]) for r in rows])) < 1e-6


# ---------------------------------------------------------------- drop-in formats vs the reference's own output, HIP scorer
@pytest.fixture()
def ref_pipeline(gpu):
    """What tools/gen_golden_formats.py injected into the reference, on the HIP scorer: tiny seeded BERT, HashTokenizer,
    rule-based parser.  Restores the module state afterwards (other tests use the MiniLM-shaped `scorer` fixture)."""
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer
    from lmms_owc_amd.pipelines import text
    from tests import format_fixtures as F
    from tests import recipes

    saved = (text.sentence_bert_model, text.sentence_bert_processor, text.concept_extractor, text.concept_nlp)
    c = recipes.bert_cfg("tiny")
    F.install_text_pipeline(SentenceScorer(BertWeights(c, recipes.bert_weights(c, 1234), gpu)))
    yield F
    text.sentence_bert_model, text.sentence_bert_processor, text.concept_extractor, text.concept_nlp = saved


def test_engine_files_match_reference_tracker_gpu(ref_pipeline, tmp_path):
    """simple_evaluate + EngineTracker with the HIP scorer behind the group metrics: samples JSONL byte-identical to the
    file the REFERENCE's engine + tracker wrote for the same task and answers, results JSON equal key by key with the
    scorer's values within 2e-5 (tests/golden/engine_formats.json)."""
    _, files = ref_pipeline.run_engine(tmp_path / "out")
    ref_pipeline.check_engine_files(files, tol=2e-5)


def test_eval_metrics_cli_matches_reference_gpu(ref_pipeline, tmp_path, capsys):
    """G9: eval_metrics.py on the reference's samples file -> same rewritten columns / formatting / printed table as the
    reference's eval_metrics.main produced (tests/golden/eval_metrics.json), sentence metrics on the HIP scorer."""
    ref_pipeline.check_eval_metrics(tmp_path, capsys, tol=2e-5)


def test_multi_round_engine_files_and_eval_metrics_match_reference_gpu(ref_pipeline, tmp_path, capsys):
    """The `generate_until_multi_round` task type through the same two checks (nested round answers, last-round scoring,
    eval_metrics.py's unwrap), on the HIP scorer."""
    _, files = ref_pipeline.run_engine(tmp_path / "out", multi_round=True)
    ref_pipeline.check_engine_files(files, tol=2e-5, multi_round=True)
    ref_pipeline.check_eval_metrics(tmp_path / "em", capsys, tol=2e-5, multi_round=True)


def test_concept_semantic_similarity_matches_reference_gpu(ref_pipeline):
    """The reference's concept_semantic_similarity with the injected parser: same concept lists, similarities and the four
    reductions within 2e-5 (tests/golden/concept_similarity.json)."""
    ref_pipeline.check_concept_similarity(tol=2e-5)


def test_real_checkpoint_loading_path(gpu, tmp_path):
    """An on-disk HF-style checkpoint (config.json, sharded safetensors, fast tokenizer with a chat template) goes
    through the same loader a real Qwen2-VL directory would; generation must equal the engine fed directly."""
    import torch

    from lmms_owc_amd.engine.qwen2vl import Qwen2VLEngine, Qwen2VLWeights
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.models._qwen2_vl import dims_from_hf_config
    from lmms_owc_amd.tasks import load_task
    from tests import ckpt_util

    d = tmp_path / "Qwen2-VL-tiny"
    info = ckpt_util.write_qwen2vl_checkpoint(d, legacy_names=True)
    lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path=str(d), batch_size=2)
    assert lm.eot_token_id == info["specials"]["<|im_end|>"] and lm.chat_template
    task = load_task("synthetic:3:56x56:2")
    task.build_all_requests(limit=None, rank=0, world_size=1)
    lm.task_dict[task.task_name] = task.dataset
    out = lm.generate_until(task.instances)
    assert len(out) == 3 and all(isinstance(x, str) for x in out)
    # prompt built through the tokenizer's chat template: specials + one image_pad run of the right length
    ids = lm._prompt_ids("What type of object is in this photo?", [4])
    s = info["specials"]
    assert (ids == s["<|image_pad|>"]).sum() == 4 and ids[0] == s["<|im_start|>"] and (ids == s["<|vision_start|>"]).sum() == 1
    # same weights fed directly
    dims = dims_from_hf_config(json.loads((d / "config.json").read_text()))
    eng = Qwen2VLEngine(Qwen2VLWeights.from_state_dict(dims, info["weights"], gpu))
    from lmms_owc_amd import ops
    from lmms_owc_amd.models import imageproc

    img = imageproc.prepare_image(task.docs[0]["visual"], 4 * 28 * 28, 1024 * 28 * 28)
    pix = ops.patchify_u8(torch.from_numpy(img[None]).to(gpu), imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD)
    emb = eng.encode_images(pix, [(1, 4, 4)])
    toks = eng.generate([ids], emb, [[(1, 4, 4)]], 8, eos_token_id=s["<|im_end|>"], pad_token_id=s["<|endoftext|>"]).cpu().numpy()[0]
    stop = np.flatnonzero(toks == s["<|im_end|>"])
    direct = lm.tokenizer.batch_decode([toks[: stop[0]] if len(stop) else toks], skip_special_tokens=True)[0]
    single = lm.generate_until([task.instances[0]])
    assert single[0] == direct
    # generation_config.json of the checkpoint (what HF merges into the reference's generate call): top_k for sampled requests and
    # repetition_penalty for EVERY request reach the engine; a checkpoint without the file runs plain greedy (above)
    assert lm._model.repetition_penalty == 1.0
    (d / "generation_config.json").write_text(json.dumps({"do_sample": True, "temperature": 0.01, "top_p": 0.001, "top_k": 1,
                                                          "repetition_penalty": 1.05, "eos_token_id": [s["<|im_end|>"], s["<|endoftext|>"]]}))
    lm2 = get_model("custom-model", model_type="qwen2-vl", model_name_or_path=str(d), batch_size=2)
    assert lm2._model.repetition_penalty == 1.05 and lm2._default_top_k == 1
    lm2.task_dict[task.task_name] = task.dataset
    toks2 = eng.generate([ids], emb, [[(1, 4, 4)]], 8, eos_token_id=s["<|im_end|>"], pad_token_id=s["<|endoftext|>"],
                         repetition_penalty=1.05).cpu().numpy()[0]
    stop2 = np.flatnonzero(toks2 == s["<|im_end|>"])
    direct2 = lm2.tokenizer.batch_decode([toks2[: stop2[0]] if len(stop2) else toks2], skip_special_tokens=True)[0]
    assert lm2.generate_until([task.instances[0]])[0] == direct2


def test_sentence_encoder_loads_from_directory(gpu, tmp_path, monkeypatch):
    from lmms_owc_amd.pipelines import text
    from oracle import bert_np
    from tests import ckpt_util

    d = tmp_path / "minilm"
    info = ckpt_util.write_bert_checkpoint(d)
    monkeypatch.setenv("OWC_SENTENCE_BERT_PATH", str(d))
    monkeypatch.setattr(text, "sentence_bert_model", None)
    monkeypatch.setattr(text, "sentence_bert_processor", None)
    out = text.encode_sentence_bert({"text": ["sea lion", "what type of dog is this ?"]})
    z = np.array(out["text_sentence_bert_embeds"], dtype=np.float32)
    enc = text.sentence_bert_processor(["sea lion", "what type of dog is this ?"], padding=True, truncation=True, return_tensors="np")
    want = bert_np.sentence_embed(info["weights"], info["cfg"], np.asarray(enc["input_ids"]), np.asarray(enc["attention_mask"]))
    np.testing.assert_allclose(z, want, atol=2e-5)


@pytest.mark.parametrize("name", ["tiny", "tiny-next"])
def test_llava_generate_until_batch_invariant_and_ordered(gpu, name):
    """LLaVA plug-in (1.5 and NeXT/anyres): batching must not change any answer; answers come back in request order."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    task = load_task("synthetic:6:70x120:3")   # non-square images: anyres picks a 2-tile canvas, 1.5 crops
    task.build_all_requests(limit=None, rank=0, world_size=1)
    outs = []
    for bs in (1, 4):
        lm = get_model("custom-model", model_type="llava", model_name_or_path=f"synthetic:{name}", batch_size=bs)
        lm.task_dict[task.task_name] = task.dataset
        outs.append(lm.generate_until(task.instances))
    assert len(outs[0]) == 6 and outs[0] == outs[1]
    assert lm.generate_until([]) == []


@pytest.mark.parametrize("nxt", [False, True])
def test_llava_real_checkpoint_loading_path(gpu, tmp_path, nxt):
    """An on-disk llava-hf style checkpoint (legacy 4.47 names, fast tokenizer without a chat template -> Vicuna
    fallback) through the real loader; generation must equal the engine fed directly with the same weights."""
    import torch

    from lmms_owc_amd.engine.llava import LlavaEngine, LlavaWeights
    from lmms_owc_amd.models import get_model, imageproc
    from lmms_owc_amd.models._llava_hf import dims_from_hf_config
    from lmms_owc_amd.tasks import load_task
    from tests import ckpt_util

    d = tmp_path / "llava-tiny"
    info = ckpt_util.write_llava_checkpoint(d, legacy_names=True, next_=nxt)
    lm = get_model("custom-model", model_type="llava", model_name_or_path=str(d), batch_size=2)
    assert lm.eot_token_id == 2
    task = load_task("synthetic:3:70x120:2")
    task.build_all_requests(limit=None, rank=0, world_size=1)
    lm.task_dict[task.task_name] = task.dataset
    out = lm.generate_until(task.instances)
    assert len(out) == 3 and all(isinstance(x, str) for x in out)
    dims = dims_from_hf_config(json.loads((d / "config.json").read_text()))
    eng = LlavaEngine(LlavaWeights.from_state_dict(dims, info["weights"], gpu))
    views, size = lm._views(task.docs[0]["visual"])
    feats = eng.encode_views(eng.patchify(torch.from_numpy(views).to(gpu), imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD))
    rows = eng.feature_rows([views.shape[0]], [size])
    n_tok = len(rows[0])
    assert (n_tok == 16) == (not nxt)
    ctx = task.instances[0].args[0]
    ids = lm._prompt_ids(f"<image>\n{ctx}", [n_tok])
    assert ids[0] == 1 and (ids == info["image_token"]).sum() == n_tok   # BOS + expanded placeholder run
    toks = eng.generate_from_features([ids], feats, [rows[0]], int(task.instances[0].args[1].get("max_new_tokens", 8)),
                                      eos_token_id=2, pad_token_id=2).cpu().numpy()[0]
    stop = np.flatnonzero(toks == 2)
    direct = lm.tokenizer.batch_decode([toks[: stop[0]] if len(stop) else toks], skip_special_tokens=True)[0]
    assert lm.generate_until([task.instances[0]])[0] == direct

    # LLaVA.loglikelihood (reference _llava_hf.py:169-258) on the same checkpoint: (loss, greedy flag) per request, against the numpy
    # oracle fed with the ids the reference's recipe yields (always-prepended <image>, Vicuna fallback template, labels masked up
    # to the UN-expanded prompt length); a str target and a callable target
    from lmms_owc_amd.models._llava_hf import vicuna_prompt
    from lmms_owc_amd.tasks import TaskInstance
    from oracle import llava_np

    name, split = task.task_name, task.instances[0].args[5]
    reqs = [TaskInstance("loglikelihood", (ctx, " a sea lion", task.instances[0].args[2], 0, name, split), 0),
            TaskInstance("loglikelihood", (ctx, lambda doc: " " + str(doc["target"]), task.instances[1].args[2], 1, name, split), 1)]
    got = lm.loglikelihood(reqs)
    assert len(got) == 2 and all(isinstance(l, float) and isinstance(e, bool) for l, e in got)
    cont = [" a sea lion", " " + str(task.docs[1]["target"])]
    for i, (loss, eq) in enumerate(got):
        msgs = [{"role": "user", "content": f"<image>\n{ctx}"}, {"role": "assistant", "content": cont[i]}]
        n_ctx = len(lm.tokenizer.encode(vicuna_prompt(msgs[:1], "</s>", True), add_special_tokens=True))
        v_i, size_i = lm._views(task.docs[i]["visual"])
        rows_i = eng.feature_rows([v_i.shape[0]], [size_i])[0]
        ids_i = lm._expand(lm.tokenizer.encode(vicuna_prompt(msgs, "</s>", False), add_special_tokens=True), [len(rows_i)])
        assert ids_i[0] == 1 and 1 <= n_ctx < len(ids_i) and (ids_i == info["image_token"]).sum() == len(rows_i)
        pix = (v_i.astype(np.float32) / 255.0 - np.array(imageproc.OPENAI_CLIP_MEAN, np.float32)[None, :, None, None]) \
            / np.array(imageproc.OPENAI_CLIP_STD, np.float32)[None, :, None, None]
        kw = {"image_sizes": [size_i], "views_per_image": [v_i.shape[0]]} if nxt else {}
        o_loss, o_eq = llava_np.loglikelihood(info["weights"], info["cfg"], ids_i, pix, n_ctx, bf16=True, **kw)
        print(f"[loglik plugin] request {i}: HIP loss {loss:.5f} oracle {o_loss:.5f} flags {eq} {o_eq}")
        assert abs(loss - o_loss) <= 0.01 * o_loss and eq == o_eq, (i, loss, o_loss, eq, o_eq)


class IdTokenizer:
    """A text is a string of space-separated token ids (what tools/gen_golden.py fed the reference's scorer)."""

    def __call__(self, text, padding=True, truncation=True, return_tensors="np"):
        rows = [[int(t) for t in s.split()] for s in text]
        L = max(len(r) for r in rows)
        return {"input_ids": np.array([r + [0] * (L - len(r)) for r in rows], dtype=np.int64),
                "attention_mask": np.array([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=np.int64)}


def test_eval_ranking_cli_matches_reference(gpu, tmp_path, capsys):
    """eval_ranking.py end to end on the HIP scorer: every game outcome (win / draw / loss at the 0.05 threshold) and both
    printed Elo tables equal what the reference's own script produced on CPU fp32 for the same runs and BERT weights."""
    import eval_ranking
    from lmms_owc_amd.engine.scorer import BertWeights, SentenceScorer
    from lmms_owc_amd.pipelines import text
    from tests import recipes

    gold = json.loads((__import__("pathlib").Path(__file__).parent / "golden" / "ranking.json").read_text())["cases"]
    c = recipes.bert_cfg("tiny")
    text.set_sentence_bert(SentenceScorer(BertWeights(c, recipes.bert_weights(c, 1234), gpu)), IdTokenizer())
    recipes.ranking_runs(tmp_path)
    res = eval_ranking.main(eval_ranking.build_parser().parse_args(["-i", str(tmp_path), "-c", "semantic_similarity", "-b", "10", "-n", "200",
                                                                    "--log-level", "WARNING"]))
    assert res["toytask"]["scores"] == gold["default"]["scores"]
    assert capsys.readouterr().out == gold["default"]["stdout"]


def test_multi_round_generation(gpu, scorer):
    """generate_until_multi_round (the `*_llamav_o1` tasks): a tuple of per-round answers per document, independent of the
    batch size, every round generated from the running conversation (checked against a hand-built conversation fed to the
    engine directly), and the evaluate loop scores the last round."""
    import torch

    from lmms_owc_amd import ops
    from lmms_owc_amd.engine.evaluate import simple_evaluate
    from lmms_owc_amd.models import get_model, imageproc
    from lmms_owc_amd.pipelines import text
    from lmms_owc_amd.tasks import load_task

    text.set_sentence_bert(scorer, HashTokenizer())   # (an earlier test installs its own tokenizer)
    task = load_task("synthetic-mr:5:56x84:3")
    task.build_all_requests(limit=None, rank=0, world_size=1)
    outs = []
    for bs in (1, 3):
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs)
        lm.task_dict[task.task_name] = task.dataset
        outs.append(lm.generate_until_multi_round(task.instances))
    assert outs[0] == outs[1] and len(outs[0]) == 5 and all(isinstance(o, tuple) and len(o) == 3 for o in outs[0])
    # round 2 of document 1, rebuilt by hand: system + (user: image + q0) + (assistant: a0) + (user: q1) + (assistant: a1) + (user: q2)
    tok, eng = lm.tokenizer, lm.model
    img = imageproc.prepare_image(task.docs[1]["visual"], lm._min_pixels, lm._max_pixels)
    grid = (1, img.shape[1] // 14, img.shape[2] // 14)
    emb = eng.encode_images(ops.patchify_u8(torch.from_numpy(img[None]).to(gpu), imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD), [grid])
    a0, a1, a2 = outs[0][1]
    ids = tok.chat_ids_turns([("user", task.prompts[0], [grid[1] * grid[2] // 4]), ("assistant", a0, []), ("user", task.prompts[1], []),
                              ("assistant", a1, []), ("user", task.prompts[2], [])])
    toks = eng.generate([np.asarray(ids, np.int32)], emb, [[grid]], 6, eos_token_id=tok.eos_token_id, pad_token_id=0).cpu().numpy()[0]
    stop = np.flatnonzero(toks == tok.eos_token_id)
    assert tok.decode(toks[: stop[0]] if len(stop) else toks) == a2
    res = simple_evaluate(model="custom-model", model_args="model_type=qwen2-vl,model_name_or_path=synthetic:tiny", tasks=["synthetic-mr:4:56x56:2"],
                          batch_size=2, limit=4)
    smp = res["samples"]["synthetic"]
    assert len(smp) == 4 and len(smp[0]["filtered_resps"][0]) == 3 and "semantic_similarity,none" in res["results"]["synthetic"]


def test_multi_round_new_image_in_a_later_round_and_per_document_termination(gpu):
    """Round 4 (found by running the reference's wrapper, tests/test_wrapper_protocol.py): a later round may bring a NEW image - it is
    encoded when it appears and its rows are scattered behind the round-0 image's - and every document of a batch ends on its OWN
    terminal signal.  Real engine: document 1's second round (two images in the conversation) equals the hand-built conversation fed
    to the engine directly; the answers do not depend on the batch size; documents have 1 / 2 / 3 rounds."""
    import torch
    from PIL import Image

    from lmms_owc_amd import ops
    from lmms_owc_amd.models import get_model, imageproc
    from lmms_owc_amd.tasks import TaskInstance

    r = np.random.default_rng(5)
    docs = [{"i": i, "img": Image.fromarray(r.integers(0, 256, (56 + 28 * i, 84, 3), dtype=np.uint8), "RGB"),
             "extra": Image.fromarray(r.integers(0, 256, (84, 56, 3), dtype=np.uint8), "RGB")} for i in range(3)]
    qs = ["What is in the photo?", "And in this second one?", "So what do they share?"]

    def d2v(doc):
        return [doc["img"]]

    def d2t(doc, round_idx=0, previous_round_results=None, last_round_info=None):
        prev = list(previous_round_results or [])
        if round_idx > doc["i"]:                       # document i runs i + 1 rounds
            return None, None, True, prev, last_round_info
        return ([doc["extra"]] if round_idx == 1 else None), qs[round_idx], False, prev, last_round_info

    outs = []
    for bs in (1, 3):
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs)
        lm.task_dict["mr"] = {"test": docs}
        reqs = [TaskInstance(request_type="generate_until_multi_round", idx=0, metadata={"task": "mr", "doc_id": d["i"], "repeats": 1},
                             arguments=("<image>" + qs[0], {"max_new_tokens": 6, "do_sample": False}, d2v, d2t, d["i"], "mr", "test")) for d in docs]
        outs.append(lm.generate_until_multi_round(reqs))
    assert outs[0] == outs[1] and [len(t) for t in outs[0]] == [1, 2, 3]
    # document 2, round 1, by hand: system + (user: image A + q0) + (assistant: a0) + (user: image B + q1)
    tok, eng = lm.tokenizer, lm.model
    arrs = [imageproc.prepare_image(v, lm._min_pixels, lm._max_pixels) for v in (docs[2]["img"], docs[2]["extra"])]
    grids = [(1, a.shape[1] // 14, a.shape[2] // 14) for a in arrs]
    emb = torch.cat([eng.encode_images(ops.patchify_u8(torch.from_numpy(a[None]).to(gpu), imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD), [g])
                     for a, g in zip(arrs, grids)])
    a0, a1, _ = outs[0][2]
    n = [g[1] * g[2] // 4 for g in grids]
    ids = tok.chat_ids_turns([("user", qs[0], [n[0]]), ("assistant", a0, []), ("user", qs[1], [n[1]])])
    toks = eng.generate([np.asarray(ids, np.int32)], emb, [grids], 6, eos_token_id=tok.eos_token_id, pad_token_id=0).cpu().numpy()[0]
    stop = np.flatnonzero(toks == tok.eos_token_id)
    assert tok.decode(toks[: stop[0]] if len(stop) else toks) == a1


@pytest.mark.parametrize("name", ["tiny", "tiny-next"])
def test_llava_multi_round_generation(gpu, scorer, name):
    """LLaVA.generate_until_multi_round (/root/reference/src/models/_llava_hf.py:394-584): NO history - every round is an
    independent single-turn prompt of that round's (visuals, context); a tuple of per-round answers per document, independent
    of the batch size; round 0 equals `generate_until` on the same request; a later round whose `doc_to_text` returns no visual
    equals `generate_until` on a text-only request with that round's question; the evaluate loop scores the last round."""
    from lmms_owc_amd.engine.evaluate import simple_evaluate
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.pipelines import text
    from lmms_owc_amd.tasks import ClassificationTask, load_task

    text.set_sentence_bert(scorer, HashTokenizer())
    task = load_task("synthetic-mr:5:70x120:3")
    outs = []
    for bs in (1, 3):
        lm = get_model("custom-model", model_type="llava", model_name_or_path=f"synthetic:{name}", batch_size=bs, engine_batch=0)
        lm.task_dict[task.task_name] = task.dataset
        task.build_all_requests(limit=None, rank=0, world_size=1)
        outs.append(lm.generate_until_multi_round(task.instances))
    assert outs[0] == outs[1] and len(outs[0]) == 5 and all(isinstance(o, tuple) and len(o) == 3 for o in outs[0])
    # the same rounds as single-turn requests: round 0 with the image, rounds 1 and 2 text-only
    for rnd, with_image in ((0, True), (1, False), (2, False)):
        single = ClassificationTask("synthetic", task.docs, generation_kwargs={"max_new_tokens": 6, "do_sample": False})
        single.prompt = task.prompts[rnd]
        if not with_image:
            single.doc_to_visual = lambda doc: []
        lm.task_dict["synthetic"] = single.dataset
        single.build_all_requests(limit=None, rank=0, world_size=1)
        assert lm.generate_until(single.instances) == [o[rnd] for o in outs[0]], rnd
    res = simple_evaluate(model="custom-model", model_args=f"model_type=llava,model_name_or_path=synthetic:{name}", tasks=["synthetic-mr:4:56x56:2"],
                          batch_size=2, limit=4)
    smp = res["samples"]["synthetic"]
    assert len(smp) == 4 and len(smp[0]["filtered_resps"][0]) == 3 and "semantic_similarity,none" in res["results"]["synthetic"]


@pytest.mark.parametrize("model_type,name", [("qwen2-vl", "tiny"), ("llava", "tiny-next")])
def test_fp8_decoder_through_the_plugin(gpu, model_type, name):
    """`--model_args decoder_dtype=fp8`: the plug-ins build the e4m3fn decoder; answers are batch-invariant and their FIRST generated
    token (same context on both sides) agrees with the bf16 decoder's for most documents (random-weight miniatures: logits are
    near-flat, the model-level bar of tests/test_fp8_model_gpu.py is 75 % top-1 agreement over 32 prompts; here 12 documents)."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    task = load_task("synthetic:12:70x120:3")
    outs = {}
    for key, kw in {"fp8_1": dict(decoder_dtype="fp8", batch_size=1), "fp8_4": dict(decoder_dtype="fp8", batch_size=4), "bf16": dict(batch_size=4)}.items():
        lm = get_model("custom-model", model_type=model_type, model_name_or_path=f"synthetic:{name}", **kw)
        lm.task_dict[task.task_name] = task.dataset
        task.build_all_requests(limit=None, rank=0, world_size=1)
        outs[key] = [r[:1].tolist() for r in lm._generate_rows(task.instances)] if hasattr(lm, "_generate_rows") else None
        task.build_all_requests(limit=None, rank=0, world_size=1)
        outs[key + "_text"] = lm.generate_until(task.instances)
    assert outs["fp8_1_text"] == outs["fp8_4_text"] and len(outs["fp8_4_text"]) == 12
    if outs["bf16"] is not None:
        agree = np.mean([a == b for a, b in zip(outs["fp8_4"], outs["bf16"])])
        print(f"[fp8-plugin] {model_type}: first-token agreement with the bf16 decoder {agree:.2f} over {len(outs['bf16'])} documents")
        assert agree >= 0.5, (agree, outs["fp8_4"], outs["bf16"])
    assert lm.model.w.llm.weight_dtype == 0
    with pytest.raises(ValueError):
        get_model("custom-model", model_type=model_type, model_name_or_path=f"synthetic:{name}", decoder_dtype="int4")


def test_token_records_equal_strings_and_pipeline_chunks(gpu):
    """`generate_until_tokens` + `decode_tokens` (the fixed-width records of the end-of-task RCCL gather) give exactly the strings
    `generate_until` returns; the two-stage pipeline (preparation thread working in small units + pinned staging reuse + adaptive
    assembly of engine passes + deferred token read-back) returns the same answers for every batch size, in request order."""
    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import load_task

    task = load_task("synthetic:300:56x84:3")      # 300 requests: batch 7 = 43 passes of one unit; batch 256 = units of 64, passes as they get ready
    outs = {}
    for bs in (7, 256):
        lm = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=bs, engine_batch=0)
        lm.task_dict[task.task_name] = task.dataset
        task.build_all_requests(limit=None, rank=0, world_size=1)
        outs[bs] = lm.generate_until(task.instances)
        sizes = lm.last_timing["pass_sizes"]
        assert sum(sizes) == 300 and max(sizes) <= bs and lm.last_timing["chunks"] == len(sizes)
        assert len(sizes) == 43 if bs == 7 else 2 <= len(sizes) <= 5
    assert outs[7] == outs[256] and len(outs[7]) == 300
    # engine_batch="auto" (the default): batch_size is only a lower bound, the whole task fits one engine pass - same answers
    auto = get_model("custom-model", model_type="qwen2-vl", model_name_or_path="synthetic:tiny", batch_size=1)
    auto.task_dict[task.task_name] = task.dataset
    task.build_all_requests(limit=None, rank=0, world_size=1)
    assert auto.engine_batch(8) >= 300 and auto.generate_until(task.instances) == outs[7] and sum(auto.last_timing["pass_sizes"]) == 300
    task.build_all_requests(limit=None, rank=0, world_size=1)
    mat, n = lm.generate_until_tokens(task.instances)
    assert mat.shape[0] == 300 and mat.dtype == np.int32 and n.shape == (300,)
    assert lm.decode_tokens([mat[i, : n[i]] for i in range(300)]) == outs[7]
    # a second plug-in around the SAME engine (no second weight set) answers identically
    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL

    twin = Qwen2VL.from_engine(lm._model, ByteTokenizer(), batch_size=64)
    twin.task_dict[task.task_name] = task.dataset
    task.build_all_requests(limit=None, rank=0, world_size=1)
    assert twin.generate_until(task.instances) == outs[7]
