"""oracle/multiround.py pinned on the REFERENCE's own run (tests/golden/wrapper_protocol.json, written by
tools/gen_golden_wrappers.py: /root/reference/src/models/_qwen2_vl.py:350-616 and _llava_hf.py:440-584 executed in the build
container on a stand-in checkpoint whose answer is a function of the rendered prompt).  The restated protocol, driven with the same
stand-in, must hand the model the same prompts with the same number of images, round by round, and return the same tuples - then
tests/test_host_logic.py's comparison of the product with this oracle is a comparison with the reference's behaviour."""
import json
from pathlib import Path

import pytest

from tests import recipes

GOLD = Path(__file__).parent / "golden" / "wrapper_protocol.json"


def _in_order(sub: list, full: list) -> bool:
    it = iter(full)
    return all(any(x == y for y in it) for x in sub)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_qwen2vl_multi_round_oracle_equals_the_reference_run(case):
    from oracle import multiround as MR

    gold = json.loads(GOLD.read_text())["qwen2vl"][case]
    docs, d2v, d2t = recipes.mr_docs_and_task()
    assert gold["contexts"] == [recipes.mr_context(d) for d in docs]
    calls_all, results = [], []
    for doc in docs:
        calls = []

        def generate_text(message):
            text = MR.render_qwen2vl_chat(message)
            n_img = sum(1 for turn in message if isinstance(turn["content"], list) for c in turn["content"] if c.get("type") == "image")
            calls.append((text, n_img))
            return recipes.mr_answer_of(text)

        # the reference's default stop string is the decoded EOT token (:420); the stand-in tokenizer decodes it to "<|im_end|>"
        results.append(list(MR.reference_multi_round(doc, recipes.mr_context(doc), d2v, d2t, gold["gen_kwargs"], generate_text, "<|im_end|>")))
        calls_all.append(calls)
    assert results == gold["results"]                       # per request, original order, every round's (cut) answer
    ref_calls = [(t["texts"][0], t["images"]) for t in gold["trace"]]
    assert sorted(ref_calls) == sorted(c for calls in calls_all for c in calls)          # the same generate calls ...
    assert all(_in_order(calls, ref_calls) for calls in calls_all)                        # ... in round order per request
    # what the reference passes to HF generate (its defaults: :559-566; greedy, one beam)
    want_new = {0: 48, 1: 128, 2: 16}[case]
    for t in gold["trace"]:
        kw = t["generate_kwargs"]
        assert kw["max_new_tokens"] == want_new and kw["do_sample"] is False and kw["num_beams"] == 1 and kw["temperature"] == 0
    if case != 2:   # an `until` string cuts the answer and what the next round is told (the trailing space stays: no strip)
        assert any(a.endswith(" ") for r in gold["results"] for a in r) and not any("STOP" in a for r in gold["results"] for a in r)


@pytest.mark.parametrize("case", [0, 1])
def test_llava_multi_round_oracle_equals_the_reference_run(case):
    from lmms_owc_amd.models._llava_hf import vicuna_prompt   # (pinned on the reference's template by test_host_logic.py)
    from oracle import multiround as MR

    gold = json.loads(GOLD.read_text())["llava"][case]
    docs, d2v, d2t = recipes.mr_docs_and_task()
    calls_all, results = [], []
    for doc in docs:
        calls = []

        def generate_text(ctx, visuals):
            text = vicuna_prompt([{"role": "user", "content": ctx}])
            calls.append((text, len(visuals)))
            return recipes.mr_answer_of(text)

        results.append(list(MR.reference_multi_round_llava(doc, recipes.mr_context(doc), d2v, d2t, gold["gen_kwargs"], generate_text)))
        calls_all.append(calls)
    assert results == gold["results"]
    ref_calls = [(t["texts"][0], t["images"]) for t in gold["trace"]]
    assert sorted(ref_calls) == sorted(c for calls in calls_all for c in calls)
    assert all(_in_order(calls, ref_calls) for calls in calls_all)
    want_new = {0: 48, 1: 1024}[case]
    assert all(t["generate_kwargs"]["max_new_tokens"] == want_new and t["generate_kwargs"]["do_sample"] is False for t in gold["trace"])
    # LLaVA's wrapper pops `until` and never applies it (:461-470): the stop string and its tail stay in the answers
    assert any("STOP trailing" in a for r in gold["results"] for a in r)


def _text_of_ids(tok, ids) -> str:
    """Prompt ids of the byte tokenizer -> the rendered chat string (a run of image placeholders = one `<|image_pad|>`)."""
    names = {tok.im_start: "<|im_start|>", tok.im_end: "<|im_end|>", tok.vision_start: "<|vision_start|>", tok.vision_end: "<|vision_end|>"}
    out, buf, prev = [], [], None
    for t in (int(x) for x in ids):
        if 3 <= t < 259:
            buf.append(t - 3)
        else:
            if buf:
                out.append(bytes(buf).decode())
                buf = []
            if t == tok.image_pad:
                if prev != tok.image_pad:
                    out.append("<|image_pad|>")
            else:
                out.append(names[t])
        prev = t
    if buf:
        out.append(bytes(buf).decode())
    return "".join(out)


@pytest.mark.parametrize("case,data_urls", [(0, False), (1, False), (2, False), (0, True)])
def test_product_multi_round_equals_the_reference_run(case, data_urls):
    """The PRODUCT's host path against the reference's own run, no oracle in between: `Qwen2VL.generate_until_multi_round` (batched,
    three requests per engine pass) around a stand-in engine that decodes the prompt ids it is handed back into text and answers
    with the golden's stand-in function of that text.  Every answer is a hash of the rendered conversation, so equal result tuples
    mean equal prompts in every round: system turn, image placeholders in the turns that carry images, `until`-cut assistant turns
    (trailing space and all), round order, early terminal signal, original request order."""
    import numpy as np
    import torch

    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL
    from lmms_owc_amd.tasks import TaskInstance

    gold = json.loads(GOLD.read_text())["qwen2vl"][case]
    tok = ByteTokenizer()
    seen = []

    class Dims:
        image_token_id, decoder_dtype = tok.image_pad, "bf16"

    class FakeEngine:
        d, device = Dims(), torch.device("cpu")

        def encode_images(self, pix, grids):
            return None

        def generate(self, prompts, emb, grids, max_new, eos_token_id=-1, pad_token_id=0, **_):
            out = np.full((len(prompts), max_new), pad_token_id, np.int32)
            for i, p in enumerate(prompts):
                text = _text_of_ids(tok, p)
                seen.append((text, max_new))
                t = (tok.encode(recipes.mr_answer_of(text)) + [eos_token_id])[:max_new]
                out[i, : len(t)] = t
            return torch.from_numpy(out)

    class HostOnly(Qwen2VL):
        def _pixel_values(self, images):
            return None

    docs, d2v, d2t = recipes.mr_docs_and_task()
    if data_urls:   # a task that hands the conversation back in the REFERENCE's message format: images as base64 JPEG data URLs (:485-500)
        import base64
        from io import BytesIO

        base = d2t

        def d2t(doc, round_idx=0, previous_round_results=None, last_round_info=None):   # noqa: F811
            vis, text, terminal, prev, info = base(doc, round_idx=round_idx, previous_round_results=previous_round_results,
                                                   last_round_info=last_round_info)
            for msg in (info or {}).get("messages", []):
                for turn in msg:
                    for c in turn["content"] if isinstance(turn["content"], list) else []:
                        if c.get("type") == "image" and not isinstance(c["image"], str):
                            buf = BytesIO()
                            c["image"].convert("RGB").save(buf, format="JPEG")
                            c["image"] = "data:image/jpeg;base64," + base64.b64encode(buf.getvalue()).decode()
            return vis, text, terminal, prev, info

    lm = HostOnly.from_engine(FakeEngine(), tok, batch_size=3)
    lm.task_dict["mr"] = {"test": docs}
    reqs = [TaskInstance(request_type="generate_until_multi_round", idx=0, metadata={"task": "mr", "doc_id": d["id"], "repeats": 1},
                         arguments=(recipes.mr_context(d), dict(gold["gen_kwargs"]), d2v, d2t, d["id"], "mr", "test")) for d in docs]
    try:
        got = lm.generate_until_multi_round(reqs)
    finally:
        lm._pool.shutdown()
        lm._prep_thread.shutdown()
    ref_texts = sorted(t["texts"][0] for t in gold["trace"])
    if case != 2:     # (case 2: 16 new tokens cut the stand-in's answers short of what the reference's stand-in returned whole)
        assert [list(t) for t in got] == gold["results"]
        assert sorted(t for t, _ in seen) == ref_texts                       # the same rendered prompts, call for call
    assert {m for _, m in seen} == {gold["trace"][0]["generate_kwargs"]["max_new_tokens"]}     # 48 / the default 128 / 16


@pytest.mark.parametrize("case", [0, 1])
def test_product_llava_multi_round_equals_the_reference_run(case):
    """`LLaVA.generate_until_multi_round` (batched, three requests per pass) against the reference's own run: the stand-in engine
    collapses every image's placeholder run back to one `<image>`, decodes the prompt and answers with the golden's function of it -
    equal tuples mean equal single-turn prompts in every round (image tokens prepended when the context has none, none for a round
    without visuals, `until` never applied, every document on its own terminal signal)."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from lmms_owc_amd.engine.llava import DIMS, LlavaDims, LlavaEngine
    from lmms_owc_amd.models._base import CacheHook
    from lmms_owc_amd.models._llava_hf import LLaVA, LlavaByteTokenizer
    from lmms_owc_amd.tasks import TaskInstance

    gold = json.loads(GOLD.read_text())["llava"][case]
    tok = LlavaByteTokenizer()
    dims = LlavaDims(**{**DIMS["tiny"].__dict__, "image_token_id": tok.image_token_id})
    seen = []

    class FakeEngine:
        d, device = dims, torch.device("cpu")
        feature_rows = LlavaEngine.feature_rows

        def generate_from_features(self, prompts, feats, rows_per_prompt, max_new, eos_token_id=-1, pad_token_id=0, **_):
            out = np.full((len(prompts), max_new), pad_token_id, np.int32)
            for i, p in enumerate(prompts):
                p = [int(t) for t in p]
                ids = [t for k, t in enumerate(p) if not (t == dims.image_token_id and k and p[k - 1] == dims.image_token_id)]
                # consecutive images are separated by a space in the prompt (`<image> <image>`), so runs never merge
                text, buf = "", []
                for t in ids + [None]:
                    if t is not None and 3 <= t < 259:
                        buf.append(t - 3)
                        continue
                    text += bytes(buf).decode()
                    buf = []
                    if t == dims.image_token_id:
                        text += "<image>"
                seen.append((text, max_new))
                t = (tok.encode(recipes.mr_answer_of(text)) + [eos_token_id])[:max_new]
                out[i, : len(t)] = t
            return torch.from_numpy(out)

    class HostOnly(LLaVA):
        def _encode_visuals(self, flat, feature_cache=None):
            prepared = [self._views(v) for v in flat]
            return None, (self._model.feature_rows([p[0].shape[0] for p in prepared], [p[1] for p in prepared]) if prepared else [])

    lm = HostOnly.__new__(HostOnly)
    lm._engine_batch_arg, lm._decoder_dtype, lm._chat_template = 0, "bf16", None
    lm._device, lm._rank, lm._world_size, lm.batch_size_per_gpu = torch.device("cpu"), 0, 1, 3
    lm.cache_hook, lm.task_dict = CacheHook(None), {}
    lm._tokenizer = lm._processor = tok
    lm._dims, lm._model, lm._pool = dims, FakeEngine(), ThreadPoolExecutor(max_workers=2)
    docs, d2v, d2t = recipes.mr_docs_and_task()
    lm.task_dict["mr"] = {"test": docs}
    reqs = [TaskInstance(request_type="generate_until_multi_round", idx=0, metadata={"task": "mr", "doc_id": d["id"], "repeats": 1},
                         arguments=(recipes.mr_context(d), dict(gold["gen_kwargs"]), d2v, d2t, d["id"], "mr", "test")) for d in docs]
    try:
        got = lm.generate_until_multi_round(reqs)
    finally:
        lm._pool.shutdown()
    assert [list(t) for t in got] == gold["results"]
    assert sorted(t for t, _ in seen) == sorted(t["texts"][0] for t in gold["trace"])
    assert {m for _, m in seen} == {gold["trace"][0]["generate_kwargs"]["max_new_tokens"]}     # 48 / the default 1024


def _gk(gold: dict, doc: dict) -> dict:
    """The request's own gen_kwargs (case 3 of the single-round goldens mixes two settings by document)."""
    g = gold["gen_kwargs"]
    return dict(g[doc["id"] % len(g)] if isinstance(g, list) else g)


class _DoneEvent:
    def query(self):
        return True

    def synchronize(self):
        pass


def _llava_text(tok, image_token_id, p) -> str:
    """Prompt ids of the LLaVA byte tokenizer -> the rendered prompt (every image's placeholder run = one `<image>`)."""
    p = [int(t) for t in p]
    ids = [t for k, t in enumerate(p) if not (t == image_token_id and k and p[k - 1] == image_token_id)]
    text, buf = "", []
    for t in ids + [None]:
        if t is not None and 3 <= t < 259:
            buf.append(t - 3)
            continue
        text += bytes(buf).decode()
        buf = []
        if t == image_token_id:
            text += "<image>"
    return text


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_product_generate_until_equals_the_reference_run(case):
    """The HOT PATH's host loop against the reference's own run (`Qwen2VL.generate_until`, /root/reference/src/models/_qwen2_vl.py:
    143-348, executed by tools/gen_golden_wrappers.py on the stand-in checkpoint): six requests - with / without an `<image>`
    marker, two images, no image, trailing blanks - through the product's pass pipeline (units of three, merged passes) around a
    stand-in engine.  Equal strings in request order mean equal rendered prompts (the answer is a hash of the prompt), the same
    default `max_new_tokens`, `until` popped and NOT applied, nothing stripped."""
    import numpy as np
    import torch

    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL
    from lmms_owc_amd.tasks import TaskInstance

    gold = json.loads(GOLD.read_text())["qwen2vl_single"][case]
    tok = ByteTokenizer()
    seen = []

    class Dims:
        image_token_id, decoder_dtype = tok.image_pad, "bf16"

    class FakeEngine:
        d, device = Dims(), torch.device("cpu")

        def generate(self, prompts, emb, grids, max_new, eos_token_id=-1, pad_token_id=0, **_):
            out = np.full((len(prompts), max_new), pad_token_id, np.int32)
            for i, p in enumerate(prompts):
                text = _text_of_ids(tok, p)
                seen.append((text, max_new, sum(len(g) for g in [grids[i]])))
                t = (tok.encode(recipes.mr_answer_of(text)) + [eos_token_id])[:max_new]
                out[i, : len(t)] = t
            return torch.from_numpy(out)

    class HostOnly(Qwen2VL):
        def _pinned_take(self, shape):
            return torch.empty(shape, dtype=torch.uint8)

        def _pinned_give(self, groups):
            pass

        def _launch_chunk(self, prep, eos_token_id, pad, carry=None):
            sampled.append(prep["sampling"])
            return self._model.generate(prep["prompts"], None, prep["grids"], prep["max_new"], eos_token_id=eos_token_id, pad_token_id=pad), _DoneEvent()

    docs, d2v = recipes.su_docs_and_task()
    sampled = []
    lm = HostOnly.from_engine(FakeEngine(), tok, batch_size=3)
    lm._no_carry = True
    lm.task_dict["su"] = {"test": docs}
    reqs = [TaskInstance(request_type="generate_until", idx=0, metadata={"task": "su", "doc_id": d["id"], "repeats": 1},
                         arguments=(recipes.su_context(d), _gk(gold, d), d2v, d["id"], "su", "test")) for d in docs]
    try:
        got = lm.generate_until(reqs)
    finally:
        lm._pool.shutdown()
        lm._prep_thread.shutdown()
    assert got == gold["results"]
    assert sorted((t, n) for t, _, n in seen) == sorted((t["texts"][0], t["images"]) for t in gold["trace"])     # prompts AND image counts
    # per prompt the generation length the reference asked HF for: 64 / the default 128 / (case 3) 64 or 96 by the request's own
    # gen_kwargs - requests with different settings never share a pass (the Collator groups by them)
    assert sorted((t, m) for t, m, _ in seen) == sorted((t["texts"][0], t["generate_kwargs"]["max_new_tokens"]) for t in gold["trace"])
    assert all("until" not in r.args[1] for r in reqs)             # popped from the request's own dict, as the reference does (:211-219)
    # `do_sample = temperature > 0` (:308-329): what the reference hands HF generate is what the on-device sampler is given
    kw = gold["trace"][0]["generate_kwargs"]
    assert kw["num_beams"] == 1 and kw["do_sample"] is (case == 2)
    if case == 2:
        assert all(s_ is not None and abs(s_["temperature"] - kw["temperature"]) < 1e-6 and abs(s_["top_p"] - kw["top_p"]) < 1e-6 for s_ in sampled)
    else:
        assert all(s_ is None for s_ in sampled) and kw["temperature"] == 0


@pytest.mark.parametrize("case", [0, 1, 2, 3])
def test_product_llava_generate_until_equals_the_reference_run(case):
    """`LLaVA.generate_until` (/root/reference/src/models/_llava_hf.py:260-392) the same way."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from lmms_owc_amd.engine.llava import DIMS, LlavaDims, LlavaEngine
    from lmms_owc_amd.models._base import CacheHook
    from lmms_owc_amd.models._llava_hf import LLaVA, LlavaByteTokenizer
    from lmms_owc_amd.tasks import TaskInstance

    gold = json.loads(GOLD.read_text())["llava_single"][case]
    tok = LlavaByteTokenizer()
    dims = LlavaDims(**{**DIMS["tiny"].__dict__, "image_token_id": tok.image_token_id})
    seen = []

    class FakeEngine:
        d, device = dims, torch.device("cpu")
        feature_rows = LlavaEngine.feature_rows

    class HostOnly(LLaVA):
        def _pinned_take(self, shape):
            return torch.empty(shape, dtype=torch.uint8)

        def _pinned_give(self, groups):
            pass

        def _launch_chunk(self, prep, eos_token_id, pad, carry=None):
            sampled.append(prep["sampling"])
            out = np.full((prep["n"], prep["max_new"]), eos_token_id, np.int32)
            for i, p in enumerate(prep["prompts"]):
                text = _llava_text(tok, dims.image_token_id, p)
                seen.append((text, prep["max_new"], prep["images_per_prompt"][i]))
                t = (tok.encode(recipes.mr_answer_of(text)) + [eos_token_id])[: prep["max_new"]]
                out[i, : len(t)] = t
            return torch.from_numpy(out), _DoneEvent()

    lm = HostOnly.__new__(HostOnly)
    lm._engine_batch_arg, lm._decoder_dtype, lm._chat_template = 0, "bf16", None
    lm._device, lm._rank, lm._world_size, lm.batch_size_per_gpu = torch.device("cpu"), 0, 1, 3
    lm.cache_hook, lm.task_dict, lm._no_carry = CacheHook(None), {}, True
    lm._tokenizer = lm._processor = tok
    lm._dims, lm._model = dims, FakeEngine()
    lm._pool, lm._prep_thread = ThreadPoolExecutor(max_workers=2), ThreadPoolExecutor(max_workers=1)
    sampled = []
    docs, d2v = recipes.su_docs_and_task()
    lm.task_dict["su"] = {"test": docs}
    reqs = [TaskInstance(request_type="generate_until", idx=0, metadata={"task": "su", "doc_id": d["id"], "repeats": 1},
                         arguments=(recipes.su_context(d), _gk(gold, d), d2v, d["id"], "su", "test")) for d in docs]
    try:
        got = lm.generate_until(reqs)
    finally:
        lm._pool.shutdown()
        lm._prep_thread.shutdown()
    assert got == gold["results"]
    assert sorted((t, n) for t, _, n in seen) == sorted((t["texts"][0], t["images"]) for t in gold["trace"])
    assert sorted((t, m) for t, m, _ in seen) == sorted((t["texts"][0], t["generate_kwargs"]["max_new_tokens"]) for t in gold["trace"])
    kw = gold["trace"][0]["generate_kwargs"]
    assert kw["num_beams"] == 1 and kw["do_sample"] is (case == 2)
    if case == 2:
        assert all(s_ is not None and abs(s_["temperature"] - kw["temperature"]) < 1e-6 and abs(s_["top_p"] - kw["top_p"]) < 1e-6 for s_ in sampled)
    else:
        assert all(s_ is None for s_ in sampled)


def test_product_llava_loglikelihood_equals_the_reference_run():
    """`LLaVA.loglikelihood` against the reference's own run (/root/reference/src/models/_llava_hf.py:169-258 on a stand-in decoder whose
    logits are `recipes.ll_logits(ids)`): six requests (one / two / no image, string and callable targets).  The product hands its
    engine's scoring call the same ids - image tokens always prepended, the Vicuna prompt and prompt + continuation renderings, every
    `<image>` expanded to the image's feature count - and the same `start` (the prompt's length tokenised WITHOUT expansion, the
    reference's label mask), and returns the same (mean shifted cross-entropy, unshifted greedy flag)."""
    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor

    from lmms_owc_amd.engine.llava import DIMS, LlavaDims, LlavaEngine
    from lmms_owc_amd.models._base import CacheHook
    from lmms_owc_amd.models._llava_hf import LLaVA, LlavaByteTokenizer
    from lmms_owc_amd.tasks import TaskInstance

    gold = json.loads(GOLD.read_text())["llava_loglikelihood"]
    tok = LlavaByteTokenizer()
    dims = LlavaDims(**{**DIMS["tiny"].__dict__, "image_token_id": tok.image_token_id})
    assert dims.grid ** 2 == gold["tokens_per_image"]
    calls = []

    class FakeEngine:
        d, device = dims, torch.device("cpu")
        feature_rows = LlavaEngine.feature_rows

        def score(self, ids, feats, grids, start, img_rows=None):
            ids = np.asarray(ids)
            assert int((ids == dims.image_token_id).sum()) == len(img_rows)
            calls.append({"sequence_length": len(ids), "masked_leading_positions": int(start)})
            z = recipes.ll_logits(ids)
            lsm = z - np.log(np.exp(z - z.max(1, keepdims=True)).sum(1, keepdims=True)) - z.max(1, keepdims=True)
            lp = np.array([lsm[i - 1, ids[i]] for i in range(start, len(ids))])
            return lp.astype(np.float32), z[start:].argmax(1).astype(np.int32)

    class HostOnly(LLaVA):
        def _encode_visuals(self, flat, feature_cache=None):
            prepared = [self._views(v) for v in flat]
            return None, (self._model.feature_rows([p[0].shape[0] for p in prepared], [p[1] for p in prepared]) if prepared else [])

    lm = HostOnly.__new__(HostOnly)
    lm._engine_batch_arg, lm._decoder_dtype, lm._chat_template = 0, "bf16", None
    lm._device, lm._rank, lm._world_size, lm.batch_size_per_gpu = torch.device("cpu"), 0, 1, 1
    lm.cache_hook, lm.task_dict = CacheHook(None), {}
    lm._tokenizer = lm._processor = tok
    lm._dims, lm._model, lm._pool = dims, FakeEngine(), ThreadPoolExecutor(max_workers=2)
    docs, d2v = recipes.su_docs_and_task()
    lm.task_dict["su"] = {"test": docs}
    reqs = [TaskInstance(request_type="loglikelihood", idx=0, metadata={"task": "su", "doc_id": d["id"], "repeats": 1},
                         arguments=(recipes.ll_context(d), recipes.ll_continuation(d) if d["id"] % 2 else (lambda doc: recipes.ll_continuation(doc)),
                                    d2v, d["id"], "su", "test")) for d in docs]
    try:
        got = lm.loglikelihood(reqs)
    finally:
        lm._pool.shutdown()
    assert [c["sequence_length"] for c in calls] == [t["sequence_length"] for t in gold["trace"]]
    assert [c["masked_leading_positions"] for c in calls] == [t["masked_leading_positions"] for t in gold["trace"]]
    assert [m for _, m in got] == [m for _, m in gold["results"]] and any(m for _, m in got) and not all(m for _, m in got)
    np.testing.assert_allclose([l for l, _ in got], [l for l, _ in gold["results"]], rtol=2e-6)


def test_prompt_ids_with_a_real_tokenizer_equal_hf_processor(tmp_path):
    """The checkpoint path of the prompt builder (`Qwen2VL._prompt_ids` / `_messages_ids` with an HF tokenizer: chat template through
    the tokenizer, every `<|image_pad|>` expanded to its image's token count) against HF's own `Qwen2VLProcessor.__call__` - what the
    reference runs at `_qwen2_vl.py:289-305` - on the tiny on-disk tokenizer of tests/ckpt_util.py: one, two and no images, a
    two-turn conversation.  (The processor object is assembled without its video processor: transformers 5.x wants torchvision
    for that class; the image / text path is HF's code.)"""
    import numpy as np
    import torch
    from PIL import Image
    from transformers import AutoTokenizer, Qwen2VLImageProcessor, Qwen2VLProcessor

    from lmms_owc_amd.models import imageproc
    from lmms_owc_amd.models._qwen2_vl import SYSTEM_PROMPT, Qwen2VL
    from tests import ckpt_util

    cfg = recipes.tiny_cfg()
    specials = {"<|endoftext|>": 490, "<|im_start|>": 491, "<|im_end|>": 492, "<|vision_start|>": 493, "<|vision_end|>": 494,
                "<|image_pad|>": cfg.image_token_id}
    ckpt_util.write_tokenizer(tmp_path, cfg.text.vocab_size, specials)
    tok = AutoTokenizer.from_pretrained(str(tmp_path))
    proc = object.__new__(Qwen2VLProcessor)
    proc.image_processor, proc.tokenizer = Qwen2VLImageProcessor(min_pixels=4 * 784, max_pixels=1024 * 784), tok
    proc.video_processor, proc.chat_template = None, tok.chat_template
    proc.image_token, proc.video_token = "<|image_pad|>", "<|video_pad|>"
    proc.image_token_id, proc.video_token_id = cfg.image_token_id, None

    class Dims:
        image_token_id, decoder_dtype = cfg.image_token_id, "bf16"

    class FakeEngine:
        d, device = Dims(), torch.device("cpu")

    lm = Qwen2VL.from_engine(FakeEngine(), tok)
    try:
        r = np.random.default_rng(2)
        imgs = [Image.fromarray(r.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for h, w in ((56, 84), (112, 56))]
        n_tok = []
        for im in imgs:
            a = imageproc.prepare_image(im, 4 * 784, 1024 * 784, jpeg=False)
            n_tok.append(a.shape[1] * a.shape[2] // (14 * 14 * 4))
        q = "what type of object is in this photo ?"

        def hf_ids(messages, images):
            text = tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
            out = proc(text=[text], images=images or None, padding=True, return_tensors="pt")
            return out["input_ids"][0].tolist()

        system = {"role": "system", "content": SYSTEM_PROMPT}
        # single round, as the reference builds the message (:223-282): one image / none
        for k in (1, 0):
            content = [{"type": "image", "image": imgs[0]}] * k + [{"type": "text", "text": q}]
            want = hf_ids([system, {"role": "user", "content": content}], imgs[:k])
            assert lm._prompt_ids(q, n_tok[:k]).tolist() == want
            assert int((np.asarray(want) == cfg.image_token_id).sum()) == sum(n_tok[:k])
        # a two-turn conversation with an image in each user turn (multi-round, :477-533)
        msgs = [system, {"role": "user", "content": [{"type": "image", "image": imgs[0]}, {"type": "text", "text": q}]},
                {"role": "assistant", "content": [{"type": "text", "text": "a sea lion"}]},
                {"role": "user", "content": [{"type": "image", "image": imgs[1]}, {"type": "text", "text": "is this a dog ?"}]}]
        assert lm._messages_ids(msgs, n_tok).tolist() == hf_ids(msgs, imgs)
    finally:
        lm._pool.shutdown()
        lm._prep_thread.shutdown()


@pytest.mark.parametrize("next_", [False, True])
def test_llava_prompt_ids_with_a_real_tokenizer_equal_hf_processor(tmp_path, next_):
    """`LLaVA._prompt_ids` on a checkpoint's own tokenizer against HF's `LlavaProcessor` / `LlavaNextProcessor` (what the reference calls
    at `_llava_hf.py:347`): BOS, tokenisation of the rendered prompt, and every `<image>` expanded to the image's feature count - for
    LLaVA-NeXT the anyres count (best resolution, un-padded tile grid + one newline per row + the base view) that HF's processor
    computes from the image size, for landscape / portrait / extreme / square images; two images in one prompt."""
    import json as _json

    import numpy as np
    import torch
    from concurrent.futures import ThreadPoolExecutor
    from PIL import Image
    from transformers import AutoTokenizer, CLIPImageProcessor, LlavaNextImageProcessor, LlavaNextProcessor, LlavaProcessor

    from lmms_owc_amd.engine.llava import LlavaEngine
    from lmms_owc_amd.models._base import CacheHook
    from lmms_owc_amd.models._llava_hf import LLaVA, dims_from_hf_config
    from tests import ckpt_util

    info = ckpt_util.write_llava_checkpoint(tmp_path, next_=next_)
    cfg = info["cfg"]
    tok = AutoTokenizer.from_pretrained(str(tmp_path))
    size = cfg.vision.image_size
    kw = dict(tokenizer=tok, patch_size=14, vision_feature_select_strategy="default", num_additional_image_tokens=1)
    if next_:
        ip = LlavaNextImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size},
                                     image_grid_pinpoints=[list(p) for p in cfg.image_grid_pinpoints])
        proc = LlavaNextProcessor(image_processor=ip, **kw)
    else:
        proc = LlavaProcessor(image_processor=CLIPImageProcessor(size={"shortest_edge": size}, crop_size={"height": size, "width": size}), **kw)
    dims = dims_from_hf_config(_json.loads((tmp_path / "config.json").read_text()))

    class FakeEngine:
        d, device = dims, torch.device("cpu")
        feature_rows = LlavaEngine.feature_rows

    lm = LLaVA.__new__(LLaVA)
    lm._engine_batch_arg, lm._decoder_dtype, lm._chat_template = 0, "bf16", None
    lm._device, lm._rank, lm._world_size, lm.batch_size_per_gpu = torch.device("cpu"), 0, 1, 1
    lm.cache_hook, lm.task_dict = CacheHook(None), {}
    lm._tokenizer = lm._processor = tok
    lm._dims, lm._model, lm._pool = dims, FakeEngine(), ThreadPoolExecutor(max_workers=1)
    r = np.random.default_rng(3)
    try:
        for sizes in ([(60, 90)], [(90, 60)], [(40, 120)], [(33, 33)], [(60, 90), (50, 50)]):
            imgs = [Image.fromarray(r.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for h, w in sizes]
            ctx = " ".join(["<image>"] * len(imgs)) + "\nwhat type of object is in this photo ?"
            counts = []
            for im in imgs:
                views, size_hw = lm._views(im)
                counts.append(len(lm._model.feature_rows([views.shape[0]], [size_hw])[0]))
            want = proc(images=imgs, text=lm._render(ctx), return_tensors="pt")["input_ids"][0].tolist()
            got = lm._prompt_ids(ctx, counts).tolist()
            assert got == want, (sizes, counts, sum(1 for t in want if t == dims.image_token_id))
    finally:
        lm._pool.shutdown()


def test_stragglers_are_only_handed_to_a_pass_of_their_own_generation_settings():
    """Straggler hand-over with MIXED gen_kwargs in one request list (round 4's ADVICE): a pass applies one generation length and
    one set of sampling switches to every row of its decode batch, so an unfinished sequence may only travel to a pass of ITS key.
    Four settings (greedy 64, greedy 96, sampled t = 0.7, sampled t = 0.3), five text-only requests each, passes of two: a
    stand-in `_launch_chunk` hands over its last own row whenever the pipeline allows it (`below` > 0) and checks that every
    carried-in sequence was exported under this pass's key; the last pass of every key must be told to drain (`below` == 0)."""
    import numpy as np
    import torch

    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL
    from lmms_owc_amd.tasks import TaskInstance

    tok = ByteTokenizer()

    class Dims:
        image_token_id, decoder_dtype = tok.image_pad, "bf16"

    class FakeEngine:
        d, device = Dims(), torch.device("cpu")

    log = []

    def answer(prompt) -> list:
        return tok.encode(recipes.mr_answer_of(_text_of_ids(tok, prompt)))

    class HostOnly(Qwen2VL):
        def _pinned_take(self, shape):
            return torch.empty(shape, dtype=torch.uint8)

        def _pinned_give(self, groups):
            pass

        def _launch_chunk(self, prep, eos_token_id, pad, carry=None):
            key = prep["key"] if "key" in prep else (prep["max_new"], None if prep["sampling"] is None else
                                                     (prep["sampling"]["temperature"], prep["sampling"]["top_p"], prep["sampling"]["top_k"]))
            out = np.full((prep["n"], prep["max_new"]), pad, np.int32)
            rows = []
            for i, p in enumerate(prep["prompts"]):
                t = (answer(p) + [eos_token_id])[: prep["max_new"]]
                out[i, : len(t)] = t
                rows.append(out[i].copy())
            log.append({"key": key, "n": prep["n"], "below": None if carry is None else carry["below"],
                        "in": 0 if carry is None or carry["in"] is None else len(carry["in"]["tags"])})
            if carry is not None:
                cin = carry["in"]
                carry["finished"] = []
                if cin is not None:
                    assert all(k == key for k in cin["key"]), (cin["key"], key)   # a carried sequence arrives in a pass of its own key
                    carry["finished"] = list(zip(cin["tags"], cin["rows"]))
                carry["out"], carry["unfinished_rows"] = None, []
                if carry["below"]:      # the pipeline allows a hand-over: the last own row travels
                    r = prep["n"] - 1
                    carry["out"] = {"tags": [carry["tags"][r]], "rows": [rows[r]], "key": [key]}
                    carry["unfinished_rows"] = [r]
                    out[r] = pad          # (its row of this pass's buffer is incomplete: the pipeline must not read it)
            return torch.from_numpy(out), _DoneEvent()

    settings = [{"max_new_tokens": 64}, {"max_new_tokens": 96}, {"max_new_tokens": 64, "temperature": 0.7, "top_p": 0.9},
                {"max_new_tokens": 64, "temperature": 0.3, "top_p": 0.9}]
    docs = [{"id": i, "label": f"class{i}"} for i in range(20)]
    lm = HostOnly.from_engine(FakeEngine(), tok, batch_size=2)
    lm.task_dict["mix"] = {"test": docs}
    reqs = [TaskInstance(request_type="generate_until", idx=0, metadata={"task": "mix", "doc_id": d["id"], "repeats": 1},
                         arguments=(f"Say something about item {d['id']}.", dict(settings[d["id"] % 4], until=["\n\n"]),
                                    lambda doc: [], d["id"], "mix", "test")) for d in docs]
    try:
        got = lm.generate_until(reqs)
    finally:
        lm._pool.shutdown()
        lm._prep_thread.shutdown()
    want = [tok.decode(answer(lm._prompt_ids(r.args[0], []))[: r.args[1].get("max_new_tokens", 128)]) for r in reqs]
    assert got == want
    assert len({e["key"] for e in log}) == 4 and sum(e["n"] for e in log) == 20
    assert sum(1 for e in log if e["below"]) >= 4                       # hand-overs did happen ...
    for a, b in zip(log, log[1:] + [None]):
        if b is None or b["key"] != a["key"]:
            assert not a["below"], (a, b)                               # ... but the last pass of a key drains its stragglers
        if b is not None and b["key"] != a["key"]:
            assert b["in"] == 0, (a, b)                                 # and nothing crosses into a pass of another key
