"""oracle/multiround.py pinned on the REFERENCE's own run (tests/golden/multiround_protocol.json, written by
tools/gen_golden_multiround.py: /root/reference/src/models/_qwen2_vl.py:350-616 and _llava_hf.py:440-584 executed in the build
container on a stand-in checkpoint whose answer is a function of the rendered prompt).  The restated protocol, driven with the same
stand-in, must hand the model the same prompts with the same number of images, round by round, and return the same tuples - then
tests/test_host_logic.py's comparison of the product with this oracle is a comparison with the reference's behaviour."""
import json
from pathlib import Path

import pytest

from tests import recipes

GOLD = Path(__file__).parent / "golden" / "multiround_protocol.json"


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
