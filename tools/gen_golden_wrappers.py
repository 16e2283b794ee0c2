#!/usr/bin/env python3
"""Golden vectors for the generation PROTOCOL of the two model wrappers - what the host loop hands the model and what it returns -
produced by RUNNING THE REFERENCE's own `Qwen2VL.generate_until_multi_round` (/root/reference/src/models/_qwen2_vl.py:350-616),
`LLaVA.generate_until_multi_round` (/root/reference/src/models/_llava_hf.py:394-584) and the single-round `generate_until` of both
(`_qwen2_vl.py:143-348`, `_llava_hf.py:260-392`: the hot path's host loop) in this container.  Writes

  tests/golden/wrapper_protocol.json

What runs is the reference's code: its Collator, its round loop, what it hands to / takes back from the task's `doc_to_text`
(`previous_round_results`, `last_round_info`), its message lists, the `until` cut, the result tuples and their order.  What is
replaced is only what is absent offline: the checkpoint (a model whose `generate` answers with a deterministic function of the
rendered prompt text), the HF processor (renders with the chat template published with the checkpoints, tokenises one id per
character) and `qwen_vl_utils.process_vision_info` (collects the image entries of the messages).  `src.models.__init__` is not
executed (it imports every wrapper: torchvision, llava, ...): the two wrapper modules are imported as submodules of a bare
package.  The task side (docs, `doc_to_visual`, `doc_to_text`) is test INPUT written here, not reference code.

    python tools/gen_golden_wrappers.py      # needs /root/reference; the fixture travels, the reference does not

The recorded trace per request and round - rendered prompt text, number of images handed to the processor, generation
arguments - and the returned tuples are what tests/test_host_logic.py checks oracle/multiround.py against (and through it the
product's `generate_until_multi_round`, tests/test_host_logic.py::test_multi_round_generation_follows_the_reference_protocol).
"""
from __future__ import annotations

import hashlib
import importlib
import importlib.machinery
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests import recipes  # noqa: E402
from tools import gen_golden as G  # noqa: E402

GOLD = ROOT / "tests" / "golden"
REF = Path("/root/reference")


answer_of, image_of, docs_and_task = recipes.mr_answer_of, recipes.mr_image_of, recipes.mr_docs_and_task


class CharTokenizer:
    """One id per character (+ an EOS id): enough for the reference's Collator key and its decode of the EOT token."""
    eos_token_id = 1
    pad_token_id = 0

    def encode(self, text, add_special_tokens=None):
        return [2 + (ord(c) % 250) for c in text]

    def decode(self, ids):
        if isinstance(ids, int):
            ids = [ids]
        return "".join("<|im_end|>" if int(i) == 1 else "?" for i in ids)


class FakeInputs(dict):
    def to(self, *a, **k):
        return self

    @property
    def input_ids(self):
        return self["input_ids"]


def make_requests(TaskInstance, docs, doc_to_visual, doc_to_text, gen_kwargs):
    reqs = []
    for d in docs:
        args = (recipes.mr_context(d), dict(gen_kwargs), doc_to_visual, doc_to_text, d["id"], "mr", "test")
        try:
            inst = TaskInstance(request_type="generate_until_multi_round", arguments=args, idx=0,
                                metadata={"task": "mr", "doc_id": d["id"], "repeats": 1})
        except TypeError:
            inst = types.SimpleNamespace(args=args)
        reqs.append(inst)
    return reqs


def import_wrappers():
    G.import_reference()
    from transformers import AutoProcessor, AutoTokenizer, Qwen2VLForConditionalGeneration  # noqa: F401  (before any stub exists)

    for name in ("qwen_vl_utils", "llava"):
        if name not in sys.modules:
            mod = G._Stub(name)
            mod.__spec__ = importlib.machinery.ModuleSpec(name, None)
            mod.__path__ = []
            sys.modules[name] = mod
    pkg = types.ModuleType("src.models")
    pkg.__path__ = [str(REF / "src" / "models")]
    pkg.__spec__ = importlib.machinery.ModuleSpec("src.models", None, is_package=True)
    sys.modules["src.models"] = pkg      # (its __init__ imports every wrapper; only these two modules are needed)
    qm = importlib.import_module("src.models._qwen2_vl")
    lm = importlib.import_module("src.models._llava_hf")
    from src.data.tasks import TaskInstance

    return qm, lm, TaskInstance


def run_qwen(qm, TaskInstance, gen_kwargs):
    from oracle.multiround import render_qwen2vl_chat

    trace = []

    def process_vision_info(messages):
        imgs = [c for msg in messages for turn in msg if isinstance(turn.get("content"), list) for c in turn["content"]
                if c.get("type") == "image"]
        return (imgs or None), None

    qm.process_vision_info = process_vision_info
    docs, d2v, d2t = docs_and_task()

    class Processor:
        def apply_chat_template(self, msg, tokenize=False, add_generation_prompt=True):
            return render_qwen2vl_chat(msg, add_generation_prompt)

        def __call__(self, text=None, images=None, videos=None, padding=True, return_tensors="pt"):
            self.last_texts = list(text)
            self.last_images = 0 if images is None else len(images)
            ids = [CharTokenizer().encode(t) for t in text]
            width = max(len(i) for i in ids)
            return FakeInputs(input_ids=torch.tensor([[0] * (width - len(i)) + i for i in ids]))

        def batch_decode(self, seqs, skip_special_tokens=True, clean_up_tokenization_spaces=False):
            return [self.answers[int(s[0])] for s in seqs]

    proc = Processor()

    class Net:
        device = torch.device("cpu")

        def generate(self, input_ids=None, **kw):
            proc.answers = [answer_of(t) for t in proc.last_texts]
            trace.append({"texts": proc.last_texts, "images": proc.last_images,
                          "generate_kwargs": {k: (v if not isinstance(v, torch.Tensor) else "tensor") for k, v in sorted(kw.items())}})
            new = torch.tensor([[i] for i in range(len(proc.answers))])
            return torch.cat([input_ids, new], dim=1)

    obj = object.__new__(qm.Qwen2VL)
    tok = CharTokenizer()
    fields = {"_tokenizer": tok, "_processor": proc, "processor": proc, "_model": Net(), "batch_size_per_gpu": 1, "_rank": 0,
              "_world_size": 1, "device_map": "cpu", "_device": torch.device("cpu"), "_use_cache": True,
              "task_dict": {"mr": {"test": docs}}, "cache_hook": types.SimpleNamespace(add_partial=lambda *a, **k: None)}
    for k, v in fields.items():
        try:
            object.__setattr__(obj, k, v)
        except AttributeError:
            pass
    res = obj.generate_until_multi_round(make_requests(TaskInstance, docs, d2v, d2t, gen_kwargs))
    return {"results": [list(r) for r in res], "trace": trace, "gen_kwargs": gen_kwargs,
            "contexts": [r.args[0] for r in make_requests(TaskInstance, docs, d2v, d2t, gen_kwargs)]}


def run_llava(lm, TaskInstance, gen_kwargs):
    """The reference's LLaVA.generate_until_multi_round (src/models/_llava_hf.py:440-584) on the same task and stand-in model; the
    tokenizer renders with the module's own VICUNA_CHAT_TEMPLATE (the reference's fall-back when the checkpoint has none)."""
    from jinja2.sandbox import ImmutableSandboxedEnvironment

    trace = []
    docs, d2v, d2t = docs_and_task()

    class Tok(CharTokenizer):
        chat_template = None

        def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
            return env.from_string(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt,
                                                              bos_token="<s>", eos_token="</s>")

        def batch_decode(self, cont, skip_special_tokens=True):
            return [proc.answers[int(s[0])] for s in cont]

    class Processor:
        def __call__(self, images=None, text=None, return_tensors="pt"):
            self.last_texts = [text] if isinstance(text, str) else list(text)
            self.last_images = 0 if images is None else len(images)
            ids = [CharTokenizer().encode(t) for t in self.last_texts]
            return FakeInputs(input_ids=torch.tensor(ids))

    proc = Processor()

    class Net:
        device, dtype = torch.device("cpu"), torch.float32

        def generate(self, input_ids=None, **kw):
            proc.answers = [answer_of(t) for t in proc.last_texts]
            trace.append({"texts": proc.last_texts, "images": proc.last_images,
                          "generate_kwargs": {k: (v if not isinstance(v, torch.Tensor) else "tensor") for k, v in sorted(kw.items())
                                              if k != "image_sizes"}})
            return torch.cat([input_ids, torch.tensor([[i] for i in range(len(proc.answers))])], dim=1)

    obj = object.__new__(lm.LLaVA)
    fields = {"_tokenizer": Tok(), "_processor": proc, "processor": proc, "_model": Net(), "batch_size_per_gpu": 1, "_rank": 0,
              "_world_size": 1, "device_map": "cpu", "_device": torch.device("cpu"), "_use_cache": True, "_chat_template": None,
              "task_dict": {"mr": {"test": docs}}, "cache_hook": types.SimpleNamespace(add_partial=lambda *a, **k: None),
              "accelerator": types.SimpleNamespace(is_main_process=False, unwrap_model=lambda m: m), "_device_map": "cpu"}
    for k, v in fields.items():
        try:
            object.__setattr__(obj, k, v)
        except AttributeError:
            pass
    res = obj.generate_until_multi_round(make_requests(TaskInstance, docs, d2v, d2t, gen_kwargs))
    return {"results": [list(r) for r in res], "trace": trace, "gen_kwargs": gen_kwargs,
            "contexts": [r.args[0] for r in make_requests(TaskInstance, docs, d2v, d2t, gen_kwargs)]}


def make_single_requests(TaskInstance, docs, doc_to_visual, gen_kwargs):
    reqs = []
    for d in docs:
        gk = gen_kwargs[d["id"] % len(gen_kwargs)] if isinstance(gen_kwargs, list) else gen_kwargs   # a list: mixed generation settings
        args = (recipes.su_context(d), dict(gk), doc_to_visual, d["id"], "su", "test")
        reqs.append(TaskInstance(request_type="generate_until", arguments=args, idx=0, metadata={"task": "su", "doc_id": d["id"], "repeats": 1}))
    return reqs


def run_single(module, which: str, TaskInstance, gen_kwargs):
    """The reference's single-round `generate_until` (the hot path's host loop: _qwen2_vl.py:143-348 / _llava_hf.py:260-392) on the
    stand-in checkpoint: six requests, the trace of generate calls and the returned strings in request order."""
    box = {}
    # (the wrapper object and its stand-ins are built by run_qwen / run_llava; the method they call is swapped for `generate_until`)
    runner = run_qwen if which == "qwen" else run_llava
    cls = module.Qwen2VL if which == "qwen" else module.LLaVA
    docs, d2v = recipes.su_docs_and_task()
    orig = cls.generate_until_multi_round

    def call_single(self, _requests):
        self.task_dict = {"su": {"test": docs}}
        box["res"] = self.generate_until(make_single_requests(TaskInstance, docs, d2v, gen_kwargs))
        return [(r,) for r in box["res"]]

    cls.generate_until_multi_round = call_single
    try:
        out = runner(module, TaskInstance, gen_kwargs[0] if isinstance(gen_kwargs, list) else gen_kwargs)
    finally:
        cls.generate_until_multi_round = orig
    return {"results": list(box["res"]), "trace": out["trace"], "gen_kwargs": gen_kwargs,
            "contexts": [recipes.su_context(d) for d in docs]}


def run_llava_loglik(lm, TaskInstance):
    """The reference's LLaVA.loglikelihood (src/models/_llava_hf.py:169-258) on a stand-in decoder (logits = recipes.ll_logits of the
    ids, loss = HF's mean shifted cross-entropy over the unmasked labels): per request the two rendered texts, the image count, how
    many leading positions the labels mask covers, the sequence length the model saw, and the returned (loss, greedy flag)."""
    from jinja2.sandbox import ImmutableSandboxedEnvironment

    from lmms_owc_amd.engine.llava import DIMS
    from lmms_owc_amd.models._llava_hf import LlavaByteTokenizer

    btok, n_per_image = LlavaByteTokenizer(), DIMS["tiny"].grid ** 2
    trace = []
    docs, d2v = recipes.su_docs_and_task()

    class Tok(CharTokenizer):
        chat_template = None

        def apply_chat_template(self, messages, tokenize=False, add_generation_prompt=True):
            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
            return env.from_string(self.chat_template).render(messages=messages, add_generation_prompt=add_generation_prompt,
                                                              bos_token="<s>", eos_token="</s>")

    class Processor:
        def __call__(self, text=None, images=None, return_tensors="pt"):
            ids = btok.encode(text[0], add_special_tokens=True)
            if images:   # (HF's LlavaProcessor expands every <image> to the image's feature count only when images are given)
                ids = [t for i in ids for t in ([i] * n_per_image if i == btok.image_token_id else [i])]
            self.last = {"text": text[0], "images": 0 if images is None else len(images)}
            return FakeInputs(input_ids=torch.tensor([ids]))

    proc = Processor()

    class Net:
        device, dtype = torch.device("cpu"), torch.float32

        def __call__(self, input_ids=None, labels=None, **kw):
            logits = torch.from_numpy(recipes.ll_logits(input_ids[0].numpy()))[None]
            shift_logits, shift_labels = logits[0, :-1], labels[0, 1:]
            loss = torch.nn.functional.cross_entropy(shift_logits, shift_labels.long(), ignore_index=-100)
            trace[-1].update({"masked_leading_positions": int((labels[0] == -100).sum()), "sequence_length": int(input_ids.shape[1])})
            return {"loss": loss, "logits": logits}

    obj = object.__new__(lm.LLaVA)
    fields = {"_tokenizer": Tok(), "_processor": proc, "processor": proc, "_model": Net(), "batch_size_per_gpu": 1, "_rank": 0,
              "_world_size": 1, "_device_map": "cpu", "_use_cache": True, "_chat_template": None, "task_dict": {"su": {"test": docs}},
              "accelerator": types.SimpleNamespace(is_main_process=False, unwrap_model=lambda m: m)}
    for k, v in fields.items():
        try:
            object.__setattr__(obj, k, v)
        except AttributeError:
            pass
    reqs = []
    for d in docs:
        target = recipes.ll_continuation(d) if d["id"] % 2 else (lambda doc: recipes.ll_continuation(doc))   # str and callable targets
        reqs.append(TaskInstance(request_type="loglikelihood", arguments=(recipes.ll_context(d), target, d2v, d["id"], "su", "test"), idx=0,
                                 metadata={"task": "su", "doc_id": d["id"], "repeats": 1}))
    orig_call = proc.__class__.__call__

    def spy(self, text=None, images=None, return_tensors="pt"):
        out = orig_call(self, text=text, images=images, return_tensors=return_tensors)
        if images is not None:
            trace.append({"prompt_and_continuation": text[0], "images": len(images)})
        else:
            trace[-1]["prompt"] = text[0]
        return out

    proc.__class__.__call__ = spy
    res = obj.loglikelihood(reqs)
    return {"results": [[float(a), bool(b)] for a, b in res], "trace": trace, "contexts": [recipes.ll_context(d) for d in docs],
            "continuations": [recipes.ll_continuation(d) for d in docs], "tokens_per_image": n_per_image}


def main():
    qm, lm, TaskInstance = import_wrappers()
    out = {"versions": G.versions(),
           "what": "reference Qwen2VL.generate_until_multi_round (src/models/_qwen2_vl.py:350-616) run on a stand-in checkpoint: "
                   "per generate call the rendered prompts, the image count handed to the processor and the generation arguments; "
                   "the returned per-request tuples of round answers (original request order)",
           "qwen2vl": [run_qwen(qm, TaskInstance, gk) for gk in (
               {"max_new_tokens": 48, "do_sample": False, "until": ["STOP"]},
               {"until": "STOP"},
               {"max_new_tokens": 16, "temperature": 0})],
           "llava": [run_llava(lm, TaskInstance, gk) for gk in ({"max_new_tokens": 48, "do_sample": False, "until": ["STOP"]}, {})],
           "qwen2vl_single": [run_single(qm, "qwen", TaskInstance, gk) for gk in (
               {"max_new_tokens": 64, "do_sample": False, "until": ["STOP"]}, {}, {"max_new_tokens": 64, "temperature": 0.8, "top_p": 0.9},
               [{"max_new_tokens": 64, "do_sample": False}, {"max_new_tokens": 96, "do_sample": False, "until": ["STOP"]}])],
           "llava_single": [run_single(lm, "llava", TaskInstance, gk) for gk in (
               {"max_new_tokens": 64, "do_sample": False, "until": ["STOP"]}, {}, {"max_new_tokens": 64, "temperature": 0.8, "top_p": 0.9},
               [{"max_new_tokens": 64, "do_sample": False}, {"max_new_tokens": 96, "do_sample": False, "until": ["STOP"]}])],
           "llava_loglikelihood": run_llava_loglik(lm, TaskInstance)}
    GOLD.mkdir(parents=True, exist_ok=True)
    (GOLD / "wrapper_protocol.json").write_text(json.dumps(out, indent=1, sort_keys=True) + "\n")
    print("wrote", GOLD / "wrapper_protocol.json", {k: len(v) if isinstance(v, list) else "-" for k, v in out.items()})


if __name__ == "__main__":
    main()
