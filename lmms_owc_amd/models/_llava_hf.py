"""`LLaVA` model plug-in on the MI355X engine — drop-in for /root/reference/src/models/_llava_hf.py.

Same constructor kwargs (`:64-76`), same registry names (`:586-615`) and the same `generate_until` contract
(`:260-392`: prepend `<image>` tokens when the context has none, chat template with the Vicuna fallback, greedy
decode until the tokenizer's EOS, `list[TaskInstance] -> list[str]` in request order), covering both families the
reference's `load_model` maps (`:130-147`): LLaVA-1.5 (one 336x336 CLIP view) and LLaVA-NeXT / 1.6 (anyres tiling).
Requests are BATCHED (the reference runs batch size 1): every view of a chunk goes through the HIP CLIP tower
together, prompts are prefilled packed, decode runs the whole batch per step.
`model_name_or_path="synthetic:<name>"` (names of engine/llava.py DIMS) builds seeded random weights + a byte
tokenizer; a local HF checkpoint directory loads real weights, tokenizer and chat template.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from .. import _lib, utils
from ..engine.llava import DIMS, NEXT_PINPOINTS, LlavaDims, LlavaEngine, LlavaWeights
from . import imageproc
from ._api import register_model
from ._base import Model, PassPipeline, beams_from_gen_kwargs, pass_key, sampling_from_gen_kwargs
from ._qwen2_vl import ByteTokenizer, LazyCheckpoint

__all__ = ["LLaVA"]

DEFAULT_IMAGE_TOKEN = "<image>"  # noqa: S105
_VICUNA_SYSTEM = ("A chat between a curious user and an artificial intelligence assistant. The assistant gives helpful, "
                  "detailed, and polite answers to the user's questions.")


def _flatten(nested: list) -> list:
    return [x for sub in nested for x in sub]


def vicuna_prompt(messages: list[dict], eos_token: str = "</s>", add_generation_prompt: bool = True) -> str:
    """Text the reference's fallback chat template (`_llava_hf.py:23`) renders: system preamble before the first
    turn, `USER: ... ` / ` ASSISTANT: ...</s>` turns, and a trailing `ASSISTANT:` generation prompt."""
    out = []
    for i, m in enumerate(messages):
        if i == 0:
            out.append(f"{_VICUNA_SYSTEM} USER: {m['content']} ")
        elif m["role"] == "user":
            out.append(f"USER: {m['content']} ")
        else:
            out.append(f" ASSISTANT: {m['content']}{eos_token}")
    if add_generation_prompt:
        out.append("ASSISTANT:")
    return "".join(out)


class LlavaByteTokenizer(ByteTokenizer):
    """Synthetic-run tokenizer: UTF-8 bytes + 3, BOS = 2, `<image>` = one special id above the byte range."""

    bos_token_id, image_token_id = 2, 264
    eos_token, chat_template = "</s>", None

    def encode(self, text: str, add_special_tokens: bool = False) -> list[int]:
        ids = [self.bos_token_id] if add_special_tokens else []
        for i, part in enumerate(text.split(DEFAULT_IMAGE_TOKEN)):
            if i:
                ids.append(self.image_token_id)
            ids += [b + 3 for b in part.encode("utf-8")]
        return ids


class LLaVA(PassPipeline, Model):
    def __init__(self, model_name_or_path: str = "llava-hf/llava-1.5-7b-hf", attn_implementation: str | None = None,
                 chat_template: str | None = None, use_cache: bool = True, batch_size: int = 1, device_map: str = "auto",
                 dtype: str | torch.dtype = "bfloat16", load_in_8bit: bool = False, load_in_4bit: bool = False,
                 decoder_dtype: str = "bf16", engine_batch: int | str = "auto", **kwargs) -> None:
        # two kwargs beyond the reference's: fp8 decoder projections and `engine_batch` - `batch_size` is a
        # lower bound, tokens are batch-invariant (see Qwen2VL.__init__): "auto" = what fits in HBM (<= 512), n = exactly n, 0 = off
        if decoder_dtype not in ("bf16", "fp8"):
            raise ValueError("decoder_dtype must be 'bf16' or 'fp8'")
        if engine_batch != "auto" and (not isinstance(engine_batch, int) or engine_batch < 0):
            raise ValueError("engine_batch must be 'auto' or an integer >= 0")
        self._engine_batch_arg = engine_batch
        self._decoder_dtype = decoder_dtype
        self._model_name_or_path = model_name_or_path
        self._attn_implementation = attn_implementation  # accepted; attention is always the fused HIP kernel
        self._chat_template = chat_template
        self._use_cache = use_cache                      # the HIP decoder always uses its KV cache
        super().__init__(batch_size=batch_size, device_map=device_map, dtype=dtype, load_in_8bit=load_in_8bit,
                         load_in_4bit=load_in_4bit, distributed_types=["FSDP", "MULTI_GPU", "DEEPSPEED"], **kwargs)

    # ------------------------------------------------------------------ loading
    def load_model(self) -> None:
        name = self._model_name_or_path
        if name.startswith("synthetic:"):
            tok = LlavaByteTokenizer()
            dims = LlavaDims(**{**DIMS[name.split(":", 1)[1]].__dict__, "image_token_id": tok.image_token_id,
                                "decoder_dtype": self._decoder_dtype})
            weights = LlavaWeights.random(dims, self._device, seed=1234)
            self._tokenizer = tok
        else:
            path = Path(name)
            if not path.is_dir():
                from huggingface_hub import snapshot_download

                path = Path(snapshot_download(name))
            dims = dims_from_hf_config(json.loads((path / "config.json").read_text()))
            dims = LlavaDims(**{**dims.__dict__, "decoder_dtype": self._decoder_dtype})
            weights = LlavaWeights.from_state_dict(dims, LlavaCheckpoint(path), self._device)
            from transformers import AutoTokenizer

            self._tokenizer = AutoTokenizer.from_pretrained(str(path))
            self._tokenizer.padding_side = "left"  # as the reference (:163); prompts are packed, never padded, here
        self._dims = dims
        self._start_workers()     # PIL worker pool + preparation thread + pinned staging (PassPipeline)
        self._model = LlavaEngine(weights)
        self._processor = self._tokenizer

    def engine_batch(self, max_new_tokens: int = 64) -> int:
        """Requests per engine pass (see Qwen2VL.engine_batch): KV cache + projected features for the largest image the
        processor produces (5 anyres views, or one 336-px view), a quarter of the free HBM, at most 512."""
        arg = self._engine_batch_arg
        if arg == 0:
            return self.batch_size
        if arg != "auto":
            return max(self.batch_size, int(arg))
        d = self._dims
        views = 5 if d.grid_pinpoints else 1
        tokens = views * d.tokens + 128 + int(max_new_tokens)
        per_req = (d.n_layers * 2 * d.n_kv_heads * d.head_dim * 2) * tokens + views * d.tokens * (d.d_model * 2 + d.v_embed * 24)
        cache = self.__dict__.setdefault("_auto_batch", {})   # once per max_new_tokens (see Qwen2VL.engine_batch)
        if max_new_tokens not in cache:
            free, _ = torch.cuda.mem_get_info(self._device)
            free += torch.cuda.memory_reserved(self._device) - torch.cuda.memory_allocated(self._device)
            free += getattr(self._model, "held_bytes", lambda: 0)()   # the engine's own K / V pair + workspace are reused by the next pass
            cache[max_new_tokens] = max(self.batch_size, min(512, int(0.25 * free / per_req)))
        return cache[max_new_tokens]

    def loglikelihood(self, requests: list) -> list[tuple[float, bool]]:
        """(loss, greedy-match) per request, with the reference's semantics (src/models/_llava_hf.py:169-258): request args are
        (context, doc_to_target | str, doc_to_visual, doc_id, task, split); `<image>` tokens are ALWAYS prepended (:204-206);
        prompt = template(user turn, generation prompt), prompt + continuation = template(user, assistant) without one (:213-227);
        the labels mask is the length of the prompt tokenised WITHOUT image expansion (:231-232: expanded image positions and the
        rest of the prompt stay in the loss), loss = HF's mean shifted cross-entropy, and the greedy flag compares the UNSHIFTED
        argmax with the ids (:246-251).  One prefill per request (owc_llm_prefill's scoring mode), batch size 1 like the reference."""
        res = []
        eng, tok = self._model, self._tokenizer
        for context, doc_to_target, doc_to_visual, doc_id, task, split in [r.args for r in requests]:
            doc = self.task_dict[task][split][doc_id]
            continuation = doc_to_target if isinstance(doc_to_target, str) else doc_to_target(doc)
            visuals = _flatten([doc_to_visual(doc)])
            context = f"{' '.join([DEFAULT_IMAGE_TOKEN] * len(visuals))}\n{context}"
            messages = [{"role": "user", "content": context}, {"role": "assistant", "content": continuation}]
            prompt = self._render_messages(messages[:-1], True)
            full = self._render_messages(messages, False)
            feats, rows = self._encode_visuals(visuals)
            ids = self._expand(tok.encode(full, add_special_tokens=True), [len(r) for r in rows])
            n_ctx = len(tok.encode(prompt, add_special_tokens=True))
            if not 1 <= n_ctx < len(ids):
                raise ValueError("loglikelihood: empty continuation")
            lp, top = eng.score(ids, feats, [], n_ctx, img_rows=np.concatenate(rows) if rows else np.zeros(0, np.int64))
            res.append((float(-np.mean(lp, dtype=np.float64)), bool(np.array_equal(top, ids[n_ctx:]))))
        return res

    def generate_until_multi_round(self, requests: list) -> list[tuple]:
        """Multi-round generation (/root/reference/src/models/_llava_hf.py:394-584; the `*_llamav_o1` task configs).  As the
        reference warns, this wrapper keeps NO history: every round is an independent single-turn prompt built from what the
        task's `doc_to_text(doc, round_idx=, previous_round_results=, last_round_info=)` returns for that round - its visuals
        and its context (`<image>` tokens prepended when the context has none, the same chat template as `generate_until`),
        greedy, EOS only (`until` is read and never applied, :455-466); a document's rounds stop at ITS terminal signal (the
        reference's batch is one document, :496); the result per request is the tuple of per-round answers.  Batched over documents; an
        image that appears in several rounds goes through the host-side resize / anyres tiling once.
        One deliberate difference: the classification tasks' `doc_to_text_multi_round` returns `visual = None` after round 0
        (`_caltech101_utils.py:66-72`); the reference then evaluates `list(*visuals)` on `(None,)` and raises TypeError (:500-501),
        i.e. it cannot finish those tasks with this wrapper.  Here such a round is a text-only prompt."""
        res: list[tuple] = []
        tok = self._tokenizer

        def _collate(x):
            return -len(tok.encode(x[0], add_special_tokens=False)), x[0]

        reordered = utils.Collator([reg.args for reg in requests], _collate, grouping=True)
        max_new_all = max([int(r.args[1].get("max_new_tokens", 1024)) for r in requests] + [1])
        for chunk in reordered.get_batched(n=self.engine_batch(max_new_all), batch_fn=None):
            contexts, all_gen_kwargs, doc_to_visual, doc_to_text, doc_ids, tasks, splits = zip(*chunk, strict=True)
            task, split = tasks[0], splits[0]
            gen_kwargs = dict(all_gen_kwargs[0])
            until = gen_kwargs.pop("until", None)
            if until is not None and not isinstance(until, (str, list)):
                raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str,list] but got {type(until)}")
            for g in all_gen_kwargs:
                g.pop("until", None)
            max_new = int(gen_kwargs.get("max_new_tokens", 1024))
            sampling = sampling_from_gen_kwargs(gen_kwargs, getattr(self, "_default_top_k", 50))
            docs = [self.task_dict[task][split][did] for did in doc_ids]
            visuals_per_doc = [list(doc_to_visual[0](d)) for d in docs]
            contexts = list(contexts)
            feature_cache: dict = {}            # id(PIL image) -> (views uint8, original size): a round's repeat is not re-encoded
            results: list[list[str]] = [[] for _ in docs]   # [doc] -> its per-round answers so far
            active, round_idx = list(range(len(docs))), 0
            while True:
                if round_idx:
                    # every document follows ITS OWN terminal signal (the reference runs one document per batch, :496): a finished
                    # document leaves the following rounds, its result is the list its task returned last
                    still = []
                    for i in active:
                        vis, ctx, terminal, rr, _info = doc_to_text[0](docs[i], round_idx=round_idx, previous_round_results=list(results[i]),
                                                                       last_round_info=None)
                        # the per-round results continue from what the task RETURNS, as in the reference (:496-506): a task may
                        # rewrite or truncate earlier answers
                        results[i] = list(rr)
                        if terminal:
                            continue
                        visuals_per_doc[i] = list(vis) if vis is not None else []
                        contexts[i] = ctx
                        still.append(i)
                    active = still
                if not active:
                    break
                smp = None if sampling is None else {**sampling, "stream_ids": [int(doc_ids[i]) * 64 + round_idx for i in active]}
                rows = self._generate_chunk([contexts[i] for i in active], [visuals_per_doc[i] for i in active], max_new, feature_cache,
                                            sampling=smp, num_beams=beams_from_gen_kwargs(gen_kwargs))
                for i, ans in zip(active, self.decode_tokens(rows)):
                    results[i].append(ans)
                round_idx += 1
            res.extend(tuple(r) for r in results)
            round_results = [list(r) for r in results]
            self.cache_hook.add_partial("generate_until_multi_round", (contexts[0], gen_kwargs), round_results)
        return reordered.get_original(res)

    # ------------------------------------------------------------------ prompt building
    def _render(self, context: str) -> str:
        return self._render_messages([{"role": "user", "content": context}], True)

    def _render_messages(self, messages: list[dict], add_generation_prompt: bool) -> str:
        tok = self._tokenizer
        template = self._chat_template if self._chat_template is not None else getattr(tok, "chat_template", None)
        if template is None:
            return vicuna_prompt(messages, getattr(tok, "eos_token", "</s>") or "</s>", add_generation_prompt)
        if isinstance(tok, ByteTokenizer):
            from jinja2.sandbox import ImmutableSandboxedEnvironment

            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
            return env.from_string(template).render(messages=messages, add_generation_prompt=add_generation_prompt, eos_token=tok.eos_token)
        tok.chat_template = template
        return tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=add_generation_prompt)

    def _prompt_ids(self, context: str, n_image_tokens: list[int]) -> np.ndarray:
        """Rendered prompt -> ids with every `<image>` placeholder expanded to that image's feature count
        (what LlavaProcessor / LlavaNextProcessor do on the text before tokenising)."""
        return self._expand(self._tokenizer.encode(self._render(context), add_special_tokens=True), n_image_tokens)

    def _expand(self, ids, n_image_tokens: list[int]) -> np.ndarray:
        out, it = [], iter(n_image_tokens)
        for t in ids:
            if t == self._dims.image_token_id:
                out.extend([t] * next(it))
            else:
                out.append(t)
        if next(it, None) is not None:
            raise ValueError("fewer <image> placeholders in the prompt than images in the request")
        return np.asarray(out, dtype=np.int32)

    def _views(self, img):
        d = self._dims
        if d.grid_pinpoints:
            return imageproc.anyres_views(img, d.grid_pinpoints, d.image_size)
        return imageproc.clip_view(img, d.image_size)[None], (img.size[1], img.size[0])

    # ------------------------------------------------------------------ the hot loop
    def generate_until(self, requests: list) -> list[str]:
        answers = self.decode_tokens(self._generate_rows(requests))
        for req, ans in zip(requests, answers):
            self.cache_hook.add_partial("generate_until", (req.args[0], req.args[1]), ans)
        return answers

    def generate_until_tokens(self, requests: list) -> tuple[np.ndarray, np.ndarray]:
        """Fixed-width token records for the engine's end-of-task RCCL gather (see Qwen2VL.generate_until_tokens)."""
        rows = self._generate_rows(requests)
        T = max([1] + [len(r) for r in rows])
        mat = np.zeros((len(rows), T), np.int32)
        for i, r in enumerate(rows):
            mat[i, : len(r)] = r
        return mat, np.array([len(r) for r in rows], np.int32)

    def decode_tokens(self, rows: list) -> list[str]:
        return self._tokenizer.batch_decode([np.asarray(r) for r in rows], skip_special_tokens=True)

    def _encode_visuals(self, flat: list, feature_cache: dict | None = None):
        """Images -> (projected CLIP feature rows of all their views, the feature-row list of every image in prompt order)."""
        eng = self._model
        if feature_cache is None:
            prepared = list(self._pool.map(self._views, flat))  # PIL resampling releases the GIL
        else:
            new = [v for v in {id(v): v for v in flat}.values() if id(v) not in feature_cache]
            for v, p in zip(new, self._pool.map(self._views, new)):
                feature_cache[id(v)] = (v, p)               # the image object is kept alive with its id
            prepared = [feature_cache[id(v)][1] for v in flat]
        views_per_image = [p[0].shape[0] for p in prepared]
        sizes = [p[1] for p in prepared]
        feats, rows = None, []
        if prepared:
            u8 = _lib.h2d(np.concatenate([p[0] for p in prepared]), self._device)
            feats = eng.encode_views(eng.patchify(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD))
            rows = eng.feature_rows(views_per_image, sizes)
        return feats, rows

    def _generate_chunk(self, contexts, visuals_per_doc, max_new: int, feature_cache: dict | None = None,
                        sampling: dict | None = None, num_beams: int = 1) -> list[np.ndarray]:
        """One engine pass: single-turn prompts (context + that document's images) -> greedy token rows cut at EOS."""
        eng, tok = self._model, self._tokenizer
        feats, rows = self._encode_visuals([v for vs in visuals_per_doc for v in vs], feature_cache)
        prompts, rows_per_prompt, cur = [], [], 0
        for ctx, visuals in zip(contexts, visuals_per_doc):
            mine = rows[cur:cur + len(visuals)]
            cur += len(visuals)
            if DEFAULT_IMAGE_TOKEN not in ctx:  # e.g. classification prompts carry no image token (:324-327, :506-509)
                ctx = f"{' '.join([DEFAULT_IMAGE_TOKEN] * len(visuals))}\n{ctx}"
            prompts.append(self._prompt_ids(ctx, [len(r) for r in mine]))
            rows_per_prompt.append(np.concatenate(mine) if mine else np.zeros(0, np.int64))
        eos = tok.eos_token_id
        if num_beams > 1:
            out = eng.generate_beam(prompts, feats, [[] for _ in prompts], max_new, num_beams, eos_token_id=eos, pad_token_id=eos,
                                    img_rows=rows_per_prompt).cpu().numpy()
        else:
            out = eng.generate_from_features(prompts, feats, rows_per_prompt, max_new, eos_token_id=eos, pad_token_id=eos,
                                             sampling=sampling).cpu().numpy()
        res = []
        for r in out:
            stop = np.flatnonzero(r == eos)
            res.append(r[: stop[0]] if len(stop) else r)
        return res

    def _generate_rows(self, requests: list) -> list[np.ndarray]:
        """Token rows (cut at EOS) per request, in request order, through the two-stage pass pipeline of `_base.PassPipeline`
        (round 4: the host work of a chunk - CLIP resize / crop or anyres tiling, prompt ids - used to run in series with its
        GPU work; LLaVA-1.5-7B from PIL images: 102 -> see tools/bench_llava_pil.py)."""
        return self._run_passes(requests, default_max_new=1024)

    def _collate_encode(self, text: str):
        return self._tokenizer.encode(text, add_special_tokens=False)

    def _prepare_chunk(self, chunk) -> dict:
        """HOST stage of one unit (preparation thread; the views are cut on the PIL pool): per image its uint8 views and size, per
        request its prompt ids with every `<image>` expanded to that image's feature count, all views stacked in pinned memory."""
        contexts, all_gen_kwargs, doc_to_visual, doc_ids, tasks, splits = zip(*chunk, strict=True)
        task, split = tasks[0], splits[0]
        for g in all_gen_kwargs:   # the reference (batch size 1) pops it from EVERY request's own dict, which is what the
            g.pop("until", None)   # samples file later records under `arguments` (_engine.py:262-266, _tracker.py:318-322)
        gen_kwargs = dict(all_gen_kwargs[0])
        gen_kwargs.pop("until", None)  # read and never applied by the reference (:310-320)
        max_new = int(gen_kwargs.get("max_new_tokens", 1024))
        sampling = sampling_from_gen_kwargs(gen_kwargs, getattr(self, "_default_top_k", 50))
        docs = self.task_dict[task][split]
        eng = self._model

        def fetch(did):
            return [self._views(v) for v in doc_to_visual[0](docs[did])]

        per_doc = list(self._pool.map(fetch, doc_ids))
        prompts, images_per_prompt, views, sizes = [], [], [], []
        for ctx, imgs in zip(contexts, per_doc):
            counts = []
            for v, size in imgs:
                counts.append(len(eng.feature_rows([v.shape[0]], [size])[0]))
                views.append(v)
                sizes.append(size)
            if DEFAULT_IMAGE_TOKEN not in ctx:  # e.g. classification prompts carry no image token (:324-327, :506-509)
                ctx = f"{' '.join([DEFAULT_IMAGE_TOKEN] * len(imgs))}\n{ctx}"
            prompts.append(self._prompt_ids(ctx, counts))
            images_per_prompt.append(len(imgs))
        groups = []
        if views:
            n_v = sum(v.shape[0] for v in views)
            buf = self._pinned_take((n_v, *views[0].shape[1:]))
            dst, offs = buf.numpy(), np.cumsum([0] + [v.shape[0] for v in views])
            list(self._pool.map(lambda k: np.copyto(dst[offs[k]:offs[k + 1]], views[k]), range(len(views))))
            groups.append(buf)
        num_beams = beams_from_gen_kwargs(gen_kwargs)
        key = pass_key(max_new, sampling, num_beams)
        return {"prompts": prompts, "images_per_prompt": images_per_prompt, "views_per_image": [v.shape[0] for v in views], "sizes": sizes,
                "groups": groups, "max_new": max_new, "n": len(chunk), "sampling": sampling, "doc_ids": [int(x) for x in doc_ids], "key": key, "num_beams": num_beams}

    @staticmethod
    def _merge_preps(preps: list[dict]) -> dict:
        cat = lambda k: [x for p in preps for x in p[k]]  # noqa: E731
        return {"prompts": cat("prompts"), "images_per_prompt": cat("images_per_prompt"), "views_per_image": cat("views_per_image"),
                "sizes": cat("sizes"), "groups": cat("groups"), "max_new": preps[0]["max_new"], "n": sum(p["n"] for p in preps),
                "sampling": preps[0]["sampling"], "doc_ids": cat("doc_ids"), "num_beams": preps[0].get("num_beams", 1)}

    def _launch_chunk(self, prep: dict, eos_token_id: int, pad: int, carry: dict | None = None):
        """GPU stage of one pass (everything enqueued, nothing waits): H2D of the staged views + owc_clip_patchify_u8 + CLIP tower +
        projector, the packed feature rows of every `<image>` token, prefill + decode; ids come back through a pinned buffer."""
        eng = self._model
        feats, rows_per_prompt = None, [np.zeros(0, np.int64)] * prep["n"]
        if prep["groups"]:
            devs = self._h2d_groups(prep["groups"])          # pinned stacks: on the copy stream, beside the previous pass's kernels
            u8 = devs[0] if len(devs) == 1 else torch.cat(devs)
            feats = eng.encode_views(eng.patchify(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD))
            rows = eng.feature_rows(prep["views_per_image"], prep["sizes"])
            rows_per_prompt, cur = [], 0
            for k in prep["images_per_prompt"]:
                mine = rows[cur:cur + k]
                cur += k
                rows_per_prompt.append(np.concatenate(mine) if mine else np.zeros(0, np.int64))
        smp = None if prep.get("sampling") is None else {**prep["sampling"], "stream_ids": prep["doc_ids"]}   # one stream per document
        # (pad = EOS, as the reference passes pad_token_id=self.eot_token_id, :365-376)
        pad = eos_token_id if eos_token_id is not None and eos_token_id >= 0 else 0
        if prep.get("num_beams", 1) > 1:
            out = eng.generate_beam(prep["prompts"], feats, [[] for _ in prep["prompts"]], prep["max_new"], prep["num_beams"],
                                    eos_token_id=eos_token_id, pad_token_id=pad, img_rows=rows_per_prompt)
        else:
            out = eng.generate_from_features(prep["prompts"], feats, rows_per_prompt, prep["max_new"], eos_token_id=eos_token_id,
                                             pad_token_id=pad, sampling=smp, carry=carry)
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev


def dims_from_hf_config(cfg: dict) -> LlavaDims:
    """config.json of a llava-hf checkpoint (model_type `llava` or `llava_next`) -> engine dims."""
    if cfg.get("model_type", "llava") not in ("llava", "llava_next"):
        raise ValueError(f"unsupported LLaVA model_type {cfg.get('model_type')!r} (llava, llava_next)")
    t, v = cfg.get("text_config", {}), cfg.get("vision_config", {})
    if cfg.get("vision_feature_select_strategy", "default") != "default":
        raise ValueError("only vision_feature_select_strategy='default' is implemented")
    if v.get("patch_size", 14) != 14 or v.get("hidden_act", "quick_gelu") != "quick_gelu":
        raise ValueError("the CLIP tower must be a ViT-*/14 with quick_gelu")
    heads = t.get("num_attention_heads", 32)
    hidden = t.get("hidden_size", 4096)
    rope = t.get("rope_parameters") or {}
    pin = cfg.get("image_grid_pinpoints")
    if cfg.get("model_type") == "llava_next" and not pin:
        pin = NEXT_PINPOINTS
    return LlavaDims(
        v_layers=v.get("num_hidden_layers", 24), v_embed=v.get("hidden_size", 1024), v_heads=v.get("num_attention_heads", 16),
        v_mlp=v.get("intermediate_size", 4096), image_size=v.get("image_size", 336), feature_layer=cfg.get("vision_feature_layer", -2),
        v_ln_eps=v.get("layer_norm_eps", 1e-5), n_layers=t.get("num_hidden_layers", 32), d_model=hidden, n_q_heads=heads,
        n_kv_heads=t.get("num_key_value_heads", heads), head_dim=t.get("head_dim") or hidden // heads,
        d_ff=t.get("intermediate_size", 11008), vocab=t.get("vocab_size", 32064),
        tie_embeddings=bool(cfg.get("tie_word_embeddings", False)), rms_eps=t.get("rms_norm_eps", 1e-5),
        rope_theta=rope.get("rope_theta", t.get("rope_theta", 10000.0)), image_token_id=cfg.get("image_token_index", cfg.get("image_token_id", 32000)),
        max_positions=8192 if pin else 4096, grid_pinpoints=tuple(tuple(p) for p in pin) if pin else None)


class LlavaCheckpoint(LazyCheckpoint):
    """Accepts the 4.47 (`vision_tower.vision_model.`, `language_model.model.`, `multi_modal_projector.`,
    `image_newline`, `language_model.lm_head.`) and the 5.x (`model.vision_tower.`, `model.language_model.`) naming."""

    def __getitem__(self, name: str):
        legacy = (name.replace("model.vision_tower.", "vision_tower.vision_model.")
                  .replace("model.language_model.", "language_model.model.")
                  .replace("model.multi_modal_projector.", "multi_modal_projector.")
                  .replace("model.image_newline", "image_newline"))
        if name == "lm_head.weight":
            legacy = "language_model.lm_head.weight"
        mid = name.replace("model.vision_tower.", "model.vision_tower.vision_model.")
        for cand in (name, mid, legacy):
            if cand in self._files:
                return self._files[cand].get_tensor(cand)
        raise KeyError(name)


@register_model("llava-next-mistral-7b")
def llava_next_mistral_7b(**model_kwargs) -> LLaVA:
    return LLaVA(model_kwargs.pop("model_name_or_path", "llava-hf/llava-v1.6-mistral-7b-hf"), **model_kwargs)


@register_model("llava-next-vicuna-7b")
def llava_next_vicuna_7b(**model_kwargs) -> LLaVA:
    return LLaVA(model_kwargs.pop("model_name_or_path", "llava-hf/llava-v1.6-vicuna-7b-hf"), **model_kwargs)


@register_model("llava-1.5-13b")
def llava_15_13b(**model_kwargs) -> LLaVA:
    return LLaVA(model_kwargs.pop("model_name_or_path", "llava-hf/llava-1.5-13b-hf"), **model_kwargs)


@register_model("llava-1.5-7b")
def llava_15_7b(**model_kwargs) -> LLaVA:
    return LLaVA(model_kwargs.pop("model_name_or_path", "llava-hf/llava-1.5-7b-hf"), **model_kwargs)
