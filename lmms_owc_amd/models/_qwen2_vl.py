"""`Qwen2VL` model plug-in on the MI355X engine — drop-in for /root/reference/src/models/_qwen2_vl.py.

Same constructor kwargs (`:59-72`), same registry names (`:619-632`), same `generate_until` contract
(`list[TaskInstance] -> list[str]`, same length and order, `:143-348`), but requests are BATCHED: images of a
chunk go through the HIP vision tower together, prompts are prefilled packed and decoded as one batch.
`model_name_or_path="synthetic:<registry-name>"` builds seeded random weights + a byte tokenizer (no
checkpoint exists offline); a local HF checkpoint directory loads real weights, tokenizer and chat template.
"""

from __future__ import annotations

import json
from pathlib import Path

import numpy as np
import torch

from .. import _lib, ops, utils
from ..engine.qwen2vl import DIMS, Qwen2VLDims, Qwen2VLEngine, Qwen2VLWeights
from . import imageproc
from ._api import register_model
from ._base import Model, PassPipeline, beams_from_gen_kwargs, pass_key, sampling_from_gen_kwargs

__all__ = ["Qwen2VL"]

SYSTEM_PROMPT = "You are a helpful assistant."


class ByteTokenizer:
    """Stand-in tokenizer for synthetic runs: UTF-8 bytes + 3, specials above the byte range."""

    name_or_path = "synthetic-bytes"
    pad_token_id, eos_token_id = 0, 1
    im_start, im_end, vision_start, vision_end, image_pad = 260, 261, 262, 263, 264

    def encode(self, text: str) -> list[int]:
        return [b + 3 for b in text.encode("utf-8")]

    def decode(self, ids, skip_special_tokens: bool = True) -> str:
        if isinstance(ids, int):
            ids = [ids]
        return bytes(int(i) - 3 for i in ids if 3 <= int(i) < 259).decode("utf-8", errors="replace")

    def batch_decode(self, rows, skip_special_tokens: bool = True, **_) -> list[str]:
        return [self.decode(r) for r in rows]

    def chat_ids(self, question: str, n_image_tokens: list[int]) -> list[int]:
        return self.chat_ids_turns([("user", question, n_image_tokens)])

    def chat_ids_turns(self, turns: list[tuple], system: str = SYSTEM_PROMPT) -> list[int]:
        """ChatML ids of a conversation: turns = [(role, text, n_image_tokens per image)], generation prompt appended."""
        ids = [self.im_start] + self.encode("system\n" + system) + [self.im_end] + self.encode("\n")
        for role, text, n_image_tokens in turns:
            ids += [self.im_start] + self.encode(role + "\n")
            for n in n_image_tokens:
                ids += [self.vision_start] + [self.image_pad] * n + [self.vision_end]
            ids += self.encode(text) + [self.im_end] + self.encode("\n")
        return ids + [self.im_start] + self.encode("assistant\n")


class Qwen2VL(PassPipeline, Model):
    def __init__(self, model_name_or_path: str = "Qwen/Qwen2-VL-7B-Instruct", use_cache: bool = True,
                 use_flash_attention_2: bool | None = False, max_pixels: int = 1024 * 28 * 28,
                 min_pixels: int = 4 * 28 * 28, batch_size: int = 1, device_map: str = "auto",
                 dtype: str | torch.dtype = "bfloat16", load_in_8bit: bool = False, load_in_4bit: bool = False,
                 decoder_dtype: str = "bf16", engine_batch: int | str = "auto", **kwargs) -> None:
        # Two kwargs the reference does not have.  `decoder_dtype`: "fp8" runs the decoder projections on the e4m3fn path
        # (`--model_args decoder_dtype=fp8`).  `engine_batch`: how many requests one pass through the
        # engine takes.  The reference's scripts hard-code `--batch_size 1` (scripts/schedule_batch.sh:86, its wrapper supports
        # nothing else); here a sequence's tokens do not depend on what else is in the batch (tested bit for bit), so
        # `batch_size` is only a LOWER bound: "auto" (default) raises it to what fits in HBM, at most 2048 (10.7 images/s at
        # batch 1, 138 at 64, 240 at 2048 for the 7B model); an integer fixes it; 0 keeps exactly `batch_size`.
        # The reference's bitsandbytes switches stay rejected.
        if decoder_dtype not in ("bf16", "fp8"):
            raise ValueError("decoder_dtype must be 'bf16' or 'fp8'")
        if engine_batch != "auto" and (not isinstance(engine_batch, int) or engine_batch < 0):
            raise ValueError("engine_batch must be 'auto' or an integer >= 0")
        self._engine_batch_arg = engine_batch
        self._decoder_dtype = decoder_dtype
        self._model_name_or_path = model_name_or_path
        self._use_cache = use_cache                    # the HIP decoder always uses its KV cache
        self._use_flash_attention_2 = use_flash_attention_2  # accepted; attention is always the fused HIP kernel
        self._max_pixels = max_pixels
        self._min_pixels = min_pixels
        super().__init__(batch_size=batch_size, device_map=device_map, dtype=dtype, load_in_8bit=load_in_8bit,
                         load_in_4bit=load_in_4bit, distributed_types=["FSDP", "MULTI_GPU"], **kwargs)

    @classmethod
    def from_engine(cls, engine: Qwen2VLEngine, tokenizer, batch_size: int = 1, eos_token_id: int | None = None) -> "Qwen2VL":
        """A plug-in around an engine that already holds its weights (bench.py's PIL leg, tests): no second weight set.
        `eos_token_id=-1` disables EOS stopping (forced-length generation for benchmarks)."""
        self = cls.__new__(cls)
        self._engine_batch_arg = 0
        self._model_name_or_path = "engine"
        self._decoder_dtype = engine.d.decoder_dtype
        self._max_pixels, self._min_pixels = 1024 * 28 * 28, 4 * 28 * 28
        self._device = engine.device
        self._rank, self._world_size = 0, 1
        self.batch_size_per_gpu = int(batch_size)
        from ._base import CacheHook

        self.cache_hook = CacheHook(None)
        self.chat_template = None
        self.apply_chat_template = False
        self.task_dict = {}
        if eos_token_id is not None:
            tokenizer.eos_token_id = eos_token_id
        self._tokenizer = self._processor = tokenizer
        if hasattr(tokenizer, "image_pad"):
            tokenizer.image_pad = engine.d.image_token_id   # the engine scatters image rows at ITS placeholder id
        self._dims = engine.d
        self._model = engine
        self._start_workers()
        return self

    def engine_batch(self, max_new_tokens: int = 64) -> int:
        """Requests per engine pass: `batch_size`, raised (engine_batch="auto") to what the free HBM holds - KV cache for the
        largest image the processor admits (max_pixels / 784 image tokens + prompt + generation), its pixel values and merged
        embeddings per request; a quarter of the free memory, because two chunks are in flight (chunk k+1 is enqueued while
        chunk k runs) and the per-launch-group workspaces need room - and at most 2048.  The bound is for the WORST image
        size the processor admits; runs on small images can pass a larger `engine_batch` explicitly."""
        arg = getattr(self, "_engine_batch_arg", 0)
        if arg == 0:
            return self.batch_size
        if arg != "auto":
            return max(self.batch_size, int(arg))
        d = self._dims
        tokens = self._max_pixels // 784 + 64 + int(max_new_tokens)
        per_req = (d.n_layers * 2 * d.n_kv_heads * d.head_dim * 2) * tokens + (self._max_pixels // 196) * 1176 * 2 \
            + (self._max_pixels // 784) * d.d_model * 2
        # computed ONCE per max_new_tokens: after the first task torch's caching allocator holds the KV cache and workspaces,
        # `mem_get_info` would report them as used and later tasks would get a smaller batch, new chunk sizes and new allocations.
        # What torch has reserved but not allocated is reusable by the next pass, so it counts as free.
        cache = self.__dict__.setdefault("_auto_batch", {})
        if max_new_tokens not in cache:
            free, _ = torch.cuda.mem_get_info(self._device)
            free += torch.cuda.memory_reserved(self._device) - torch.cuda.memory_allocated(self._device)
            free += getattr(self._model, "held_bytes", lambda: 0)()   # the engine's own K / V pair + workspace are reused by the next pass
            cache[max_new_tokens] = max(self.batch_size, min(2048, int(0.25 * free / per_req)))
        return cache[max_new_tokens]

    # ------------------------------------------------------------------ loading
    def load_model(self) -> None:
        name = self._model_name_or_path
        if name.startswith("synthetic:"):
            key = name.split(":", 1)[1]
            dims = DIMS[key]
            tok = ByteTokenizer()
            dims = Qwen2VLDims(**{**dims.__dict__, "image_token_id": tok.image_pad, "decoder_dtype": self._decoder_dtype})
            weights = Qwen2VLWeights.random(dims, self._device, seed=1234)
            self._tokenizer = tok
        else:
            path = Path(name)
            if not path.is_dir():
                from huggingface_hub import snapshot_download

                path = Path(snapshot_download(name))
            dims = dims_from_hf_config(json.loads((path / "config.json").read_text()))
            dims = Qwen2VLDims(**{**dims.__dict__, "decoder_dtype": self._decoder_dtype})
            weights = Qwen2VLWeights.from_state_dict(dims, LazyCheckpoint(path), self._device)
            from transformers import AutoTokenizer

            self._tokenizer = AutoTokenizer.from_pretrained(str(path))
            self.chat_template = getattr(self._tokenizer, "chat_template", None)
            # HF merges generation_config.json into every generate() call for the fields the call does not pass.  The reference
            # passes do_sample / temperature / top_p / num_beams / max_new_tokens / eos / pad (src/models/_qwen2_vl.py:319-329), so
            # what the file still decides: `top_k` (when a request samples) and `repetition_penalty` - which HF applies to GREEDY
            # decoding as well (RepetitionPenaltyLogitsProcessor is added whenever the value is not 1.0; the Qwen2-VL / Qwen2.5-VL
            # instruct checkpoints ship 1.05)
            gc = path / "generation_config.json"
            if gc.exists():
                gcfg = json.loads(gc.read_text())
                self._default_top_k = int(gcfg.get("top_k", 50) or 0)
                self._repetition_penalty = float(gcfg.get("repetition_penalty") or 1.0)
        self._dims = dims
        self._start_workers()
        self._model = Qwen2VLEngine(weights)
        self._model.repetition_penalty = float(getattr(self, "_repetition_penalty", 1.0))
        self._processor = self._tokenizer

    def loglikelihood(self, requests: list) -> list[tuple[float, bool]]:
        raise NotImplementedError("Loglikelihood is not implemented for Qwen2_VL")  # as the reference (:141)

    def generate_until_multi_round(self, requests: list) -> list[tuple]:
        """Multi-round dialogue (the `*_llamav_o1` tasks; reference :350-616).  Per document and round r >= 1 the task's
        `doc_to_text(doc, round_idx=r, previous_round_results=, last_round_info=)` is asked for the next user turn and returns
        `(visuals, context, terminal, round_results, last_round_info)`; the reference (batch size 1) hands it
        `last_round_info = {"messages": [conversation]}` - the running HF-style message list - and continues FROM THE MESSAGES THE
        TASK RETURNS (`:479-480`; a task that returns no info restarts the conversation at the system prompt, and the per-round
        results continue from the list the task returns, `:455-461`).  Restated here per document of a batch: the same protocol,
        the same message structure (an image entry holds the PIL image where the reference stores a base64 JPEG data URL), the
        user turn appended, greedy generation, the answer cut at the `until` terms and appended as the assistant turn; the
        result per request is the tuple of per-round answers.  Batched over documents; an image's embedding is computed in the
        round that first shows it and reused by the later rounds (the reference re-encodes the conversation's images every
        round); a later round's visuals - or images a task puts into the messages it hands back - are encoded when they appear."""
        res: list[tuple] = []

        def _collate(x):
            return -len(self._tokenizer.encode(x[0])), x[0]

        reordered = utils.Collator([reg.args for reg in requests], _collate, grouping=True)
        for chunk in reordered.get_batched(n=self.batch_size, batch_fn=None):
            contexts, all_gen_kwargs, doc_to_visual, doc_to_text, doc_ids, tasks, splits = zip(*chunk, strict=True)
            task, split = tasks[0], splits[0]
            gen_kwargs = dict(all_gen_kwargs[0])
            tok = self._tokenizer
            until = gen_kwargs.pop("until", [tok.decode([tok.eos_token_id])])
            until = [until] if isinstance(until, str) else until
            if not isinstance(until, list):
                raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str,list] but got {type(until)}")
            max_new = int(gen_kwargs.get("max_new_tokens", 128))
            sampling = sampling_from_gen_kwargs(gen_kwargs, getattr(self, "_default_top_k", 50))
            num_beams = beams_from_gen_kwargs(gen_kwargs)
            docs = [self.task_dict[task][split][did] for did in doc_ids]
            # round 0: the reference flattens the batch's visuals and hands message i the i-th entry (:409-413, :482-483) - at its
            # batch size of one that is the document's FIRST image; later rounds get the list the task returns, whole
            visuals_per_doc = [list(doc_to_visual[0](d))[:1] for d in docs]
            emb = None                                  # embeddings of every image seen so far in this batch, in arrival order
            image_slot = [{} for _ in docs]             # [doc] PIL object (identity) -> image index of its document
            grids_per_doc = [[] for _ in docs]          # [doc][image] -> (t, h, w) patch grid
            row_range, n_rows = [[] for _ in docs], 0   # [doc][image] -> (first, last + 1) row of `emb`

            def embed_new(per_doc: list[list]) -> None:
                """Prepare + encode the images not seen before (any round may bring new ones) and register their rows."""
                nonlocal emb, n_rows
                new = [(i, v) for i, vs in enumerate(per_doc) for v in vs if id(v) not in image_slot[i]]
                new = [(i, v) for k, (i, v) in enumerate(new) if all(id(v) != id(w) or i != j for j, w in new[:k])]
                if not new:
                    return
                def prep(v):
                    # a PIL image goes through the JPEG round trip the reference applies when it builds the message (:485-490); a task
                    # that hands back a message in the reference's own format - a base64 JPEG data URL - is decoded as it is
                    if isinstance(v, str):
                        import base64
                        from io import BytesIO

                        from PIL import Image

                        if not v.startswith("data:image") or ";base64," not in v:
                            raise ValueError("multi-round: an image entry of a message must be a PIL image or a base64 data URL")
                        return imageproc.prepare_image(Image.open(BytesIO(base64.b64decode(v.split(";base64,", 1)[1]))),
                                                       self._min_pixels, self._max_pixels, jpeg=False)
                    return imageproc.prepare_image(v, self._min_pixels, self._max_pixels)

                arrs = list(self._pool.map(prep, [v for _, v in new]))
                grids = [(1, a.shape[1] // 14, a.shape[2] // 14) for a in arrs]
                e = self._model.encode_images(self._pixel_values(arrs), grids)
                emb = e if emb is None or e is None else torch.cat([emb, e])
                for (i, v), g in zip(new, grids):
                    image_slot[i][id(v)] = len(grids_per_doc[i])
                    grids_per_doc[i].append(g)
                    row_range[i].append((n_rows, n_rows + g[1] * g[2] // 4))
                    n_rows += g[1] * g[2] // 4

            messages: list[list] = [[] for _ in docs]   # the running conversation of every document (HF message dicts)
            results: list[list[str]] = [[] for _ in docs]   # [doc] -> its per-round answers so far
            texts = [c.replace("<image>", "") for c in contexts]
            round_visuals = visuals_per_doc
            active, round_idx = list(range(len(docs))), 0
            pad = tok.pad_token_id if tok.pad_token_id is not None else 0
            while True:
                if round_idx:
                    # every document follows ITS OWN terminal signal (the reference runs one document per batch, :462): a finished
                    # document leaves the following rounds, its result is the list its task returned last
                    still = []
                    for i in active:
                        vis, text, terminal, rr, info = doc_to_text[0](docs[i], round_idx=round_idx, previous_round_results=list(results[i]),
                                                                       last_round_info={"messages": [messages[i]]})
                        # the per-round results continue from what the task RETURNS, as in the reference (:455-461): a task may
                        # rewrite or truncate earlier answers
                        results[i] = list(rr)
                        if terminal:
                            continue
                        texts[i] = text.replace("<image>", "")
                        round_visuals[i] = [] if vis is None else ([vis] if not isinstance(vis, (list, tuple)) else list(vis))
                        # the conversation continues from the messages the task hands back (:479-480)
                        messages[i] = list(info["messages"][0]) if info and "messages" in info else []
                        still.append(i)
                    active = still
                if not active:
                    break
                for i in active:
                    if not messages[i]:
                        messages[i] = [{"role": "system", "content": SYSTEM_PROMPT}]
                    content = [{"type": "image", "image": v} for v in round_visuals[i]] + [{"type": "text", "text": texts[i]}]
                    messages[i].append({"role": "user", "content": content})
                # the conversation's images, in order (the reference re-reads them from the messages every round, :541): an image
                # seen in an earlier round (identity) reuses its embedding rows, a new one - a later round's visual, or one a task put
                # into the messages it handed back - is encoded now.  A conversation the task restarted holds none.
                conv_imgs = [[c["image"] for m in messages[i] if isinstance(m["content"], list) for c in m["content"] if c.get("type") == "image"]
                             if i in active else [] for i in range(len(docs))]
                embed_new(conv_imgs)
                prompts, round_grids, round_rows = [], [], []
                for i in active:
                    ks = [image_slot[i][id(v)] for v in conv_imgs[i]]
                    round_grids.append([grids_per_doc[i][k] for k in ks])
                    round_rows.append(np.concatenate([np.arange(*row_range[i][k]) for k in ks]) if ks else np.zeros(0, np.int64))
                    prompts.append(self._messages_ids(messages[i], [g[1] * g[2] // 4 for g in round_grids[-1]]))
                # (sampling: one stream per document and ROUND - a round's draws must not repeat the previous round's)
                smp = None if sampling is None else {**sampling, "stream_ids": [int(doc_ids[i]) * 64 + round_idx for i in active]}
                if num_beams > 1:
                    out = self._model.generate_beam(prompts, emb, round_grids, max_new, num_beams, eos_token_id=tok.eos_token_id,
                                                    pad_token_id=pad, img_rows=round_rows).cpu().numpy()
                else:
                    out = self._model.generate(prompts, emb, round_grids, max_new, eos_token_id=tok.eos_token_id, pad_token_id=pad,
                                               img_rows=round_rows, sampling=smp).cpu().numpy()
                rows = []
                for r in out:
                    stop = np.flatnonzero(r == tok.eos_token_id)
                    rows.append(r[: stop[0]] if len(stop) else r)
                answers = tok.batch_decode(rows, skip_special_tokens=True, clean_up_tokenization_spaces=False)
                for i, ans in zip(active, answers):
                    for term in until:
                        if len(term) > 0:
                            ans = ans.split(term)[0]
                    messages[i].append({"role": "assistant", "content": [{"type": "text", "text": ans}]})
                    results[i].append(ans)
                round_idx += 1
            res.extend(tuple(r) for r in results)
            round_results = [list(r) for r in results]
            self.cache_hook.add_partial("generate_until_multi_round", (contexts[0], gen_kwargs), round_results)
        return reordered.get_original(res)

    def _messages_ids(self, messages: list[dict], n_image_tokens: list[int]) -> np.ndarray:
        """HF-style message list -> prompt ids with the generation prompt appended; image entry k expands to n_image_tokens[k]
        placeholders.  A leading system message is taken as is, otherwise the default one is supplied (Qwen2-VL chat template)."""
        tok = self._tokenizer
        if isinstance(tok, ByteTokenizer):
            it = iter(n_image_tokens)
            msgs = list(messages)
            system = SYSTEM_PROMPT
            if msgs and msgs[0]["role"] == "system":
                c = msgs.pop(0)["content"]
                system = c if isinstance(c, str) else "".join(x.get("text", "") for x in c)
            turns = []
            for m in msgs:
                c = m["content"]
                if isinstance(c, str):
                    turns.append((m["role"], c, []))
                else:
                    turns.append((m["role"], "".join(x["text"] for x in c if "text" in x), [next(it) for x in c if x.get("type") == "image"]))
            return np.asarray(tok.chat_ids_turns(turns, system=system), dtype=np.int32)
        plain = [{"role": m["role"], "content": m["content"] if isinstance(m["content"], str) else
                  [{"type": "image"} if c.get("type") == "image" else c for c in m["content"]]} for m in messages]
        ids = tok.encode(tok.apply_chat_template(plain, tokenize=False, add_generation_prompt=True))
        out, it = [], iter(n_image_tokens)
        for t in ids:
            out.extend([t] * next(it) if t == self._dims.image_token_id else [t])
        return np.asarray(out, dtype=np.int32)

    # ------------------------------------------------------------------ prompt building
    def _prompt_ids(self, context: str, n_image_tokens: list[int]) -> np.ndarray:
        tok = self._tokenizer
        if isinstance(tok, ByteTokenizer):
            return np.asarray(tok.chat_ids(context, n_image_tokens), dtype=np.int32)
        content = [{"type": "image"} for _ in n_image_tokens] + [{"type": "text", "text": context}]
        messages = [{"role": "system", "content": SYSTEM_PROMPT}, {"role": "user", "content": content}]
        text = tok.apply_chat_template(messages, tokenize=False, add_generation_prompt=True)
        ids = tok.encode(text)
        # expand every single <|image_pad|> placeholder to the image's token count (HF processor behaviour)
        out, it = [], iter(n_image_tokens)
        for t in ids:
            if t == self._dims.image_token_id:
                out.extend([t] * next(it))
            else:
                out.append(t)
        return np.asarray(out, dtype=np.int32)

    # ------------------------------------------------------------------ the hot loop
    def generate_until(self, requests: list) -> list[str]:
        """`list[TaskInstance] -> list[str]`, same length and order (reference :143-348), batched and double-buffered."""
        answers = self.decode_tokens(self._generate_rows(requests))
        for req, ans in zip(requests, answers):
            self.cache_hook.add_partial("generate_until", (req.args[0], req.args[1]), ans)
        return answers

    def generate_until_tokens(self, requests: list) -> tuple[np.ndarray, np.ndarray]:
        """The same generation as fixed-width token records for the engine's end-of-task RCCL gather:
        int32 [n, T] (ids up to EOS, zero-padded) and int32 [n] lengths, in request order; `decode_tokens` turns them
        into the strings `generate_until` would have returned (rank 0 does that for every rank's records)."""
        rows = self._generate_rows(requests)
        T = max([1] + [len(r) for r in rows] + [int(r.args[1].get("max_new_tokens", 128)) for r in requests])
        mat = np.zeros((len(rows), T), np.int32)
        for i, r in enumerate(rows):
            mat[i, : len(r)] = r
        return mat, np.array([len(r) for r in rows], np.int32)

    def decode_tokens(self, rows: list) -> list[str]:
        return self._tokenizer.batch_decode([np.asarray(r) for r in rows], skip_special_tokens=True,
                                            clean_up_tokenization_spaces=False)

    def _prepare_chunk(self, chunk) -> dict:
        """HOST stage of one chunk (runs on the preparation thread while the GPU works on the previous chunk): image fetch,
        JPEG round trip + two-stage bicubic resize (`imageproc.prepare_image`, fanned out over the worker pool: PIL releases
        the GIL), prompt token ids, and the chunk's uint8 images stacked into PINNED host memory for an asynchronous H2D."""
        contexts, all_gen_kwargs, doc_to_visual, doc_ids, tasks, splits = zip(*chunk, strict=True)
        task, split = tasks[0], splits[0]
        for g in all_gen_kwargs:   # the reference (batch size 1) pops it from EVERY request's own dict, which is what the
            g.pop("until", None)   # samples file later records under `arguments` (_engine.py:262-266, _tracker.py:318-322)
        gen_kwargs = dict(all_gen_kwargs[0])   # popped and unused in the reference's single-round mode (:211-219)
        max_new = int(gen_kwargs.get("max_new_tokens", 128))
        sampling = sampling_from_gen_kwargs(gen_kwargs, getattr(self, "_default_top_k", 50))
        docs = self.task_dict[task][split]

        def fetch(did):   # image fetch (file open + decode in the reference's tasks) AND preparation run on the pool workers
            # (a document's FIRST image only: the reference flattens the batch's visuals and hands message i the i-th entry, :196-200 /
            # :232-233 - at its batch size of one that is image 0; pinned on its own run, tests/test_wrapper_protocol.py)
            return [imageproc.prepare_image(v, self._min_pixels, self._max_pixels) for v in list(doc_to_visual[0](docs[did]))[:1]]

        arrs_per_doc = list(self._pool.map(fetch, doc_ids))
        images, grids_per_prompt, prompts = [], [], []
        cache: dict = {}   # a task asks every image the same question: one tokenisation per distinct (question, image-token counts)
        for ctx, arrs in zip(contexts, arrs_per_doc):
            grids = [(1, a.shape[1] // 14, a.shape[2] // 14) for a in arrs]
            images += arrs
            grids_per_prompt.append(grids)
            key = (ctx, tuple(g[1] * g[2] // 4 for g in grids))
            if key not in cache:
                cache[key] = self._prompt_ids(ctx.replace("<image>", ""), list(key[1]))
            prompts.append(cache[key])
        # same-size runs share one patchify launch; each run is staged in pinned memory (parallel copies: memcpy drops the GIL)
        # (ONE fan-out for the whole unit: real datasets have runs of 1-2 images - Food-101 sizes: 80 runs per 128 images - and a
        # pool round trip per run cost an eighth of the unit's preparation, tools/probe_host_prep_scaling.py)
        groups, copies, i = [], [], 0
        while i < len(images):
            j = i
            while j < len(images) and images[j].shape == images[i].shape:
                j += 1
            buf = self._pinned_take((j - i, *images[i].shape))
            dst = buf.numpy()
            copies += [(dst[k - i], images[k]) for k in range(i, j)]
            groups.append(buf)
            i = j
        list(self._pool.map(lambda c: np.copyto(*c), copies))
        num_beams = beams_from_gen_kwargs(gen_kwargs)
        key = pass_key(max_new, sampling, num_beams)
        return {"prompts": prompts, "grids": grids_per_prompt, "groups": groups, "max_new": max_new, "n": len(chunk),
                "sampling": sampling, "doc_ids": [int(d) for d in doc_ids], "key": key, "num_beams": num_beams}

    def _generate_rows(self, requests: list) -> list[np.ndarray]:
        """Token rows (cut at EOS) per request, in request order: the two-stage pass pipeline of `_base.PassPipeline`."""
        return self._run_passes(requests, default_max_new=128)

    @staticmethod
    def _merge_preps(preps: list[dict]) -> dict:
        return {"prompts": [x for p in preps for x in p["prompts"]], "grids": [x for p in preps for x in p["grids"]],
                "groups": [x for p in preps for x in p["groups"]], "max_new": preps[0]["max_new"], "n": sum(p["n"] for p in preps),
                "sampling": preps[0]["sampling"], "doc_ids": [x for p in preps for x in p["doc_ids"]], "num_beams": preps[0].get("num_beams", 1)}

    def _launch_chunk(self, prep: dict, eos_token_id: int, pad: int, carry: dict | None = None):
        """GPU stage of one prepared chunk: H2D + patchify + vision tower + prefill + decode are ENQUEUED (nothing waits), the ids
        come back through a pinned buffer; returns (host int32 [n, T], event that passes when the buffer is filled).
        (tools/soak_host_ranks.py replaces exactly this method by a timed stand-in to soak the host side of 8 ranks.)"""
        emb = None
        if prep["groups"]:
            emb = self._model.encode_images(self._pixel_values(prep["groups"]), [g for gs in prep["grids"] for g in gs])
        smp = None if prep.get("sampling") is None else {**prep["sampling"], "stream_ids": prep["doc_ids"]}   # one stream per document
        if prep.get("num_beams", 1) > 1:
            out = self._model.generate_beam(prep["prompts"], emb, prep["grids"], prep["max_new"], prep["num_beams"],
                                            eos_token_id=eos_token_id, pad_token_id=pad)
        else:
            out = self._model.generate(prep["prompts"], emb, prep["grids"], prep["max_new"], eos_token_id=eos_token_id, pad_token_id=pad,
                                       sampling=smp, carry=carry)
        host = torch.empty(out.shape, dtype=out.dtype, pin_memory=True)
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return host, ev

    def _pixel_values(self, images: list) -> torch.Tensor:
        """uint8 images (sides % 28 == 0) -> packed pixel_values rows on the GPU.  `images`: CHW numpy arrays (same-size
        neighbours share a launch) or already-stacked [n, 3, H, W] uint8 tensors in pinned memory (asynchronous H2D)."""
        groups = []
        if images and isinstance(images[0], torch.Tensor):
            groups = list(images)
        else:
            i = 0
            while i < len(images):
                j = i
                while j < len(images) and images[j].shape == images[i].shape:
                    j += 1
                groups.append(torch.from_numpy(np.stack(images[i:j])))
                i = j
        rows = sum(g.shape[0] * (g.shape[2] // 14) * (g.shape[3] // 14) for g in groups)
        pix = torch.empty((rows, 1176), dtype=torch.bfloat16, device=self._device)
        dev_groups = self._h2d_groups(groups)
        r0 = 0
        for g, dg in zip(groups, dev_groups):
            n = g.shape[0] * (g.shape[2] // 14) * (g.shape[3] // 14)
            ops.patchify_u8(dg, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD, out=pix[r0:r0 + n])
            r0 += n
        return pix


def dims_from_hf_config(cfg: dict) -> Qwen2VLDims:
    t = cfg.get("text_config", cfg)
    v = cfg["vision_config"]
    rope = t.get("rope_parameters") or t.get("rope_scaling") or {}
    v25 = {}
    if cfg.get("model_type") == "qwen2_5_vl" or "fullatt_block_indexes" in v:   # Qwen2.5-VL (reference :106-115)
        v = {**v, "embed_dim": v["hidden_size"], "mlp_ratio": 0}
        v25 = dict(v_variant=1, v_window=v.get("window_size", 112), v_fullatt=tuple(v.get("fullatt_block_indexes", (7, 15, 23, 31))))
        if v.get("hidden_act", "silu") != "silu":
            raise ValueError("the Qwen2.5-VL vision MLP is implemented for hidden_act = silu")
    return Qwen2VLDims(
        **v25,
        v_depth=v["depth"], v_embed=v["embed_dim"], v_heads=v["num_heads"],
        v_mlp=v["intermediate_size"] if v25 else int(v["embed_dim"] * v.get("mlp_ratio", 4)),
        patch_k=v.get("in_channels", v.get("in_chans", 3)) * v.get("temporal_patch_size", 2) * v.get("patch_size", 14) ** 2,
        merge=v.get("spatial_merge_size", 2), n_layers=t["num_hidden_layers"], d_model=t["hidden_size"],
        n_q_heads=t["num_attention_heads"], n_kv_heads=t["num_key_value_heads"],
        head_dim=t["hidden_size"] // t["num_attention_heads"], d_ff=t["intermediate_size"], vocab=t["vocab_size"],
        tie_embeddings=bool(cfg.get("tie_word_embeddings", t.get("tie_word_embeddings", False))),
        rms_eps=t.get("rms_norm_eps", 1e-6), rope_theta=rope.get("rope_theta", t.get("rope_theta", 1e6)),
        mrope_section=tuple(rope.get("mrope_section", (16, 24, 24))), image_token_id=cfg.get("image_token_id", 151655))


class LazyCheckpoint:
    """name -> tensor over safetensors shards; accepts both the 4.47 (`visual.`, `model.layers.`) and the
    5.x (`model.visual.`, `model.language_model.layers.`) parameter naming."""

    def __init__(self, path: Path) -> None:
        from safetensors import safe_open

        self._files = {}
        for f in sorted(path.glob("*.safetensors")):
            h = safe_open(str(f), framework="pt", device="cpu")
            for k in h.keys():
                self._files[k] = h
        if not self._files:
            raise FileNotFoundError(f"no safetensors shards under {path}")

    def __getitem__(self, name: str):
        for cand in (name, name.replace("model.visual.", "visual."), name.replace("model.language_model.", "model.")):
            if cand in self._files:
                return self._files[cand].get_tensor(cand)
        raise KeyError(name)


@register_model("qwen2-vl-7b")
def qwen2_vl_7b(**model_kwargs) -> Qwen2VL:
    return Qwen2VL(model_kwargs.pop("model_name_or_path", "Qwen/Qwen2-VL-7B-Instruct"), **model_kwargs)


@register_model("qwen2-vl-2b")
def qwen2_vl_2b(**model_kwargs) -> Qwen2VL:
    return Qwen2VL(model_kwargs.pop("model_name_or_path", "Qwen/Qwen2-VL-2B-Instruct"), **model_kwargs)


@register_model("qwen2.5-vl-7b")
def qwen25_vl_7b(**model_kwargs) -> Qwen2VL:
    return Qwen2VL(model_kwargs.pop("model_name_or_path", "Qwen/Qwen2.5-VL-7B-Instruct"), **model_kwargs)


@register_model("qwen2.5-vl-3b")
def qwen25_vl_3b(**model_kwargs) -> Qwen2VL:
    return Qwen2VL(model_kwargs.pop("model_name_or_path", "Qwen/Qwen2.5-VL-3B-Instruct"), **model_kwargs)


@register_model("qwen2-vl-72b")
def qwen2_vl_72b(**model_kwargs) -> Qwen2VL:
    return Qwen2VL(model_kwargs.pop("model_name_or_path", "Qwen/Qwen2-VL-72B-Instruct"), **model_kwargs)
