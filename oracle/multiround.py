"""Restatement of the reference's multi-round generation PROTOCOL (TEST INFRASTRUCTURE ONLY; never imported by the product).

`reference_multi_round` follows /root/reference/src/models/_qwen2_vl.py:425-612 for ONE request (the reference's wrapper only
supports batch size 1, `_base.py:103-104`): the round loop, what is handed to and taken back from the task's `doc_to_text`
(`previous_round_results`, `last_round_info = {"messages": [...]}`, the five-tuple), the HF-style message list with the image
entry in the user turn that carries it, the `until` cut and the assistant turn.  The model call itself is a callback
(`generate_text(messages) -> str`), so the protocol is checked independently of the arithmetic (which has its own oracles).

`reference_multi_round_llava` does the same for /root/reference/src/models/_llava_hf.py:440-584 (independent single-turn rounds).

`render_qwen2vl_chat` renders a message list with the chat template PUBLISHED with the checkpoints the reference loads
(Qwen/Qwen2-VL-*-Instruct `chat_template.json`; third-party data, restated here because no checkpoint file exists offline) - what
`self.processor.apply_chat_template(msg, tokenize=False, add_generation_prompt=True)` returns at `_qwen2_vl.py:535-540`.
parity: the template text is restated from the published checkpoint, not read from it ("parity unpinned" for that string);
the PROTOCOL is pinned on the reference's own run: tools/gen_golden_wrappers.py executes the reference's two
`generate_until_multi_round` methods in the build container on a stand-in checkpoint (answer = a function of the rendered
prompt) and records every generate call (rendered prompt, image count, generation arguments) and the returned tuples in
tests/golden/wrapper_protocol.json; tests/test_wrapper_protocol.py drives the two functions below with the same stand-in
and requires the same calls and the same results.
"""

from __future__ import annotations

QWEN2VL_CHAT_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% set video_count = namespace(value=0) %}"
    "{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n"
    "{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n"
    "{% else %}{% for content in message['content'] %}"
    "{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}"
    "{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}<|vision_start|><|image_pad|><|vision_end|>"
    "{% elif content['type'] == 'video' or 'video' in content %}"
    "{% set video_count.value = video_count.value + 1 %}"
    "{% if add_vision_id %}Video {{ video_count.value }}: {% endif %}<|vision_start|><|video_pad|><|vision_end|>"
    "{% elif 'text' in content %}{{ content['text'] }}{% endif %}"
    "{% endfor %}<|im_end|>\n{% endif %}{% endfor %}"
    "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}"
)


def render_qwen2vl_chat(messages: list[dict], add_generation_prompt: bool = True) -> str:
    from jinja2.sandbox import ImmutableSandboxedEnvironment

    env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
    return env.from_string(QWEN2VL_CHAT_TEMPLATE).render(messages=messages, add_generation_prompt=add_generation_prompt,
                                                         add_vision_id=False)


def reference_multi_round(doc: dict, context: str, doc_to_visual, doc_to_text, gen_kwargs: dict, generate_text, eot_text: str):
    """One request of `generate_until_multi_round` (reference :425-612, batch size 1).  Returns the tuple of per-round answers."""
    from PIL import Image

    batched_visuals = list(doc_to_visual(doc))                                    # :409-413 (flattened over a batch of one)
    gen_kwargs = dict(gen_kwargs)
    until = [eot_text]                                                            # :420
    if "until" in gen_kwargs:                                                     # :423-431
        until = gen_kwargs.pop("until")
        if isinstance(until, str):
            until = [until]
        elif not isinstance(until, list):
            raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str,list] but got {type(until)}")
    round_idx = 0
    batched_round_results, batched_round_info = [], []                            # :434
    batched_contexts = [context]
    while True:
        last_round_info = None
        if round_idx != 0:                                                        # :439-463
            previous_round_results = [round_results[0] for round_results in batched_round_results]
            if len(batched_round_info) > 0:
                last_round_info = batched_round_info[-1][0]
            result = doc_to_text(doc, round_idx=round_idx, previous_round_results=previous_round_results,
                                 last_round_info=last_round_info)
            visuals, ctx, terminal, rr, info = result
            batched_visuals, batched_contexts = (visuals,), [ctx]
            batched_round_results = list(zip(*[rr], strict=True))                 # [round][doc] again, as tuples
            batched_round_results = [list(x) for x in batched_round_results]
            last_round_info = info
            if terminal:
                break
        ctx = batched_contexts[0]
        if "<image>" in ctx:                                                      # :468-475
            ctx = ctx.replace("<image>", "")
        message = [{"role": "system", "content": "You are a helpful assistant."}]  # :477
        if last_round_info and "messages" in last_round_info:                     # :479-480
            message = last_round_info["messages"][0]
        visual = batched_visuals[0] if len(batched_visuals) > 0 else None         # :482-483
        if isinstance(visual, Image.Image):                                       # :484-500 (the image travels as a data URL there)
            message.append({"role": "user", "content": [{"type": "image", "image": visual}, {"type": "text", "text": ctx}]})
        elif isinstance(visual, (list, tuple)) and visual and all(isinstance(v, Image.Image) for v in visual):   # :501-521
            message.append({"role": "user", "content": [{"type": "image", "image": v} for v in visual] + [{"type": "text", "text": ctx}]})
        else:                                                                     # :522-529
            message.append({"role": "user", "content": [{"type": "text", "text": ctx}]})
        answer = generate_text(message)                                           # :535-590: template, processor, generate, decode
        for term in until:                                                        # :591-595
            if len(term) > 0:
                answer = answer.split(term)[0]
        message.append({"role": "assistant", "content": [{"type": "text", "text": answer}]})   # :596-601
        round_idx += 1
        batched_round_results.append([answer])                                    # :602
        batched_round_info.append([dict(messages=[message])])                     # :603
    return tuple(r[0] for r in batched_round_results)                             # :605


def reference_multi_round_llava(doc: dict, context: str, doc_to_visual, doc_to_text, gen_kwargs: dict, generate_text, image_token: str = "<image>"):
    """One request of LLaVA's `generate_until_multi_round` (reference src/models/_llava_hf.py:440-584, batch size 1).  Unlike the Qwen2-VL
    wrapper it keeps NO conversation: every round is an independent single-turn prompt from the round's visuals and context
    (`batched_round_info` is never appended to, so `last_round_info` is always None); `until` is popped and never applied.
    `generate_text(context_with_image_tokens, visuals) -> str` stands for template + processor + generate + decode (:522-574)."""
    visuals = [v for v in doc_to_visual(doc)]                                      # :451-452 (flattened over a batch of one)
    gen_kwargs = dict(gen_kwargs)
    if "until" in gen_kwargs:                                                     # :461-470 (validated, then unused)
        until = gen_kwargs.pop("until")
        if not isinstance(until, (str, list)):
            raise ValueError(f"Expected `gen_kwargs['until']` to be of type Union[str,list] but got {type(until)}")
    round_idx = 0
    batched_round_results: list = []
    contexts = (context,)
    while True:
        if round_idx != 0:                                                        # :478-497
            previous_round_results = [round_results[0] for round_results in batched_round_results]
            vis, ctx, terminal, rr, _info = doc_to_text(doc, round_idx=round_idx, previous_round_results=previous_round_results,
                                                        last_round_info=None)
            visuals, contexts = (vis,), (ctx,)
            batched_round_results = [list(x) for x in zip(*[rr], strict=True)]
            if terminal:
                break
        ctx = contexts[0]                                                         # :499
        if isinstance(visuals, tuple):                                            # :501-502 (raises TypeError for a `None` visual, as there)
            visuals = list(*visuals)
        if image_token not in ctx:                                                # :506-509
            ctx = f"{' '.join([image_token] * len(visuals))}\n{ctx}"
        answer = generate_text(ctx, visuals)                                      # :511-574
        round_idx += 1
        batched_round_results.append([answer])                                    # :577
    return tuple(r[0] for r in batched_round_results)                             # :579
