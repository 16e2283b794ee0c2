"""Builds tiny on-disk HF-style checkpoints (config.json + safetensors + a fast tokenizer) so the real
loading paths (LazyCheckpoint, dims_from_hf_config, AutoTokenizer, chat template) are exercised offline."""
import json
from pathlib import Path

import numpy as np
import torch

from tests import recipes

CHAT_TEMPLATE = (
    "{% for message in messages %}<|im_start|>{{ message['role'] }}\n"
    "{% if message['content'] is string %}{{ message['content'] }}{% else %}{% for c in message['content'] %}"
    "{% if c['type'] == 'image' %}<|vision_start|><|image_pad|><|vision_end|>{% else %}{{ c['text'] }}{% endif %}"
    "{% endfor %}{% endif %}<|im_end|>\n{% endfor %}{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}"
)


def write_tokenizer(path: Path, vocab_size: int, specials: dict[str, int]) -> None:
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    words = ["<unk>", "what", "type", "of", "object", "is", "in", "this", "photo", "?", "you", "are", "a", "helpful",
             "assistant", ".", "system", "user", "sea", "lion", "dog", "class", "0", "1", "2", "_"]
    vocab = {w: i for i, w in enumerate(words)}
    vocab.update(specials)
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", eos_token="<|im_end|>", pad_token="<|endoftext|>",
                                   additional_special_tokens=[s for s in specials if s not in ("<|im_end|>", "<|endoftext|>")])
    fast.chat_template = CHAT_TEMPLATE
    fast.save_pretrained(str(path))


def write_qwen2vl_checkpoint(path: Path, legacy_names: bool = False) -> dict:
    """Tiny Qwen2-VL checkpoint; `legacy_names` uses the transformers-4.47 parameter prefixes (`visual.`, `model.`)."""
    from safetensors.torch import save_file

    path.mkdir(parents=True, exist_ok=True)
    cfg = recipes.tiny_cfg()
    w = recipes.qwen2vl_weights(cfg, 1234)
    specials = {"<|endoftext|>": 490, "<|im_start|>": 491, "<|im_end|>": 492, "<|vision_start|>": 493, "<|vision_end|>": 494,
                "<|image_pad|>": cfg.image_token_id}
    sd = {}
    for k, v in w.items():
        name = k
        if legacy_names:
            name = name.replace("model.visual.", "visual.").replace("model.language_model.", "model.")
        sd[name] = torch.from_numpy(v).to(torch.bfloat16).contiguous()
    keys = sorted(sd)
    save_file({k: sd[k] for k in keys[: len(keys) // 2]}, str(path / "model-00001-of-00002.safetensors"))
    save_file({k: sd[k] for k in keys[len(keys) // 2:]}, str(path / "model-00002-of-00002.safetensors"))
    t, v = cfg.text, cfg.vision
    hf = {"model_type": "qwen2_vl", "image_token_id": cfg.image_token_id, "tie_word_embeddings": False,
          "vision_config": {"depth": v.depth, "embed_dim": v.embed_dim, "num_heads": v.num_heads, "hidden_size": v.hidden_size,
                            "mlp_ratio": 4, "patch_size": 14, "spatial_merge_size": 2, "temporal_patch_size": 2, "in_chans": 3},
          "hidden_size": t.hidden_size, "num_hidden_layers": t.num_hidden_layers, "num_attention_heads": t.num_attention_heads,
          "num_key_value_heads": t.num_key_value_heads, "intermediate_size": t.intermediate_size, "vocab_size": t.vocab_size,
          "rms_norm_eps": t.rms_norm_eps, "rope_theta": t.rope_theta,
          "rope_scaling": {"type": "mrope", "mrope_section": list(t.mrope_section)}}
    (path / "config.json").write_text(json.dumps(hf))
    write_tokenizer(path, t.vocab_size, specials)
    return {"cfg": cfg, "weights": w, "specials": specials}


def write_bert_checkpoint(path: Path) -> dict:
    from safetensors.torch import save_file

    path.mkdir(parents=True, exist_ok=True)
    c = dict(recipes.bert_cfg("minilm"))
    c["vocab_size"] = 600
    w = recipes.bert_weights(c, 99)
    save_file({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in w.items()}, str(path / "model.safetensors"))
    (path / "config.json").write_text(json.dumps({"model_type": "bert", **c}))
    write_tokenizer(path, 600, {"<|endoftext|>": 590, "<|im_end|>": 591})
    return {"cfg": c, "weights": w}


def write_llava_checkpoint(path: Path, legacy_names: bool = False, next_: bool = False) -> dict:
    """Tiny llava-hf style checkpoint (LLaVA-1.5, or LLaVA-NeXT with `next_`); `legacy_names` uses the
    transformers-4.47 prefixes (`vision_tower.vision_model.`, `language_model.model.`, `language_model.lm_head.`)."""
    from safetensors.torch import save_file
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast

    path.mkdir(parents=True, exist_ok=True)
    cfg = recipes.tiny_llava_next_cfg() if next_ else recipes.tiny_llava_cfg()
    w = recipes.llava_weights(cfg, 1234)
    sd = {}
    for k, v in w.items():
        name = k
        if legacy_names:
            name = (name.replace("model.vision_tower.", "vision_tower.vision_model.").replace("model.language_model.", "language_model.model.")
                    .replace("model.multi_modal_projector.", "multi_modal_projector.").replace("model.image_newline", "image_newline"))
            if name == "lm_head.weight":
                name = "language_model.lm_head.weight"
        sd[name] = torch.from_numpy(v).to(torch.bfloat16).contiguous()
    save_file(sd, str(path / "model.safetensors"))
    t, v = cfg.text, cfg.vision
    hf = {"model_type": "llava_next" if next_ else "llava", "image_token_index": cfg.image_token_id, "vision_feature_layer": -2,
          "vision_feature_select_strategy": "default", "tie_word_embeddings": False,
          "vision_config": {"model_type": "clip_vision_model", "hidden_size": v.hidden_size, "intermediate_size": v.intermediate_size,
                            "num_hidden_layers": v.num_hidden_layers, "num_attention_heads": v.num_attention_heads,
                            "image_size": v.image_size, "patch_size": 14},
          "text_config": {"model_type": "llama", "hidden_size": t.hidden_size, "num_hidden_layers": t.num_hidden_layers,
                          "num_attention_heads": t.num_attention_heads, "num_key_value_heads": t.num_key_value_heads,
                          "intermediate_size": t.intermediate_size, "vocab_size": t.vocab_size, "rms_norm_eps": t.rms_norm_eps,
                          "rope_theta": t.rope_theta}}
    if next_:
        hf["image_grid_pinpoints"] = [list(p) for p in cfg.image_grid_pinpoints]
    (path / "config.json").write_text(json.dumps(hf))
    words = ["<unk>", "<s>", "</s>", "what", "type", "of", "object", "is", "in", "this", "image", "photo", "?", "a", "chat", "between",
             "curious", "user", "and", "an", "artificial", "intelligence", "assistant", ".", "the", "gives", "helpful", ",", "detailed",
             "polite", "answers", "to", "'", "s", "questions", "USER", "ASSISTANT", ":", "sea", "lion", "class", "0", "1", "2", "_"]
    vocab = {x: i for i, x in enumerate(words)}
    vocab["<image>"] = cfg.image_token_id
    tok = Tokenizer(models.WordLevel(vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Sequence([pre_tokenizers.WhitespaceSplit(), pre_tokenizers.Punctuation()])
    from tokenizers.processors import TemplateProcessing

    tok.post_processor = TemplateProcessing(single="<s> $A", special_tokens=[("<s>", 1)])
    fast = PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>", pad_token="<unk>",
                                   additional_special_tokens=["<image>"])
    fast.save_pretrained(str(path))   # no chat template: the wrapper falls back to the Vicuna prompt
    return {"cfg": cfg, "weights": w, "image_token": cfg.image_token_id}
