"""numpy restatement of the LLaVA (CLIP ViT + MLP projector + Llama-family decoder) greedy path
(TEST INFRASTRUCTURE ONLY).  The reference reaches this arithmetic through HF transformers
(`src/models/_llava_hf.py:365-376` calls `LlavaForConditionalGeneration.generate`; pinned 4.47.0, not
vendored).  Restated from transformers/models/clip/modeling_clip.py (CLIPVisionEmbeddings, CLIPEncoderLayer,
CLIPAttention, CLIPMLP) and transformers/models/llava/modeling_llava.py (get_image_features :144-189,
LlavaMultiModalProjector :87-106); the decoder is the Llama layer stack (same maths as `qwen2vl_np.llm_forward`
without q/k/v biases and with 1-D RoPE = three identical M-RoPE streams).  Pinned by tests/test_oracle_llava.py.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import np_ops as ops
from . import qwen2vl_np as Q
from .np_ops import maybe_bf16


@dataclass
class ClipCfg:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 336
    patch_size: int = 14
    layer_norm_eps: float = 1e-5


@dataclass
class LlavaCfg:
    vision: ClipCfg = field(default_factory=ClipCfg)
    text: Q.TextCfg = field(default_factory=lambda: Q.TextCfg(hidden_size=4096, num_hidden_layers=32, num_attention_heads=32,
                                                              num_key_value_heads=32, intermediate_size=11008, vocab_size=32064,
                                                              rms_norm_eps=1e-5, rope_theta=10000.0, tie_word_embeddings=False))
    image_token_id: int = 32000
    vision_feature_layer: int = -2
    image_grid_pinpoints: tuple | None = None  # LLaVA-NeXT anyres; None = LLaVA-1.5


VT = "model.vision_tower."


def clip_features(w: dict, cfg: LlavaCfg, pixel_values: np.ndarray, *, bf16=False) -> np.ndarray:
    """CLIPVisionModel hidden_states[vision_feature_layer] without the CLS row ("default" select strategy).
    pixel_values [n, 3, S, S] (already rescaled/normalised) -> [n, (S/14)^2, hidden]."""
    vc = cfg.vision
    E, H = vc.hidden_size, vc.num_attention_heads
    hd = E // H
    n = pixel_values.shape[0]
    g = vc.image_size // vc.patch_size
    x = maybe_bf16(pixel_values, bf16)
    patches = x.reshape(n, 3, g, vc.patch_size, g, vc.patch_size).transpose(0, 2, 4, 1, 3, 5).reshape(n * g * g, -1)
    pe = ops.linear(patches, w[VT + "embeddings.patch_embedding.weight"].reshape(E, -1), bf16=bf16).reshape(n, g * g, E)
    cls = np.broadcast_to(maybe_bf16(w[VT + "embeddings.class_embedding"], bf16)[None, None], (n, 1, E))
    h = maybe_bf16(np.concatenate([cls, pe], 1) + w[VT + "embeddings.position_embedding.weight"][None], bf16)
    h = ops.layer_norm(h, w[VT + "pre_layrnorm.weight"], w[VT + "pre_layrnorm.bias"], vc.layer_norm_eps, bf16=bf16)
    n_run = vc.num_hidden_layers + 1 + cfg.vision_feature_layer if cfg.vision_feature_layer < 0 else cfg.vision_feature_layer
    for i in range(n_run):
        p = f"{VT}encoder.layers.{i}."
        y = ops.layer_norm(h, w[p + "layer_norm1.weight"], w[p + "layer_norm1.bias"], vc.layer_norm_eps, bf16=bf16)
        T = y.shape[1]
        q = ops.linear(y, w[p + "self_attn.q_proj.weight"], w[p + "self_attn.q_proj.bias"], bf16=bf16).reshape(n, T, H, hd)
        k = ops.linear(y, w[p + "self_attn.k_proj.weight"], w[p + "self_attn.k_proj.bias"], bf16=bf16).reshape(n, T, H, hd)
        v = ops.linear(y, w[p + "self_attn.v_proj.weight"], w[p + "self_attn.v_proj.bias"], bf16=bf16).reshape(n, T, H, hd)
        a = np.stack([Q._attn(q[b].transpose(1, 0, 2), k[b].transpose(1, 0, 2), v[b].transpose(1, 0, 2), hd ** -0.5, False, bf16)
                      .transpose(1, 0, 2).reshape(T, E) for b in range(n)])
        h = maybe_bf16(h + ops.linear(a, w[p + "self_attn.out_proj.weight"], w[p + "self_attn.out_proj.bias"], bf16=bf16), bf16)
        y = ops.layer_norm(h, w[p + "layer_norm2.weight"], w[p + "layer_norm2.bias"], vc.layer_norm_eps, bf16=bf16)
        m = ops.quick_gelu(ops.linear(y, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], bf16=bf16), bf16=bf16)
        h = maybe_bf16(h + ops.linear(m, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], bf16=bf16), bf16)
    return h[:, 1:]


def project(w: dict, feats: np.ndarray, *, bf16=False) -> np.ndarray:
    """LlavaMultiModalProjector: Linear -> GELU(erf) -> Linear."""
    P = "model.multi_modal_projector."
    y = ops.gelu_erf(ops.linear(feats, w[P + "linear_1.weight"], w[P + "linear_1.bias"], bf16=bf16), bf16=bf16)
    return ops.linear(y, w[P + "linear_2.weight"], w[P + "linear_2.bias"], bf16=bf16)


def _best_resolution(size, pinpoints):
    """image_processing_utils.select_best_resolution: (h, w) pinpoint with max kept pixels, then min waste."""
    oh, ow = size
    cands = []
    for h, w in pinpoints:
        sc = min(w / ow, h / oh)
        eff = min(int(ow * sc) * int(oh * sc), ow * oh)
        cands.append((-eff, w * h - eff, len(cands), (h, w)))
    return min(cands)[3]


def pack_anyres(w: dict, cfg: LlavaCfg, feats: np.ndarray, image_size) -> np.ndarray:
    """LlavaNextModel.pack_image_features (modeling_llava_next.py:265-330) for ONE image.
    feats [1 + nh*nw, g*g, d] (view 0 = whole image resized) -> [n_tokens, d]."""
    g = cfg.vision.image_size // cfg.vision.patch_size
    bh, bw = _best_resolution(tuple(int(v) for v in image_size), cfg.image_grid_pinpoints)
    nh, nw = bh // cfg.vision.image_size, bw // cfg.vision.image_size
    d = feats.shape[-1]
    canvas = feats[1:].reshape(nh, nw, g, g, d).transpose(0, 2, 1, 3, 4).reshape(nh * g, nw * g, d)
    oh, ow = (int(v) for v in image_size)
    ch, cw = canvas.shape[:2]
    if ow / oh > cw / ch:  # unpad_image (:109-145)
        new_h = int(round(oh * (cw / ow), 7))
        pad = (ch - new_h) // 2
        canvas = canvas[pad:ch - pad]
    else:
        new_w = int(round(ow * (ch / oh), 7))
        pad = (cw - new_w) // 2
        canvas = canvas[:, pad:cw - pad]
    nl = np.broadcast_to(w["model.image_newline"][None, None], (canvas.shape[0], 1, d))
    canvas = np.concatenate([canvas, nl], 1)
    return np.concatenate([feats[0], canvas.reshape(-1, d)], 0)


def _embed(w: dict, cfg: LlavaCfg, input_ids, pixel_values, bf16, image_sizes, views_per_image):
    """inputs_embeds of one prompt: token embeddings with the image-token rows replaced by the projected (LLaVA-NeXT: packed)
    CLIP features (HF modeling_llava.py LlavaModel.forward)."""
    ids = np.asarray(input_ids).astype(np.int64)
    x = maybe_bf16(w[Q.T + "embed_tokens.weight"][ids], bf16)
    if pixel_values is not None:
        feats = project(w, clip_features(w, cfg, pixel_values, bf16=bf16), bf16=bf16)
        if cfg.image_grid_pinpoints:
            packed, v0 = [], 0
            for nv, size in zip(views_per_image, image_sizes):
                packed.append(pack_anyres(w, cfg, feats[v0:v0 + nv], size))
                v0 += nv
            feats = maybe_bf16(np.concatenate(packed, 0), bf16)
        x[ids == cfg.image_token_id] = feats.reshape(-1, feats.shape[-1])
    return ids, x


def loglikelihood(w: dict, cfg: LlavaCfg, input_ids: np.ndarray, pixel_values: np.ndarray | None, n_ctx: int, *, bf16=False,
                  image_sizes=None, views_per_image=None, return_logits=False):
    """The reference's LLaVA.loglikelihood arithmetic for ONE request (src/models/_llava_hf.py:229-252): `input_ids` is prompt +
    continuation with the image placeholders expanded, `n_ctx` the length of the prompt tokenised WITHOUT the expansion - the
    reference masks labels[:, :n_ctx] only (:232), so expanded image positions and the rest of the prompt stay in the loss.
    loss = HF's causal-LM loss: mean over i in [n_ctx, S) of -log softmax(logits[i - 1])[ids[i]] on fp32-upcast logits;
    max_equal = all(argmax(logits[i]) == ids[i] for i in [n_ctx, S)) - the UNSHIFTED comparison the reference makes (:246-251).
    Returns (loss, max_equal[, logits of positions n_ctx-1 .. S-1])."""
    qcfg = Q.Cfg(text=cfg.text, image_token_id=cfg.image_token_id)
    ids, x = _embed(w, cfg, input_ids, pixel_values, bf16, image_sizes, views_per_image)
    S = len(ids)
    pos3 = np.tile(np.arange(S)[None], (3, 1))
    h = Q.llm_forward(w, qcfg, x, pos3, Q.KVCache(cfg.text.num_hidden_layers), bf16=bf16)
    logits = Q.lm_head(w, qcfg, h[n_ctx - 1:], bf16=bf16).astype(np.float32)      # rows n_ctx-1 .. S-1
    z = logits[:-1] - logits[:-1].max(-1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(-1, keepdims=True))
    loss = float(-logp[np.arange(S - n_ctx), ids[n_ctx:]].mean())
    max_equal = bool((Q.greedy_argmax(logits[1:]) == ids[n_ctx:]).all())
    return (loss, max_equal, logits) if return_logits else (loss, max_equal)


def generate(w: dict, cfg: LlavaCfg, input_ids: np.ndarray, pixel_values: np.ndarray | None, max_new_tokens: int, *,
             bf16=False, eos_token_id: int | None = None, pad_token_id: int = 0, return_logits=False,
             image_sizes=None, views_per_image=None, forced_tokens=None):
    """Greedy generation for ONE prompt whose <image> placeholders are already expanded to one id per feature row.
    `forced_tokens`: teacher forcing - the token fed after step j is forced_tokens[j]; `out` still holds the argmax.
    LLaVA-NeXT: pixel_values holds all views of all images, `views_per_image` / `image_sizes` (h, w) split them."""
    qcfg = Q.Cfg(text=cfg.text, image_token_id=cfg.image_token_id)
    ids, x = _embed(w, cfg, input_ids, pixel_values, bf16, image_sizes, views_per_image)
    pos3 = np.tile(np.arange(len(ids))[None], (3, 1))
    cache = Q.KVCache(cfg.text.num_hidden_layers)
    h = Q.llm_forward(w, qcfg, x, pos3, cache, bf16=bf16)
    logits = Q.lm_head(w, qcfg, h[-1:], bf16=bf16)
    all_logits, out, cur = [logits[0].copy()], [], len(ids)
    for step in range(max_new_tokens):
        tok = int(Q.greedy_argmax(logits[0]))
        out.append(tok)
        if (eos_token_id is not None and tok == eos_token_id) or step == max_new_tokens - 1:
            break
        feed = tok if forced_tokens is None else int(forced_tokens[step])
        x = maybe_bf16(w[Q.T + "embed_tokens.weight"][np.array([feed])], bf16)
        h = Q.llm_forward(w, qcfg, x, np.full((3, 1), cur, np.int64), cache, bf16=bf16)
        logits = Q.lm_head(w, qcfg, h[-1:], bf16=bf16)
        all_logits.append(logits[0].copy())
        cur += 1
    return (np.array(out), np.stack(all_logits)) if return_logits else np.array(out)
