"""numpy restatement of the Qwen2-VL forward/greedy-generate path (TEST INFRASTRUCTURE ONLY).

The reference runs this arithmetic through HF transformers (`src/models/_qwen2_vl.py:319-329` calls
`Qwen2VLForConditionalGeneration.generate`; transformers is pinned at 4.47.0 in the reference's
uv.lock and is not vendored under /root/reference).  Each function below restates the published HF
algorithm and cites it as `HF:<line>` = transformers/models/qwen2_vl/modeling_qwen2_vl.py of the 5.15.0
copy installed in the build container (same maths as 4.47.0).  Pinned by tests/test_oracle_qwen2vl.py
against golden vectors generated from that HF code (tools/gen_golden.py).

Weights: dict name -> float32 ndarray with HF state-dict names (`model.visual...`, `model.language_model...`,
`lm_head.weight`).  `bf16=True` rounds activations to bfloat16 where a torch.bfloat16 module would.
"""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import np_ops as ops
from .np_ops import maybe_bf16


@dataclass
class VisionCfg:
    depth: int = 32
    embed_dim: int = 1280
    num_heads: int = 16
    mlp_ratio: float = 4.0
    hidden_size: int = 1536  # output dim (= LLM d_model)
    patch_size: int = 14
    temporal_patch_size: int = 2
    in_channels: int = 3
    spatial_merge_size: int = 2


@dataclass
class TextCfg:
    hidden_size: int = 1536
    num_hidden_layers: int = 28
    num_attention_heads: int = 12
    num_key_value_heads: int = 2
    intermediate_size: int = 8960
    vocab_size: int = 151936
    rms_norm_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: tuple = (16, 24, 24)
    tie_word_embeddings: bool = True


@dataclass
class Cfg:
    vision: VisionCfg = field(default_factory=VisionCfg)
    text: TextCfg = field(default_factory=TextCfg)
    image_token_id: int = 151655


V = "model.visual."
T = "model.language_model."


# ------------------------------------------------------------------ vision tower
def vision_position_ids(grid_thw, merge: int) -> np.ndarray:
    """transformers/vision_utils.py get_vision_position_ids: (h, w) per patch, merge-block-major order."""
    out = []
    for t, h, w in np.asarray(grid_thw).tolist():
        hp, wp = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
        shape = (h // merge, merge, w // merge, merge)
        hp = hp.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        wp = wp.reshape(shape).transpose(0, 2, 1, 3).reshape(-1)
        out.append(np.tile(np.stack([hp, wp], -1), (t, 1)))
    return np.concatenate(out, 0).astype(np.int64)


def _rotate_half(x):
    h = x.shape[-1] // 2
    return np.concatenate([-x[..., h:], x[..., :h]], -1)


def _attn(q, k, v, scale, causal, bf16):
    """eager_attention_forward (HF:317-339) for [H, S, hd] operands (GQA already expanded)."""
    s = np.einsum("hqd,hkd->hqk", q, k).astype(np.float32) * np.float32(scale)
    if causal:
        sq, sk = s.shape[1], s.shape[2]
        mask = np.arange(sk)[None, :] > (np.arange(sq)[:, None] + (sk - sq))
        s = np.where(mask[None], np.float32(-np.inf), s)
    p = maybe_bf16(ops.softmax(s, -1), bf16)
    return maybe_bf16(np.einsum("hqk,hkd->hqd", p, v).astype(np.float32), bf16)


def vit_forward(w: dict, cfg: Cfg, pixel_values: np.ndarray, grid_thw, *, bf16=False, taps: dict | None = None):
    """Qwen2VisionTransformerPretrainedModel.forward (HF:700-731) -> merged embeddings [T/4, hidden]."""
    vc = cfg.vision
    E, H = vc.embed_dim, vc.num_heads
    hd = E // H
    x = maybe_bf16(pixel_values, bf16)
    # PatchEmbed (HF:268-275): Conv3d with kernel == stride == a GEMM over the flattened patch
    x = ops.linear(x, w[V + "patch_embed.proj.weight"].reshape(E, -1), bf16=bf16)
    if taps is not None:
        taps["patch_embed"] = x.copy()
    pos = vision_position_ids(grid_thw, vc.spatial_merge_size)
    # VisionRotaryEmbedding(head_dim // 2) (HF:239-248), emb = cat(freqs, freqs) (HF:716-718)
    dim = hd // 2
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
    freqs = (pos[:, :, None].astype(np.float32) * inv_freq[None, None, :]).reshape(pos.shape[0], -1)
    emb = np.concatenate([freqs, freqs], -1)
    cos, sin = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)
    lens = [int(t * h * ww) for t, h, ww in np.asarray(grid_thw).tolist()]
    starts = np.concatenate([[0], np.cumsum(lens)])
    for i in range(vc.depth):
        p = f"{V}blocks.{i}."
        h1 = ops.layer_norm(x, w[p + "norm1.weight"], w[p + "norm1.bias"], 1e-6, bf16=bf16)
        qkv = ops.linear(h1, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"], bf16=bf16)
        n = qkv.shape[0]
        qkv = qkv.reshape(n, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        # apply_rotary_pos_emb_vision (HF:225-236): fp32 maths, one rounding
        q = maybe_bf16(q * cos[:, None, :] + _rotate_half(q) * sin[:, None, :], bf16)
        k = maybe_bf16(k * cos[:, None, :] + _rotate_half(k) * sin[:, None, :], bf16)
        o = np.empty((n, H, hd), np.float32)
        for s0, s1 in zip(starts[:-1], starts[1:]):  # per-image attention (HF:398-419)
            o[s0:s1] = _attn(q[s0:s1].transpose(1, 0, 2), k[s0:s1].transpose(1, 0, 2),
                             v[s0:s1].transpose(1, 0, 2), hd ** -0.5, False, bf16).transpose(1, 0, 2)
        a = ops.linear(o.reshape(n, E), w[p + "attn.proj.weight"], w[p + "attn.proj.bias"], bf16=bf16)
        x = maybe_bf16(x + a, bf16)
        h2 = ops.layer_norm(x, w[p + "norm2.weight"], w[p + "norm2.bias"], 1e-6, bf16=bf16)
        m = ops.quick_gelu(ops.linear(h2, w[p + "mlp.fc1.weight"], w[p + "mlp.fc1.bias"], bf16=bf16), bf16=bf16)
        m = ops.linear(m, w[p + "mlp.fc2.weight"], w[p + "mlp.fc2.bias"], bf16=bf16)
        x = maybe_bf16(x + m, bf16)
        if taps is not None and i == 0:
            taps["block0"] = x.copy()
    # PatchMerger (HF:288-291)
    mu = vc.spatial_merge_size ** 2
    y = ops.layer_norm(x, w[V + "merger.ln_q.weight"], w[V + "merger.ln_q.bias"], 1e-6, bf16=bf16)
    y = y.reshape(-1, E * mu)
    y = ops.gelu_erf(ops.linear(y, w[V + "merger.mlp.0.weight"], w[V + "merger.mlp.0.bias"], bf16=bf16), bf16=bf16)
    return ops.linear(y, w[V + "merger.mlp.2.weight"], w[V + "merger.mlp.2.bias"], bf16=bf16)


# ------------------------------------------------------------------ decoder
def rope_index(input_ids: np.ndarray, grid_thw, cfg: Cfg):
    """Qwen2VLModel.get_rope_index (HF:914-1019) for one un-padded prompt: pos [3, S], delta."""
    merge = cfg.vision.spatial_merge_size
    ids = np.asarray(input_ids).tolist()
    grids = iter(np.asarray(grid_thw).tolist()) if grid_thw is not None else iter(())
    is_img = [int(t == cfg.image_token_id) for t in ids]
    groups, start = [], 0
    for i in range(1, len(ids) + 1):
        if i == len(ids) or is_img[i] != is_img[start]:
            groups.append((is_img[start], start, i))
            start = i
    cur, chunks = 0, []
    for kind, s0, s1 in groups:
        if kind == 0:
            n = s1 - s0
            chunks.append(np.tile(np.arange(n)[None, :], (3, 1)) + cur)
            cur += n
        else:
            t, h, w = next(grids)
            gh, gw = h // merge, w // merge
            tt, hh, ww = np.meshgrid(np.arange(t), np.arange(gh) + cur, np.arange(gw) + cur, indexing="ij")
            vp = np.stack([tt.reshape(-1) + cur, hh.reshape(-1), ww.reshape(-1)], 0)
            chunks.append(vp)
            cur += max(h, w) // merge
    pos = np.concatenate(chunks, 1).astype(np.int64)
    return pos, int(pos.max() + 1 - len(ids))


def _mrope_cos_sin(pos3: np.ndarray, tc: TextCfg, hd: int, bf16: bool):
    """Qwen2VLRotaryEmbedding.forward (HF:156-170) + section selection (HF:214-219) -> cos/sin [S, hd]."""
    inv_freq = (1.0 / (np.float32(tc.rope_theta) ** (np.arange(0, hd, 2, dtype=np.float32) / np.float32(hd)))).astype(np.float32)
    freqs = pos3[:, :, None].astype(np.float32) * inv_freq[None, None, :]  # [3, S, hd/2]
    emb = np.concatenate([freqs, freqs], -1)
    cos, sin = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)
    cos, sin = maybe_bf16(cos, bf16), maybe_bf16(sin, bf16)
    sec = list(tc.mrope_section) * 2
    bounds = np.concatenate([[0], np.cumsum(sec)])
    c = np.concatenate([cos[i % 3][:, bounds[i]:bounds[i + 1]] for i in range(len(sec))], -1)
    s = np.concatenate([sin[i % 3][:, bounds[i]:bounds[i + 1]] for i in range(len(sec))], -1)
    return c, s


class KVCache:
    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def update(self, i, k, v):
        self.k[i] = k if self.k[i] is None else np.concatenate([self.k[i], k], 1)
        self.v[i] = v if self.v[i] is None else np.concatenate([self.v[i], v], 1)
        return self.k[i], self.v[i]


def llm_forward(w: dict, cfg: Cfg, x: np.ndarray, pos3: np.ndarray, cache: KVCache, *, bf16=False,
                taps: dict | None = None, fp8: dict | None = None) -> np.ndarray:
    """Qwen2VLTextModel layers + final norm (HF:762-846, :575-625) on embeddings x [S, d]; appends to cache.
    `fp8` (name -> (e4m3 codes, row scales), oracle/fp8_np.quantize_decoder): the seven decoder projections run as
    per-token-quantised fp8 linears instead (config #5; everything else stays in the model dtype)."""
    tc = cfg.text

    def lin(name, inp, bias=None):
        if fp8 is not None:
            from . import fp8_np

            return fp8_np.linear_fp8(inp, *fp8[name], bias, bf16=bf16)
        return ops.linear(inp, w[name], bias, bf16=bf16)

    d, Hq, Hkv = tc.hidden_size, tc.num_attention_heads, tc.num_key_value_heads
    hd = d // Hq
    cos, sin = _mrope_cos_sin(pos3, tc, hd, bf16)
    for i in range(tc.num_hidden_layers):
        p = f"{T}layers.{i}."
        h = ops.rms_norm(x, w[p + "input_layernorm.weight"], tc.rms_norm_eps, bf16=bf16)
        # Qwen2 has q/k/v biases, Llama-family decoders (LLaVA) do not
        q = lin(p + "self_attn.q_proj.weight", h, w.get(p + "self_attn.q_proj.bias"))
        k = lin(p + "self_attn.k_proj.weight", h, w.get(p + "self_attn.k_proj.bias"))
        v = lin(p + "self_attn.v_proj.weight", h, w.get(p + "self_attn.v_proj.bias"))
        S = x.shape[0]
        q = q.reshape(S, Hq, hd).transpose(1, 0, 2)
        k = k.reshape(S, Hkv, hd).transpose(1, 0, 2)
        v = v.reshape(S, Hkv, hd).transpose(1, 0, 2)
        # apply_multimodal_rotary_pos_emb (HF:221-222): each product / the sum round in the model dtype
        q = maybe_bf16(maybe_bf16(q * cos[None], bf16) + maybe_bf16(_rotate_half(q) * sin[None], bf16), bf16)
        k = maybe_bf16(maybe_bf16(k * cos[None], bf16) + maybe_bf16(_rotate_half(k) * sin[None], bf16), bf16)
        kk, vv = cache.update(i, k, v)
        rep = Hq // Hkv
        a = _attn(q, np.repeat(kk, rep, 0), np.repeat(vv, rep, 0), hd ** -0.5, True, bf16)
        a = a.transpose(1, 0, 2).reshape(S, Hq * hd)
        x = maybe_bf16(x + lin(p + "self_attn.o_proj.weight", a), bf16)
        if taps is not None and i == 0:
            taps["layer0_post_attn"] = x.copy()
        h = ops.rms_norm(x, w[p + "post_attention_layernorm.weight"], tc.rms_norm_eps, bf16=bf16)
        g = ops.silu(lin(p + "mlp.gate_proj.weight", h), bf16=bf16)
        u = lin(p + "mlp.up_proj.weight", h)
        m = lin(p + "mlp.down_proj.weight", maybe_bf16(g * u, bf16))
        x = maybe_bf16(x + m, bf16)
    return ops.rms_norm(x, w[T + "norm.weight"], tc.rms_norm_eps, bf16=bf16)


def lm_head(w: dict, cfg: Cfg, h: np.ndarray, *, bf16=False) -> np.ndarray:
    wt = w[T + "embed_tokens.weight"] if cfg.text.tie_word_embeddings and "lm_head.weight" not in w else w["lm_head.weight"]
    return ops.linear(h, wt, bf16=bf16)


def greedy_argmax(logits: np.ndarray) -> np.ndarray:
    """argmax with the lowest index on ties (np.argmax / torch CPU behaviour)."""
    return np.argmax(logits, -1)


def repetition_penalty_scores(logits: np.ndarray, seen_ids, penalty: float) -> np.ndarray:
    """HF RepetitionPenaltyLogitsProcessor (transformers generation/logits_process.py; third-party, absent from /root/reference,
    restated from its published behaviour and PINNED on the installed transformers in tests/test_oracle_sampling.py): the
    next-token scores are float32; every token id that occurs in `seen_ids` (HF's input_ids: the prompt and everything generated so
    far) gets score < 0 ? score * penalty : score / penalty - once, however often the id occurs (gather / scatter semantics).
    In force in the reference whenever the checkpoint's generation_config.json has `repetition_penalty` != 1: HF merges the file's
    fields into `generate` calls that do not pass them (/root/reference/src/models/_qwen2_vl.py:319-329 does not), greedy or not."""
    s = np.asarray(logits, np.float32).copy()
    idx = np.unique(np.asarray(seen_ids, np.int64))
    p = np.float32(penalty)
    v = s[idx]
    s[idx] = np.where(v < 0, v * p, v / p).astype(np.float32)
    return s


def generate(w: dict, cfg: Cfg, input_ids: np.ndarray, pixel_values: np.ndarray | None, grid_thw, max_new_tokens: int,
             *, bf16=False, eos_token_id: int | None = None, pad_token_id: int = 0, return_logits=False, fp8: dict | None = None,
             forced_tokens=None, vit=None, repetition_penalty: float = 1.0):
    """Greedy generation for ONE prompt (reference batch size is 1, src/models/_base.py:103-104):
    HF:1144-1205 (embed + image scatter + rope index) then the GenerationMixin greedy loop.
    `forced_tokens` (teacher forcing, parity tests): the token fed after step j is forced_tokens[j] instead of the
    argmax, so step j+1's logits are conditional on a given continuation; `out` still holds the argmax.
    `repetition_penalty` != 1: `repetition_penalty_scores` on every step's logits over the prompt ids + the tokens fed so far
    before the argmax (the returned logits stay the raw ones)."""
    tc = cfg.text
    ids = np.asarray(input_ids).astype(np.int64)
    x = maybe_bf16(w[T + "embed_tokens.weight"][ids], bf16)
    if pixel_values is not None:
        img = (vit or vit_forward)(w, cfg, pixel_values, grid_thw, bf16=bf16)   # `vit`: oracle/qwen25vl_np.py plugs its tower in here
        x[ids == cfg.image_token_id] = img
        pos3, delta = rope_index(ids, grid_thw, cfg)
    else:
        pos3 = np.tile(np.arange(len(ids))[None], (3, 1))
        delta = 0
    cache = KVCache(tc.num_hidden_layers)
    h = llm_forward(w, cfg, x, pos3, cache, bf16=bf16, fp8=fp8)
    logits = lm_head(w, cfg, h[-1:], bf16=bf16)
    all_logits = [logits[0].copy()]
    out, done = [], False
    cur_len = len(ids)
    history = [int(t) for t in ids]
    for step in range(max_new_tokens):
        scores = logits[0] if repetition_penalty == 1.0 else repetition_penalty_scores(logits[0], history, repetition_penalty)
        tok = pad_token_id if done else int(greedy_argmax(scores))
        out.append(tok)
        if eos_token_id is not None and tok == eos_token_id:
            done = True
        if step == max_new_tokens - 1 or done:
            if done:
                break
            continue
        feed = tok if forced_tokens is None else int(forced_tokens[step])
        history.append(feed)
        x = maybe_bf16(w[T + "embed_tokens.weight"][np.array([feed])], bf16)
        p = np.full((3, 1), cur_len + delta, np.int64)  # HF:1130-1137: arange(past, past+1) + rope_deltas
        h = llm_forward(w, cfg, x, p, cache, bf16=bf16, fp8=fp8)
        logits = lm_head(w, cfg, h[-1:], bf16=bf16)
        all_logits.append(logits[0].copy())
        cur_len += 1
    return (np.array(out), np.stack(all_logits)) if return_logits else np.array(out)
