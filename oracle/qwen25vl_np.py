"""numpy restatement of the Qwen2.5-VL vision tower (TEST INFRASTRUCTURE ONLY; see oracle/__init__.py).

The reference reaches this arithmetic through `Qwen2_5_VLForConditionalGeneration` (/root/reference/src/models/_qwen2_vl.py:106-115,
registry names `qwen2.5-vl-7b` / `qwen2.5-vl-3b` at :635-648); transformers is a third-party dependency (pinned 4.47.0 in the
reference's uv.lock, 5.15.0 in the build container: same maths).  `HF25:<line>` = transformers/models/qwen2_5_vl/
modeling_qwen2_5_vl.py of that copy, `VU` = transformers/vision_utils.py.  What differs from Qwen2-VL (oracle/qwen2vl_np.py):

* vision blocks use RMSNorm (HF25:65-78) and a gated MLP with biases, down(silu(gate(x)) * up(x)) (HF25:85-96, :300);
* attention is WINDOWED: the merged 2x2 token groups are reordered so that the groups of one 112 x 112-pixel window (4 x 4
  groups = 64 patches) are contiguous, most layers attend inside a window only, the layers in `fullatt_block_indexes` inside
  the whole image (HF25:423-461); the merger output is put back into the original order (HF25:463-465);
* the merger's ln_q is an RMSNorm (HF25:137-150).
The decoder (RMSNorm, q/k/v biases, M-RoPE, SwiGLU) is the Qwen2-VL one - `qwen2vl_np.llm_forward` / `generate` are reused; for
images the rope index is identical as well (the 2.5 changes concern video timestamps only).
Pinned by tests/test_oracle_qwen25vl.py against goldens generated from HF (tools/gen_golden.py qwen25)."""

from __future__ import annotations

from dataclasses import dataclass, field

import numpy as np

from . import np_ops as ops
from . import qwen2vl_np as Q
from .np_ops import maybe_bf16

V = Q.V


@dataclass
class Vision25Cfg:
    depth: int = 32
    embed_dim: int = 1280          # HF: hidden_size
    num_heads: int = 16
    intermediate_size: int = 3420
    hidden_size: int = 3584        # HF: out_hidden_size (= LLM d_model)
    patch_size: int = 14
    temporal_patch_size: int = 2
    in_channels: int = 3
    spatial_merge_size: int = 2
    window_size: int = 112
    fullatt_block_indexes: tuple = (7, 15, 23, 31)


@dataclass
class Cfg25:
    vision: Vision25Cfg = field(default_factory=Vision25Cfg)
    text: Q.TextCfg = field(default_factory=Q.TextCfg)
    image_token_id: int = 151655


def vision_window_index(grid_thw, merge: int, window_size: int, patch_size: int):
    """VU get_vision_window_index: (window_index over merged groups, cu_window_seqlens in PATCH units, zero-length windows removed).
    NB the padding rule pads a FULL extra window when a side is already a multiple of the window (`ws - side % ws`): those
    windows are empty and disappear in `unique_consecutive`."""
    window_index, cu, base = [], [0], 0
    ws = window_size // merge // patch_size
    unit = merge * merge
    for t, h, w in np.asarray(grid_thw).tolist():
        gh, gw = h // merge, w // merge
        index = np.arange(t * gh * gw).reshape(t, gh, gw)
        pad_h, pad_w = ws - gh % ws, ws - gw % ws
        nh, nw = (gh + pad_h) // ws, (gw + pad_w) // ws
        padded = np.full((t, gh + pad_h, gw + pad_w), -100, np.int64)
        padded[:, :gh, :gw] = index
        padded = padded.reshape(t, nh, ws, nw, ws).transpose(0, 1, 3, 2, 4).reshape(t, nh * nw, ws, ws)
        seqlens = (padded != -100).sum((2, 3)).reshape(-1)
        flat = padded.reshape(-1)
        window_index.append(flat[flat != -100] + base)
        cu.extend((np.cumsum(seqlens) * unit + cu[-1]).tolist())
        base += t * gh * gw
    cu = np.asarray(cu, np.int64)
    keep = np.concatenate([[True], cu[1:] != cu[:-1]])   # torch.unique_consecutive
    return np.concatenate(window_index).astype(np.int64), cu[keep]


def vit_forward(w: dict, cfg: Cfg25, pixel_values: np.ndarray, grid_thw, *, bf16=False, taps: dict | None = None):
    """Qwen2_5_VisionTransformerPretrainedModel.forward (HF25:408-470) -> merged embeddings [T/4, out_hidden], original order."""
    vc = cfg.vision
    E, H, mu = vc.embed_dim, vc.num_heads, vc.spatial_merge_size ** 2
    hd = E // H
    x = ops.linear(maybe_bf16(pixel_values, bf16), w[V + "patch_embed.proj.weight"].reshape(E, -1), bf16=bf16)
    n = x.shape[0]
    pos = Q.vision_position_ids(grid_thw, vc.spatial_merge_size)
    widx, cu_win = vision_window_index(grid_thw, vc.spatial_merge_size, vc.window_size, vc.patch_size)
    x = x.reshape(n // mu, mu, E)[widx].reshape(n, E)                      # HF25:437-440
    pos = pos.reshape(n // mu, mu, 2)[widx].reshape(n, 2)                  # HF25:442-445 (the rotary table rows travel along)
    dim = hd // 2
    inv_freq = (1.0 / (10000.0 ** (np.arange(0, dim, 2, dtype=np.float32) / np.float32(dim)))).astype(np.float32)
    freqs = (pos[:, :, None].astype(np.float32) * inv_freq[None, None, :]).reshape(n, -1)
    emb = np.concatenate([freqs, freqs], -1)
    cos, sin = np.cos(emb).astype(np.float32), np.sin(emb).astype(np.float32)
    lens = [int(t * h * ww) for t, h, ww in np.asarray(grid_thw).tolist()]
    cu_full = np.concatenate([[0], np.cumsum(lens)])
    for i in range(vc.depth):
        p = f"{V}blocks.{i}."
        cu = cu_full if i in vc.fullatt_block_indexes else cu_win          # HF25:448-454
        h1 = ops.rms_norm(x, w[p + "norm1.weight"], 1e-6, bf16=bf16)
        qkv = ops.linear(h1, w[p + "attn.qkv.weight"], w[p + "attn.qkv.bias"], bf16=bf16).reshape(n, 3, H, hd)
        q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
        q = maybe_bf16(q * cos[:, None, :] + Q._rotate_half(q) * sin[:, None, :], bf16)   # HF25:160-172: fp32 maths, one rounding
        k = maybe_bf16(k * cos[:, None, :] + Q._rotate_half(k) * sin[:, None, :], bf16)
        o = np.empty((n, H, hd), np.float32)
        for s0, s1 in zip(cu[:-1], cu[1:]):                                 # HF25:264-286: every chunk on its own
            o[s0:s1] = Q._attn(q[s0:s1].transpose(1, 0, 2), k[s0:s1].transpose(1, 0, 2), v[s0:s1].transpose(1, 0, 2),
                               hd ** -0.5, False, bf16).transpose(1, 0, 2)
        a = ops.linear(o.reshape(n, E), w[p + "attn.proj.weight"], w[p + "attn.proj.bias"], bf16=bf16)
        x = maybe_bf16(x + a, bf16)
        h2 = ops.rms_norm(x, w[p + "norm2.weight"], 1e-6, bf16=bf16)
        g = ops.silu(ops.linear(h2, w[p + "mlp.gate_proj.weight"], w[p + "mlp.gate_proj.bias"], bf16=bf16), bf16=bf16)
        u = ops.linear(h2, w[p + "mlp.up_proj.weight"], w[p + "mlp.up_proj.bias"], bf16=bf16)
        m = ops.linear(maybe_bf16(g * u, bf16), w[p + "mlp.down_proj.weight"], w[p + "mlp.down_proj.bias"], bf16=bf16)
        x = maybe_bf16(x + m, bf16)
        if taps is not None and i == 0:
            taps["block0"] = x.copy()
    y = ops.rms_norm(x, w[V + "merger.ln_q.weight"], 1e-6, bf16=bf16).reshape(-1, E * mu)
    y = ops.gelu_erf(ops.linear(y, w[V + "merger.mlp.0.weight"], w[V + "merger.mlp.0.bias"], bf16=bf16), bf16=bf16)
    y = ops.linear(y, w[V + "merger.mlp.2.weight"], w[V + "merger.mlp.2.bias"], bf16=bf16)
    return y[np.argsort(widx)]                                              # HF25:463-465


def generate(w: dict, cfg: Cfg25, input_ids, pixel_values, grid_thw, max_new_tokens: int, **kw):
    """Greedy generation: the Qwen2-VL loop (qwen2vl_np.generate) with this vision tower."""
    return Q.generate(w, cfg, input_ids, pixel_values, grid_thw, max_new_tokens, vit=vit_forward, **kw)
