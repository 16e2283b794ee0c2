"""LLaVA (CLIP ViT-L/14-336 + GELU projector + Llama-family decoder) on libowc_hip.so.

MI355X replacement for what the reference's `LLaVA.generate_until`
(/root/reference/src/models/_llava_hf.py:260-392) delegates to `Llava(Next)ForConditionalGeneration.generate`.
The decoder is the same prefill/decode driver as Qwen2-VL (no q/k/v bias, three identical rope streams = 1-D
RoPE); the image branch is `owc_clip_forward`.  Host code here is integer bookkeeping only.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from .. import _lib
from . import anyres
from .qwen2vl import BF16, F32, Qwen2VLEngine, Qwen2VLWeights

CLIP_PATCH = 14


@dataclass
class LlavaDims:
    # CLIP vision tower
    v_layers: int = 24
    v_embed: int = 1024
    v_heads: int = 16
    v_mlp: int = 4096
    image_size: int = 336
    feature_layer: int = -2            # hidden_states index the projector reads (HF vision_feature_layer)
    patch_k: int = 640                 # 588 zero-padded so K % 64 == 0
    v_ln_eps: float = 1e-5
    # decoder (Llama / Mistral / Yi)
    n_layers: int = 32
    d_model: int = 4096
    n_q_heads: int = 32
    n_kv_heads: int = 32
    head_dim: int = 128
    d_ff: int = 11008
    vocab: int = 32064
    tie_embeddings: bool = False
    rms_eps: float = 1e-5
    rope_theta: float = 10000.0
    mrope_section: tuple = (16, 24, 24)  # unused split: the three position streams are identical
    image_token_id: int = 32000
    max_positions: int = 4096
    merge: int = 1
    decoder_dtype: str = "bf16"        # "fp8": see Qwen2VLDims.decoder_dtype
    # LLaVA-NeXT anyres (None -> LLaVA-1.5: one 336x336 view)
    grid_pinpoints: tuple | None = None

    @property
    def grid(self) -> int:
        return self.image_size // CLIP_PATCH

    @property
    def tokens(self) -> int:           # CLS + patches per view
        return 1 + self.grid ** 2

    @property
    def v_run_layers(self) -> int:
        return self.v_layers + 1 + self.feature_layer if self.feature_layer < 0 else self.feature_layer


NEXT_PINPOINTS = ((336, 672), (672, 336), (672, 672), (1008, 336), (336, 1008))

DIMS = {
    # public config.json values of the llava-hf checkpoints the reference registers (_llava_hf.py:586-615)
    "llava-1.5-7b": LlavaDims(),
    "llava-1.5-13b": LlavaDims(n_layers=40, d_model=5120, n_q_heads=40, n_kv_heads=40, d_ff=13824),
    "llava-next-vicuna-7b": LlavaDims(grid_pinpoints=NEXT_PINPOINTS, max_positions=8192),
    "llava-next-mistral-7b": LlavaDims(n_kv_heads=8, d_ff=14336, rope_theta=1e6, grid_pinpoints=NEXT_PINPOINTS, max_positions=8192),
    # BASELINE.json config #4 (llava-v1.6-34b-hf: CLIP-L/336 + Yi-34B); reachable through `custom-model`
    "llava-next-34b": LlavaDims(n_layers=60, d_model=7168, n_q_heads=56, n_kv_heads=8, d_ff=20480, vocab=64064, rope_theta=5e6,
                                image_token_id=64000, grid_pinpoints=NEXT_PINPOINTS, max_positions=8192),
    "tiny": LlavaDims(v_layers=3, v_embed=128, v_heads=2, v_mlp=256, image_size=56, n_layers=2, d_model=256, n_q_heads=2,
                      n_kv_heads=1, d_ff=512, vocab=512, image_token_id=500, max_positions=2048),
    "tiny-next": LlavaDims(v_layers=3, v_embed=128, v_heads=2, v_mlp=256, image_size=56, n_layers=2, d_model=256, n_q_heads=2,
                           n_kv_heads=1, d_ff=512, vocab=512, image_token_id=500, max_positions=2048,
                           grid_pinpoints=((56, 112), (112, 56), (112, 112), (168, 56), (56, 168))),
}

VT, PJ, LM = "model.vision_tower.", "model.multi_modal_projector.", "model.language_model."


def param_shapes(d: LlavaDims) -> dict[str, tuple]:
    """HF state-dict names -> shapes (LlavaForConditionalGeneration / LlavaNextForConditionalGeneration)."""
    E, F, dm, hd = d.v_embed, d.v_mlp, d.d_model, d.head_dim
    s = {VT + "embeddings.class_embedding": (E,), VT + "embeddings.patch_embedding.weight": (E, 3, CLIP_PATCH, CLIP_PATCH),
         VT + "embeddings.position_embedding.weight": (d.tokens, E), VT + "pre_layrnorm.weight": (E,), VT + "pre_layrnorm.bias": (E,)}
    for i in range(d.v_run_layers):
        p = f"{VT}encoder.layers.{i}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            s[p + f"self_attn.{n}.weight"], s[p + f"self_attn.{n}.bias"] = (E, E), (E,)
        s.update({p + "layer_norm1.weight": (E,), p + "layer_norm1.bias": (E,), p + "layer_norm2.weight": (E,), p + "layer_norm2.bias": (E,),
                  p + "mlp.fc1.weight": (F, E), p + "mlp.fc1.bias": (F,), p + "mlp.fc2.weight": (E, F), p + "mlp.fc2.bias": (E,)})
    s.update({PJ + "linear_1.weight": (dm, E), PJ + "linear_1.bias": (dm,), PJ + "linear_2.weight": (dm, dm), PJ + "linear_2.bias": (dm,)})
    if d.grid_pinpoints:
        s["model.image_newline"] = (dm,)
    s[LM + "embed_tokens.weight"] = (d.vocab, dm)
    for i in range(d.n_layers):
        p = f"{LM}layers.{i}."
        s.update({p + "self_attn.q_proj.weight": (d.n_q_heads * hd, dm), p + "self_attn.k_proj.weight": (d.n_kv_heads * hd, dm),
                  p + "self_attn.v_proj.weight": (d.n_kv_heads * hd, dm), p + "self_attn.o_proj.weight": (dm, d.n_q_heads * hd),
                  p + "mlp.gate_proj.weight": (d.d_ff, dm), p + "mlp.up_proj.weight": (d.d_ff, dm), p + "mlp.down_proj.weight": (dm, d.d_ff),
                  p + "input_layernorm.weight": (dm,), p + "post_attention_layernorm.weight": (dm,)})
    s[LM + "norm.weight"] = (dm,)
    if not d.tie_embeddings:
        s["lm_head.weight"] = (d.vocab, dm)
    return s


class LlavaWeights(Qwen2VLWeights):
    """Device-resident bf16 weights + owc_clip_weights / owc_llm_weights structs (include/owc.h)."""

    def __init__(self, dims: LlavaDims, device: torch.device):
        self.dims = dims
        self.device = device
        self._keep = []
        self.clip = _lib.ClipWeights()
        self.llm = _lib.LlmWeights()
        self.newline: torch.Tensor | None = None

    @classmethod
    def random(cls, dims: LlavaDims, device, seed: int = 1234) -> "LlavaWeights":
        self = cls(dims, torch.device(device))
        shapes = param_shapes(dims)
        gen = torch.Generator(device=self.device)
        counter = [0]

        def get(name):
            shape = shapes[name]
            counter[0] += 1
            gen.manual_seed(seed * 100003 + counter[0])
            if name.endswith("bias"):
                t = torch.randn(shape, generator=gen, device=self.device, dtype=F32) * 0.02
            elif "norm" in name:
                t = 1.0 + torch.randn(shape, generator=gen, device=self.device, dtype=F32) * 0.02
            elif "embed" in name or "newline" in name:
                t = torch.randn(shape, generator=gen, device=self.device, dtype=BF16) * 0.05
            else:
                t = torch.randn(shape, generator=gen, device=self.device, dtype=BF16) * (1.0 / np.sqrt(int(np.prod(shape[1:]))))
            return t.to(BF16).contiguous()

        self._build(get)
        return self

    def _build(self, get) -> None:
        d = self.dims
        E = d.v_embed
        n_run = d.v_run_layers
        vl = (_lib.VitLayer * n_run)()
        for i in range(n_run):
            p = f"{VT}encoder.layers.{i}."
            vl[i].qkv_w = self._k(torch.cat([get(p + f"self_attn.{n}.weight") for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous())
            vl[i].qkv_b = self._k(torch.cat([get(p + f"self_attn.{n}.bias") for n in ("q_proj", "k_proj", "v_proj")], 0).contiguous())
            for f, n in (("ln1_w", "layer_norm1.weight"), ("ln1_b", "layer_norm1.bias"), ("proj_w", "self_attn.out_proj.weight"),
                         ("proj_b", "self_attn.out_proj.bias"), ("ln2_w", "layer_norm2.weight"), ("ln2_b", "layer_norm2.bias"),
                         ("fc1_w", "mlp.fc1.weight"), ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias")):
                setattr(vl[i], f, self._k(get(p + n)))
        self._clip_layers = vl
        c = self.clip
        c.n_layers, c.embed_dim, c.num_heads, c.mlp_hidden = n_run, E, d.v_heads, d.v_mlp
        c.patch_k, c.tokens, c.out_dim, c.ln_eps = d.patch_k, d.tokens, d.d_model, d.v_ln_eps
        pw = torch.zeros((E, d.patch_k), dtype=BF16, device=self.device)
        pw[:, :3 * CLIP_PATCH ** 2] = get(VT + "embeddings.patch_embedding.weight").reshape(E, -1)
        c.patch_w = self._k(pw)
        # row 0 = bf16(class_embedding + position_embedding[0]): the same single rounding HF's bf16 add makes
        pos = get(VT + "embeddings.position_embedding.weight").clone()
        pos[0] = (pos[0].float() + get(VT + "embeddings.class_embedding").float()).to(BF16)
        c.pos_cls = self._k(pos)
        c.pre_ln_w, c.pre_ln_b = self._k(get(VT + "pre_layrnorm.weight")), self._k(get(VT + "pre_layrnorm.bias"))
        c.layers = C.cast(vl, C.POINTER(_lib.VitLayer))
        c.proj1_w, c.proj1_b = self._k(get(PJ + "linear_1.weight")), self._k(get(PJ + "linear_1.bias"))
        c.proj2_w, c.proj2_b = self._k(get(PJ + "linear_2.weight")), self._k(get(PJ + "linear_2.bias"))
        if d.grid_pinpoints:
            self.newline = get("model.image_newline")
            self._keep.append(self.newline)
        self._build_llm(get, LM, qkv_bias=False)


class LlavaEngine(Qwen2VLEngine):
    """Batched LLaVA forward: uint8/bf16 views -> projected CLIP features -> greedy token ids."""

    def __init__(self, weights: LlavaWeights, *, clip_chunk_views: int = 96, **kw):
        super().__init__(weights, **kw)
        self.clip_chunk_views = clip_chunk_views

    # -- image branch ------------------------------------------------------------------
    def encode_views(self, patches: torch.Tensor) -> torch.Tensor:
        """patches [n_views * grid^2, >= patch_k] bf16 (owc_clip_patchify_u8 layout) -> projected features
        [n_views * tokens (+1 newline row for NeXT), d_model]; row v*tokens is the (unused) CLS row of view v."""
        d = self.d
        P = d.grid ** 2
        assert patches.dtype == BF16 and patches.is_cuda and patches.stride(1) == 1 and patches.shape[0] % P == 0
        n = patches.shape[0] // P
        extra = 1 if self.w.newline is not None else 0
        out = torch.empty((n * d.tokens + extra, d.d_model), dtype=BF16, device=self.device)
        if extra:
            out[-1] = self.w.newline
        for v0 in range(0, n, self.clip_chunk_views):
            v1 = min(n, v0 + self.clip_chunk_views)
            nbytes = self._lib.owc_clip_workspace_bytes(C.byref(self.w.clip), v1 - v0)
            ws = self._workspace(nbytes)
            rc = self._lib.owc_clip_forward(self._ctx, C.byref(self.w.clip), patches[v0 * P:].data_ptr(), patches.stride(0), v1 - v0,
                                            out[v0 * d.tokens:].data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, self.dev_index)
        return out

    def patchify(self, views_u8: torch.Tensor, mean, std) -> torch.Tensor:
        """uint8 [n, 3, S, S] (device) -> bf16 patches [n * grid^2, patch_k]."""
        d = self.d
        n = views_u8.shape[0]
        assert views_u8.dtype == torch.uint8 and views_u8.is_cuda and views_u8.is_contiguous() and tuple(views_u8.shape[1:]) == (3, d.image_size, d.image_size)
        out = torch.empty((n * d.grid ** 2, d.patch_k), dtype=BF16, device=self.device)
        m, s = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        rc = self._lib.owc_clip_patchify_u8(self._ctx, views_u8.data_ptr(), out.data_ptr(), out.stride(0), d.patch_k, n, d.image_size,
                                            m, s, _lib.stream_ptr())
        _lib.check(rc, self.dev_index)
        return out

    def feature_rows(self, views_per_image: list[int], image_sizes: list | None = None) -> list[np.ndarray]:
        """Row of `encode_views`' output for every <image> placeholder of each image, in prompt order.
        LLaVA-1.5: the grid^2 patch rows of the single view ("default" strategy drops CLS, modeling_llava.py:176-178).
        LLaVA-NeXT: base view + spatially arranged, un-padded tile features with one newline row per feature row
        (modeling_llava_next.py pack_image_features)."""
        d = self.d
        out, v0 = [], 0
        n_views_total = sum(views_per_image)
        for i, nv in enumerate(views_per_image):
            base = v0 * d.tokens
            if not d.grid_pinpoints:
                assert nv == 1
                out.append(base + 1 + np.arange(d.grid ** 2, dtype=np.int64))
            else:
                out.append(anyres.packed_rows(tuple(image_sizes[i]), d.grid_pinpoints, d.image_size, d.grid, base, d.tokens,
                                              newline_row=n_views_total * d.tokens))
            v0 += nv
        return out

    def generate_from_features(self, prompts: list, feats: torch.Tensor | None, rows_per_prompt: list, max_new_tokens: int, **kw):
        """prompts[b]: ids with one image_token_id per feature row; rows_per_prompt[b]: concatenated `feature_rows`."""
        return self.generate(prompts, feats, [[] for _ in prompts], max_new_tokens, img_rows=rows_per_prompt, **kw)
