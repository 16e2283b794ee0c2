"""Qwen2-VL on libowc_hip.so: weight packing + the batched image -> greedy-label engine.

This is the MI355X replacement for what the reference's `Qwen2VL.generate_until`
(/root/reference/src/models/_qwen2_vl.py:299-337) delegates to
`Qwen2VLForConditionalGeneration.generate`: vision tower, prompt prefill, greedy decode.
Unlike the reference (batch size 1, `_base.py:103-104`) it batches images: the vision tower and the
prefill run over packed variable-length sequences in chunks, decode runs the whole batch per step.

torch is used for device memory, streams and H2D copies only; all arithmetic is in the HIP library.
"""

from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np
import torch

from .. import _lib, ops
from . import positions

BF16, F32, I32 = torch.bfloat16, torch.float32, torch.int32


@dataclass
class Qwen2VLDims:
    # vision tower
    v_depth: int = 32
    v_embed: int = 1280
    v_heads: int = 16
    v_mlp: int = 5120
    patch_k: int = 1176
    merge: int = 2
    # decoder
    n_layers: int = 28
    d_model: int = 3584
    n_q_heads: int = 28
    n_kv_heads: int = 4
    head_dim: int = 128
    d_ff: int = 18944
    vocab: int = 152064
    tie_embeddings: bool = False
    rms_eps: float = 1e-6
    rope_theta: float = 1e6
    mrope_section: tuple = (16, 24, 24)
    image_token_id: int = 151655
    max_positions: int = 4096      # rope table length (prompt + generated positions)
    max_grid: int = 1024           # vision rope table length (patches per side): covers max_pixels = 1024 * 28 * 28 at the
                                   # processor's 200:1 aspect limit (sqrt(802816 * 200) / 14 = 905); checked per launch
    decoder_dtype: str = "bf16"    # "fp8": decoder projections as e4m3fn weights + per-token e4m3fn activations (config #5)
    # vision tower variant: 0 = Qwen2-VL (LayerNorm, fc1 -> quick_gelu -> fc2), 1 = Qwen2.5-VL (RMSNorm, gated SiLU MLP with
    # biases - `v_mlp` is its intermediate size, 3420 -, window attention except in the `v_fullatt` blocks)
    v_variant: int = 0
    v_window: int = 112
    v_fullatt: tuple = ()


DIMS = {
    # public config.json values
    "qwen2-vl-2b": Qwen2VLDims(n_layers=28, d_model=1536, n_q_heads=12, n_kv_heads=2, d_ff=8960, vocab=151936,
                               tie_embeddings=True),
    "qwen2-vl-7b": Qwen2VLDims(),
    "qwen2-vl-72b": Qwen2VLDims(n_layers=80, d_model=8192, n_q_heads=64, n_kv_heads=8, d_ff=29568, vocab=152064),
    # Qwen2.5-VL (the reference's registry names qwen2.5-vl-7b / -3b, src/models/_qwen2_vl.py:635-648; public config.json values)
    "qwen2.5-vl-7b": Qwen2VLDims(v_variant=1, v_mlp=3420, v_fullatt=(7, 15, 23, 31)),
    "qwen2.5-vl-3b": Qwen2VLDims(v_variant=1, v_mlp=3420, v_fullatt=(7, 15, 23, 31), n_layers=36, d_model=2048, n_q_heads=16, n_kv_heads=2,
                                 d_ff=11008, vocab=151936, tie_embeddings=True),
    "tiny25": Qwen2VLDims(v_variant=1, v_depth=3, v_embed=160, v_heads=2, v_mlp=420, v_fullatt=(1,), n_layers=2, d_model=256, n_q_heads=2,
                          n_kv_heads=1, d_ff=512, vocab=512, tie_embeddings=False, max_positions=2048, max_grid=128),
    # structure-preserving miniature for tests / smoke runs (GQA, head_dim 128, vision head_dim 80)
    "tiny": Qwen2VLDims(v_depth=2, v_embed=160, v_heads=2, v_mlp=640, n_layers=2, d_model=256, n_q_heads=2, n_kv_heads=1,
                        d_ff=512, vocab=512, tie_embeddings=False, max_positions=2048, max_grid=128),
}


def vision_qkv_row_permutation(embed: int, heads: int) -> torch.Tensor:
    """Row order of the fused vision qkv weight that owc_vit_forward expects: inside every q and k head the
    rotary partners (j, j + hd/2) become adjacent rows (2j, 2j+1); v rows are untouched."""
    hd = embed // heads
    j = torch.arange(hd // 2)
    head = torch.stack([j, j + hd // 2], dim=1).reshape(-1)            # [0, hd/2, 1, hd/2+1, ...]
    qk = (torch.arange(2 * heads)[:, None] * hd + head[None, :]).reshape(-1)
    return torch.cat([qk, torch.arange(2 * embed, 3 * embed)])


def pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    """Zero rows appended up to `rows` (first dim)."""
    if t.shape[0] == rows:
        return t
    return torch.cat([t, torch.zeros((rows - t.shape[0], *t.shape[1:]), dtype=t.dtype, device=t.device)], 0)


def interleave_gate_up(gate: torch.Tensor, up: torch.Tensor) -> torch.Tensor:
    """Row layout the SWIGLU GEMM epilogue expects: [g0..g15, u0..u15, g16..g31, u16..u31, ...]."""
    f, k = gate.shape
    assert f % 16 == 0
    return torch.stack([gate.view(f // 16, 16, k), up.view(f // 16, 16, k)], dim=1).reshape(2 * f, k).contiguous()


class Qwen2VLWeights:
    """Device-resident bf16 weights + the ctypes structs the C ABI takes (include/owc.h)."""

    def __init__(self, dims: Qwen2VLDims, device: torch.device):
        self.dims = dims
        self.device = device
        self._keep: list = []  # tensors referenced by raw pointers
        self.vit = _lib.VitWeights()
        self.llm = _lib.LlmWeights()

    # -- construction ------------------------------------------------------------------
    @classmethod
    def from_state_dict(cls, dims: Qwen2VLDims, sd, device) -> "Qwen2VLWeights":
        """`sd`: mapping HF parameter name -> tensor/ndarray (e.g. safetensors shards or a recipe)."""
        self = cls(dims, torch.device(device))

        def get(name):
            t = sd[name]
            if isinstance(t, np.ndarray):
                t = torch.from_numpy(np.ascontiguousarray(t))
            return t.to(device=self.device, dtype=BF16).contiguous()

        self._build(get)
        return self

    @classmethod
    def random(cls, dims: Qwen2VLDims, device, seed: int = 1234) -> "Qwen2VLWeights":
        """Seeded N(0, 0.02)-style synthetic weights generated directly in HBM (benchmarks: no checkpoint offline)."""
        self = cls(dims, torch.device(device))
        self._build(lambda name: random_param(dims, name, self.device, seed))
        return self

    def _k(self, t: torch.Tensor) -> int:
        self._keep.append(t)
        return t.data_ptr()

    def _build(self, get) -> None:
        d = self.dims
        V, T = "model.visual.", "model.language_model."
        hd_v = d.v_embed // d.v_heads
        # ---- vision
        vl = (_lib.VitLayer * d.v_depth)()
        perm = vision_qkv_row_permutation(d.v_embed, d.v_heads).to(self.device)
        v25 = d.v_variant == 1
        f_pad = (d.v_mlp + 127) // 128 * 128 if v25 else d.v_mlp   # 3420 -> 3456: zero gate / up rows meet zero down columns
        for i in range(d.v_depth):
            p = f"{V}blocks.{i}."
            vl[i].qkv_w = self._k(get(p + "attn.qkv.weight").index_select(0, perm).contiguous())
            vl[i].qkv_b = self._k(get(p + "attn.qkv.bias").index_select(0, perm).contiguous())
            if v25:   # RMSNorm (no bias), gated MLP with biases: gate / up rows interleaved for the SwiGLU epilogue
                for f, n in (("ln1_w", "norm1.weight"), ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                             ("ln2_w", "norm2.weight"), ("fc2_b", "mlp.down_proj.bias")):
                    setattr(vl[i], f, self._k(get(p + n)))
                vl[i].fc1_w = self._k(interleave_gate_up(pad_rows(get(p + "mlp.gate_proj.weight"), f_pad),
                                                         pad_rows(get(p + "mlp.up_proj.weight"), f_pad)))
                vl[i].fc1_b = self._k(interleave_gate_up(pad_rows(get(p + "mlp.gate_proj.bias")[:, None], f_pad),
                                                         pad_rows(get(p + "mlp.up_proj.bias")[:, None], f_pad)).reshape(-1).contiguous())
                vl[i].fc2_w = self._k(pad_rows(get(p + "mlp.down_proj.weight").t().contiguous(), f_pad).t().contiguous())
                continue
            for f, n in (("ln1_w", "norm1.weight"), ("ln1_b", "norm1.bias"),
                         ("proj_w", "attn.proj.weight"), ("proj_b", "attn.proj.bias"),
                         ("ln2_w", "norm2.weight"), ("ln2_b", "norm2.bias"), ("fc1_w", "mlp.fc1.weight"),
                         ("fc1_b", "mlp.fc1.bias"), ("fc2_w", "mlp.fc2.weight"), ("fc2_b", "mlp.fc2.bias")):
                setattr(vl[i], f, self._k(get(p + n)))
        self._vit_layers = vl
        v = self.vit
        v.depth, v.embed_dim, v.num_heads, v.mlp_hidden = d.v_depth, d.v_embed, d.v_heads, f_pad
        v.patch_k, v.out_dim, v.merge_unit, v.ln_eps = d.patch_k, d.d_model, d.merge ** 2, 1e-6
        v.patch_w = self._k(get(V + "patch_embed.proj.weight").reshape(d.v_embed, d.patch_k).contiguous())
        v.layers = C.cast(vl, C.POINTER(_lib.VitLayer))
        v.variant, v.fullatt_mask = d.v_variant, sum(1 << int(i) for i in d.v_fullatt)
        for f, n in (("merger_ln_w", "merger.ln_q.weight"), ("merger_ln_b", "merger.ln_q.bias"),
                     ("merger_fc1_w", "merger.mlp.0.weight"), ("merger_fc1_b", "merger.mlp.0.bias"),
                     ("merger_fc2_w", "merger.mlp.2.weight"), ("merger_fc2_b", "merger.mlp.2.bias")):
            if v25 and f == "merger_ln_b":
                continue   # RMSNorm
            setattr(v, f, self._k(get(V + n)))
        vc, vs = ops.rope_table(d.max_grid, hd_v // 4, hd_v // 2, 10000.0, False, self.device)
        v.rope_cos, v.rope_sin, v.rope_positions = self._k(vc), self._k(vs), d.max_grid
        # ---- decoder
        self._build_llm(get, T, qkv_bias=True)

    def _build_llm(self, get, T: str, *, qkv_bias: bool) -> None:
        """Decoder weights -> owc_llm_weights.  `T`: HF prefix of the text model; Llama-family decoders
        (LLaVA) have no q/k/v biases (qkv_b stays NULL)."""
        d = self.dims
        fp8 = getattr(d, "decoder_dtype", "bf16") == "fp8"
        if getattr(d, "decoder_dtype", "bf16") not in ("bf16", "fp8"):
            raise ValueError("decoder_dtype must be 'bf16' or 'fp8'")
        ll = (_lib.LlmLayer * d.n_layers)()
        # fp8: the 256x256 ping-pong kernel walks K-tiles of 128 in pairs, so the down projection's K = d_ff is zero-padded to a
        # multiple of 256 (72B: 29568 = 231 tiles -> 29696): zero gate / up rows give silu(0) * 0 = 0, which quantises to code 0 and
        # meets zero columns of down_proj - the same values, exactly
        ff = (d.d_ff + 255) // 256 * 256 if fp8 else d.d_ff
        for i in range(d.n_layers):
            p = f"{T}layers.{i}."
            qkv_w = torch.cat([get(p + "self_attn.q_proj.weight"), get(p + "self_attn.k_proj.weight"),
                               get(p + "self_attn.v_proj.weight")], 0).contiguous()
            gu = interleave_gate_up(pad_rows(get(p + "mlp.gate_proj.weight"), ff), pad_rows(get(p + "mlp.up_proj.weight"), ff))
            ll[i].ln1_w = self._k(get(p + "input_layernorm.weight"))
            o_w, down_w = get(p + "self_attn.o_proj.weight"), get(p + "mlp.down_proj.weight")
            if ff != d.d_ff:
                down_w = pad_rows(down_w.t().contiguous(), ff).t().contiguous()
            if fp8:  # per-output-channel e4m3fn codes + float scales; the bf16 tensors are dropped right away
                (qkv_w, ll[i].qkv_s), (o_w, ll[i].o_s) = self._q8(qkv_w), self._q8(o_w)
                (gu, ll[i].gateup_s), (down_w, ll[i].down_s) = self._q8(gu), self._q8(down_w)
            ll[i].qkv_w = self._k(qkv_w)
            if qkv_bias:
                ll[i].qkv_b = self._k(torch.cat([get(p + "self_attn.q_proj.bias"), get(p + "self_attn.k_proj.bias"),
                                                 get(p + "self_attn.v_proj.bias")], 0).contiguous())
            ll[i].o_w = self._k(o_w)
            ll[i].ln2_w = self._k(get(p + "post_attention_layernorm.weight"))
            ll[i].gateup_w = self._k(gu)
            ll[i].down_w = self._k(down_w)
        self._llm_layers = ll
        m = self.llm
        m.n_layers, m.d_model, m.n_q_heads, m.n_kv_heads = d.n_layers, d.d_model, d.n_q_heads, d.n_kv_heads
        m.head_dim, m.d_ff, m.vocab = d.head_dim, ff, d.vocab
        m.mrope_sec0, m.mrope_sec1, m.rms_eps = d.mrope_section[0], d.mrope_section[1], d.rms_eps
        self.embed = get(T + "embed_tokens.weight")
        m.embed = self._k(self.embed)
        m.layers = C.cast(ll, C.POINTER(_lib.LlmLayer))
        m.final_norm_w = self._k(get(T + "norm.weight"))
        m.lm_head_w = m.embed if d.tie_embeddings else self._k(get("lm_head.weight"))
        lc, ls = ops.rope_table(d.max_positions, d.head_dim // 2, d.head_dim, d.rope_theta, True, self.device)
        m.rope_cos, m.rope_sin, m.rope_positions = self._k(lc), self._k(ls), d.max_positions
        m.weight_dtype = _lib.WEIGHTS_FP8 if fp8 else _lib.WEIGHTS_BF16

    def _q8(self, w_bf16: torch.Tensor):
        """bf16 [N, K] -> (e4m3fn codes uint8 [N, K], pointer of the float32 row scales) via owc_quantize_rows_fp8."""
        codes, scales = ops.quantize_rows_fp8(w_bf16)
        return codes, self._k(scales)

    def nbytes(self) -> int:
        seen, total = set(), 0
        for t in self._keep:
            if t.data_ptr() not in seen:
                seen.add(t.data_ptr())
                total += t.numel() * t.element_size()
        return total


def random_param(dims: Qwen2VLDims, name: str, device, seed: int = 1234) -> torch.Tensor:
    """The synthetic value of HF parameter `name` (HF layout, bf16): a function of (seed, name) only, so the same tensor can be
    regenerated anywhere - `Qwen2VLWeights.random` packs them for the HIP engine, bench.py's CPU baseline loads the very same
    tensors into HF's Qwen2VLForConditionalGeneration."""
    import zlib

    gen = torch.Generator(device=device)
    gen.manual_seed((seed * 1000003 + zlib.crc32(name.encode())) & 0x7FFFFFFFFFFF)
    shape = _param_shape(dims, name)
    if name.endswith("bias"):
        t = torch.randn(shape, generator=gen, device=device, dtype=F32) * 0.02
    elif "norm" in name or "ln_q" in name:
        t = 1.0 + torch.randn(shape, generator=gen, device=device, dtype=F32) * 0.02
    elif "embed_tokens" in name:
        t = torch.randn(shape, generator=gen, device=device, dtype=BF16) * 0.05
    else:
        fan_in = int(np.prod(shape[1:]))
        t = torch.randn(shape, generator=gen, device=device, dtype=BF16) * (1.0 / np.sqrt(fan_in))
    return t.to(BF16).contiguous()


def hf_param_names(d: Qwen2VLDims) -> list[str]:
    """Every HF parameter name of the architecture (transformers 5.x naming), i.e. the keys `random_param` is defined on."""
    V, T = "model.visual.", "model.language_model."
    names = [V + "patch_embed.proj.weight"]
    block = ("norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight",
             "attn.proj.bias", "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias")
    if d.v_variant == 1:
        block = ("norm1.weight", "norm2.weight", "attn.qkv.weight", "attn.qkv.bias", "attn.proj.weight", "attn.proj.bias",
                 "mlp.gate_proj.weight", "mlp.gate_proj.bias", "mlp.up_proj.weight", "mlp.up_proj.bias", "mlp.down_proj.weight",
                 "mlp.down_proj.bias")
    for i in range(d.v_depth):
        names += [f"{V}blocks.{i}.{n}" for n in block]
    names += [V + n for n in ("merger.ln_q.weight", "merger.ln_q.bias", "merger.mlp.0.weight", "merger.mlp.0.bias",
                              "merger.mlp.2.weight", "merger.mlp.2.bias") if not (d.v_variant == 1 and n == "merger.ln_q.bias")]
    names.append(T + "embed_tokens.weight")
    for i in range(d.n_layers):
        names += [f"{T}layers.{i}.{n}" for n in ("self_attn.q_proj.weight", "self_attn.q_proj.bias", "self_attn.k_proj.weight",
                                                 "self_attn.k_proj.bias", "self_attn.v_proj.weight", "self_attn.v_proj.bias",
                                                 "self_attn.o_proj.weight", "mlp.gate_proj.weight", "mlp.up_proj.weight",
                                                 "mlp.down_proj.weight", "input_layernorm.weight", "post_attention_layernorm.weight")]
    names.append(T + "norm.weight")
    if not d.tie_embeddings:
        names.append("lm_head.weight")
    return names


def _param_shape(d: Qwen2VLDims, name: str) -> tuple:
    E, F, hd = d.v_embed, d.v_mlp, d.head_dim
    E4 = E * d.merge ** 2
    tail = name.split(".", 4)[-1] if "blocks." in name or "layers." in name else name
    table = {
        "model.visual.patch_embed.proj.weight": (E, d.patch_k),
        "norm1.weight": (E,), "norm1.bias": (E,), "norm2.weight": (E,), "norm2.bias": (E,),
        "attn.qkv.weight": (3 * E, E), "attn.qkv.bias": (3 * E,), "attn.proj.weight": (E, E), "attn.proj.bias": (E,),
        "mlp.fc1.weight": (F, E), "mlp.fc1.bias": (F,), "mlp.fc2.weight": (E, F), "mlp.fc2.bias": (E,),
        "model.visual.merger.ln_q.weight": (E,), "model.visual.merger.ln_q.bias": (E,),
        "model.visual.merger.mlp.0.weight": (E4, E4), "model.visual.merger.mlp.0.bias": (E4,),
        "model.visual.merger.mlp.2.weight": (d.d_model, E4), "model.visual.merger.mlp.2.bias": (d.d_model,),
        "model.language_model.embed_tokens.weight": (d.vocab, d.d_model),
        "self_attn.q_proj.weight": (d.n_q_heads * hd, d.d_model), "self_attn.q_proj.bias": (d.n_q_heads * hd,),
        "self_attn.k_proj.weight": (d.n_kv_heads * hd, d.d_model), "self_attn.k_proj.bias": (d.n_kv_heads * hd,),
        "self_attn.v_proj.weight": (d.n_kv_heads * hd, d.d_model), "self_attn.v_proj.bias": (d.n_kv_heads * hd,),
        "self_attn.o_proj.weight": (d.d_model, d.n_q_heads * hd),
        "mlp.gate_proj.weight": (d.d_ff, d.d_model), "mlp.up_proj.weight": (d.d_ff, d.d_model),
        "mlp.down_proj.weight": (d.d_model, d.d_ff),
        "input_layernorm.weight": (d.d_model,), "post_attention_layernorm.weight": (d.d_model,),
        "model.language_model.norm.weight": (d.d_model,), "lm_head.weight": (d.vocab, d.d_model),
    }
    if name in table:
        return table[name]
    if "visual.blocks." in name:
        tail_v = name.split(".", 4)[-1]
        if tail_v.startswith("mlp.") and "proj" in tail_v:   # the gated vision MLP of Qwen2.5-VL
            return {"mlp.gate_proj.weight": (F, E), "mlp.gate_proj.bias": (F,), "mlp.up_proj.weight": (F, E), "mlp.up_proj.bias": (F,),
                    "mlp.down_proj.weight": (E, F), "mlp.down_proj.bias": (E,)}[tail_v]
        return table[tail_v]
    if "language_model.layers." in name:
        return table[name.split(".", 4)[-1]]
    raise KeyError(name)


class Qwen2VLEngine:
    """Batched open-world classification forward: pixel_values + prompt ids -> greedy token ids."""

    def __init__(self, weights: Qwen2VLWeights, *, vit_chunk_tokens: int = 131072, prefill_chunk_tokens: int = 65536,
                 share_prefix: bool = True, min_shared_prefix: int = 4, graph_decode: bool = False, graph_max_batch: int = 64):
        # replay decode steps of small batches as one captured hipGraph; off by default: measured +-0 on MI355X (4.32 vs 4.25 ms
        # per 7B token-step at batch 1 - the ~250 launches of a step are enqueued ahead of the GPU either way)
        self.graph_decode = graph_decode
        self.graph_max_batch = graph_max_batch
        self.share_prefix = share_prefix            # prefill the prompts' common leading text tokens once per chunk
        self.min_shared_prefix = min_shared_prefix
        self.w = weights
        self.d = weights.dims
        self.device = weights.device
        self.dev_index = self.device.index or 0
        self.vit_chunk_tokens = vit_chunk_tokens
        self.prefill_chunk_tokens = prefill_chunk_tokens
        self._ws: torch.Tensor | None = None
        self._kv: tuple | None = None      # the engine's grow-only K / V cache pair (reserve_kv)
        self._lib = _lib.load()
        self._ctx = _lib.ctx(self.dev_index)
        # HF's repetition penalty for every generation of this engine (1.0 = off): a model plug-in sets it from the checkpoint's
        # generation_config.json (`Qwen2VL.load_model`); `generate(..., repetition_penalty=...)` overrides it per call
        self.repetition_penalty = 1.0

    # -- helpers -----------------------------------------------------------------------
    def _workspace(self, nbytes: int) -> torch.Tensor:
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def _i32(self, a) -> torch.Tensor:
        return _lib.h2d(a, self.device, np.int32)

    def reserve_kv(self, elems: int) -> tuple[torch.Tensor, torch.Tensor]:
        """The engine's K and V cache blocks (bf16, `elems` each): ONE grow-only pair per engine, handed out as views.
        Passes of a task differ in size (the adaptive ramp of `generate_until`, carried sequences, generation lengths); a fresh
        `torch.empty` per pass made every new size a fresh hipMalloc beside the cached blocks of the earlier sizes - seconds per
        pass once the blocks are tens of GB, and out-of-memory retries for an MHA decoder (LLaVA-1.5: 0.5 MB of KV per token).
        Work is stream-ordered, so the next pass may reuse the memory while the previous one is still queued.  On growth the
        old pair goes back to the driver first (`empty_cache`), so the peak is the new pair, not the sum."""
        kv = self._kv
        if kv is None or kv[0].numel() < elems:
            self._kv = kv = None
            if elems * 2 > (1 << 30):
                torch.cuda.empty_cache()
            self._kv = kv = (torch.empty(int(elems), dtype=BF16, device=self.device), torch.empty(int(elems), dtype=BF16, device=self.device))
        return kv[0][:elems], kv[1][:elems]

    def held_bytes(self) -> int:
        """Device bytes this engine keeps between passes and reuses (the K / V pair and the workspace): memory a batch-size decision
        must count as FREE, or the batch of a later task would depend on what an earlier one left allocated (`engine_batch`)."""
        kv = 2 * self._kv[0].numel() * self._kv[0].element_size() if self._kv is not None else 0
        return kv + (self._ws.numel() if self._ws is not None else 0)

    def release_kv(self) -> None:
        """Hand the K / V pair (tens of GB after a large pass) back: for a caller that is done generating and needs the memory for
        something else (another model on the same GPU).  The next pass simply reserves a new pair."""
        self._kv = None

    # -- vision tower ------------------------------------------------------------------
    def encode_images(self, pixel_values: torch.Tensor, grid_thw) -> torch.Tensor:
        """pixel_values [sum(t*h*w), 1176] bf16 (device) -> merged embeddings [sum/4, d_model] bf16."""
        assert pixel_values.dtype == BF16 and pixel_values.is_cuda and pixel_values.stride(1) == 1
        grid = [tuple(int(x) for x in g) for g in grid_thw]
        lens = [t * h * w for t, h, w in grid]
        total = sum(lens)
        assert total == pixel_values.shape[0]
        mu = self.d.merge ** 2
        out = torch.empty((total // mu, self.d.d_model), dtype=BF16, device=self.device)
        i0 = 0
        while i0 < len(grid):  # chunk whole images
            i1, tok = i0, 0
            while i1 < len(grid) and (tok == 0 or tok + lens[i1] <= self.vit_chunk_tokens):
                tok += lens[i1]
                i1 += 1
            row0 = sum(lens[:i0])
            self._vit_chunk(pixel_values[row0:row0 + tok], grid[i0:i1], lens[i0:i1], out[row0 // mu:(row0 + tok) // mu])
            i0 = i1
        return out

    def _vit_chunk(self, pix, grid, lens, out) -> None:
        T = pix.shape[0]
        max_side = max(max(h, w) for _, h, w in grid)
        if max_side > self.d.max_grid:   # the rotary table is indexed by patch coordinates (smart_resize allows aspect 200:1)
            raise ValueError(f"an image grid side of {max_side} patches exceeds the vision rotary table "
                             f"(Qwen2VLDims.max_grid = {self.d.max_grid})")
        hw = positions.vision_pos_hw(grid, self.d.merge)
        starts = np.concatenate([[0], np.cumsum(lens)[:-1]])
        seq_start, seq_len = self._i32(starts), self._i32(lens)
        nbytes = self._lib.owc_vit_workspace_bytes(C.byref(self.w.vit), T)
        ws = self._workspace(nbytes)
        if self.d.v_variant == 1:   # Qwen2.5-VL: window order (an image's rows stay contiguous), windows of <= 64 patches
            tok_index, out_index, win_start, win_len = positions.vision_windows(grid, self.d.merge, self.d.v_window, 14)
            # (named tensors, not temporaries: a temporary is released to torch's caching allocator as soon as its data_ptr() has
            # been taken, and the next upload would land in the same bytes)
            t_hw, t_tok, t_out = self._i32(hw[tok_index]), self._i32(tok_index), self._i32(out_index)
            t_ws, t_wl = self._i32(win_start), self._i32(win_len)
            rc = self._lib.owc_vit25_forward(self._ctx, C.byref(self.w.vit), pix.data_ptr(), pix.stride(0), t_hw.data_ptr(),
                                             t_tok.data_ptr(), t_out.data_ptr(), seq_start.data_ptr(), seq_len.data_ptr(), len(grid),
                                             max(lens), t_ws.data_ptr(), t_wl.data_ptr(), len(win_len), int(win_len.max()), T,
                                             max_side, out.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, self.dev_index)
            return
        pos_hw = self._i32(hw)
        rc = self._lib.owc_vit_forward(self._ctx, C.byref(self.w.vit), pix.data_ptr(), pix.stride(0), pos_hw.data_ptr(),
                                       seq_start.data_ptr(), seq_len.data_ptr(), len(grid), T, max(lens), max_side, out.data_ptr(),
                                       ws.data_ptr(), ws.numel(), _lib.stream_ptr())
        _lib.check(rc, self.dev_index)

    # -- decoder -----------------------------------------------------------------------
    def generate(self, *args, **kw):
        """`_generate_impl` (its docstring is the interface) with the context's repetition-penalty option cleared on every exit path."""
        try:
            return self._generate_impl(*args, **kw)
        finally:
            self._lib.owc_llm_set_repetition_penalty(self._ctx, 1.0, None, 0)

    def _generate_impl(self, prompts: list, img_embeds: torch.Tensor | None, grids_per_prompt: list, max_new_tokens: int,
                 *, eos_token_id: int = -1, pad_token_id: int = 0, stop_check_every: int = 1, compact_rows: bool = True,
                 return_logits: bool = False, img_rows: list | None = None, forced_tokens=None,
                 return_step_logits: bool = False, stats: dict | None = None, sampling: dict | None = None,
                 carry: dict | None = None, repetition_penalty: float | None = None):
        """Greedy generation for a batch of prompts.

        `repetition_penalty` (default: the engine's `self.repetition_penalty`, which a model plug-in sets from the checkpoint's
        generation_config.json - HF merges that file into every `generate` call the reference makes, greedy ones included,
        /root/reference/src/models/_qwen2_vl.py:319-329): HF's RepetitionPenaltyLogitsProcessor - before the argmax (or the draw)
        every token id of the sequence's prompt (image placeholders included) and everything fed to it since gets
        logit < 0 ? logit * p : logit / p in fp32 (`owc_llm_set_repetition_penalty`, include/owc.h).  1.0 = off.  The penalty is a
        function of the sequence's own history, so batch invariance holds; straggler hand-over between passes is switched off while
        it is on (a pass then runs every sequence to its end: the seen-token bitmap lives with the pass's cache slots).

        prompts[b]: 1-D int array of token ids holding image_token_id placeholders;
        grids_per_prompt[b]: list of (t, h, w) for that prompt's images, in order;
        img_embeds: rows for all image tokens of all prompts, in prompt order;
        img_rows[b] (optional): explicit row of `img_embeds` for every image token of prompt b (engines whose
        feature buffer is not already in token order, e.g. LLaVA's CLS-skipping / anyres packing).
        forced_tokens (optional, int [B, max_new_tokens]): teacher forcing - the token FED at step j+1 is
        forced_tokens[b][j] instead of the engine's own argmax (which is still what the returned tokens hold), so every
        step's logits are conditional on the reference's continuation.  With EOS handling on, a sequence ends where its
        FORCED continuation holds EOS (seeded answer lengths for the ragged-length bench leg and the compaction tests).
        EOS handling (eos_token_id >= 0): the reference runs one `generate` per image, each stopping at its own EOS
        (/root/reference/src/models/_qwen2_vl.py:319-337); here the batch decodes together, the done flags are read back
        every `stop_check_every` steps one step behind the GPU (no stall), and with `compact_rows` the finished rows are
        DROPPED from the following steps (as soon as >= 1/64 of the live rows are done): GEMM M and the attention grid shrink,
        the KV cache stays in place (slot indirection), tokens land in their original rows.  Every kernel computes a row
        independently of its neighbours, so the tokens equal the uncompacted run bit for bit (tested).
        `sampling` (optional): {"temperature": T > 0, "top_k": int (0 off), "top_p": float (None / >= 1 off), "seed": int,
        "stream_ids": int per prompt (default: its index)} - HF's `do_sample` path (temperature -> top-k -> top-p -> multinomial) on
        the library's documented Philox stream; a sequence's draws depend on (seed, its stream id, step) only.
        `stats` (optional dict) receives the live-row count of every step.
        `carry` (optional dict; EOS handling on): straggler hand-over between consecutive passes of a task.  In:
        `carry["in"]` - the unfinished sequences of the previous pass (`carry["out"]` of that call), which join this pass's decode
        steps as extra rows with their own KV rows, pending token and remaining budget; `carry["below"]` - once this pass's OWN
        live rows are at most this many (and at least one decode step ran) the loop stops and the still-running sequences are
        exported (0: run everything to the end - the last pass of a task); `carry["tags"]` - one identity per prompt;
        `carry["slots"]` (optional) - cache slots kept for carried sequences beside this pass's own.  Out:
        `carry["finished"]` = [(tag, int32 tokens incl. pad)] of carried-in sequences that ended here, `carry["out"]` = the
        export (None when nothing is left), `carry["unfinished_rows"]` = the own rows inside it (their rows of the returned
        tensor are incomplete).  A sequence's tokens do not depend on which pass finishes it (batch invariance; tested).
        Returns int32 [B, max_new_tokens] (pad after EOS) and, optionally, the first-step logits [B, vocab]
        (`return_logits`) or every step's logits [max_new_tokens, B, vocab] (`return_step_logits`).
        """
        d = self.d
        B = len(prompts)
        lens = np.array([len(p) for p in prompts], dtype=np.int64)
        s_max = int(lens.max()) + max_new_tokens
        Hkv, G = d.n_kv_heads, d.n_q_heads // d.n_kv_heads
        rep = float(self.repetition_penalty if repetition_penalty is None else repetition_penalty)
        if not rep > 0:
            raise ValueError("repetition_penalty must be > 0 (1.0 = off)")
        cin = carry.get("in") if carry is not None else None
        if rep != 1.0 and carry is not None:
            if cin is not None and len(cin["tags"]):
                raise ValueError("a carried-in sequence cannot join a pass that runs with a repetition penalty")
            carry = {**carry, "below": 0, "_caller": carry}   # no hand-over: every sequence ends in this pass
        if carry is not None and (eos_token_id < 0 or return_step_logits or return_logits or self.graph_decode):
            raise ValueError("carry needs EOS handling on and no logits output")
        NC = 0 if cin is None else len(cin["tags"])
        Bx = B + NC                                  # rows of the decode batch: this pass's prompts + the carried-in sequences
        T_out = max_new_tokens                       # width of the token buffer = steps this pass may run + 1
        if NC:
            rem = np.asarray(cin["remaining"], dtype=np.int64)             # tokens each carried sequence may still emit
            s_max = max(s_max, int((np.asarray(cin["cached"]) + rem).max()) + 2)   # (+2: a row runs <= 2 steps past its budget -
            if not carry.get("below"):                                     #  the decode loop below compacts it away in time)
                T_out = max(T_out, int(rem.max()) + 1)                     # the last pass runs every sequence to its end
            if not compact_rows:   # nothing drops a finished row: every row takes all T_out - 1 steps and writes a K / V row each
                s_max = max(s_max, int(max(lens.max(), np.asarray(cin["cached"]).max())) + T_out)
        n_slots = Bx
        if carry is not None:
            # the passes of a task should ask the allocator for the SAME two blocks: a 20 GB cache that grew by 1 % is a fresh
            # hipMalloc (~0.5 s each); slots for carried sequences = what the caller reserved (`carry["slots"]`), else up to
            # B / 8 (at least 256); key rows in steps of 16
            n_slots = B + (max(NC, int(carry["slots"])) if carry.get("slots") else (max(NC, B // 8, 256) + 255) // 256 * 256)
            s_max = (max(s_max, int(lens.max()) + max_new_tokens + 2) + 15) // 16 * 16
        cache_elems = d.n_layers * n_slots * Hkv * s_max * d.head_dim
        kc, vc = self.reserve_kv(cache_elems)
        cache = _lib.KvCache(kc.data_ptr(), vc.data_ptr(), n_slots, s_max)
        if NC:   # the carried sequences' K / V rows move into slots B.. of this pass's cache (one strided copy each)
            w_c = cin["k"].shape[3]
            kc.view(d.n_layers, n_slots, Hkv, s_max, d.head_dim)[:, B:Bx, :, :w_c].copy_(cin["k"])
            vc.view(d.n_layers, n_slots, Hkv, s_max, d.head_dim)[:, B:Bx, :, :w_c].copy_(cin["v"])

        # positions (host integer bookkeeping)
        pos_list, max_pos, img_index = self._positions_and_image_rows(prompts, grids_per_prompt, img_embeds, img_rows)
        if int(max_pos.max()) + max_new_tokens + 1 > d.max_positions:
            raise ValueError("prompt + generation exceeds the rope table (raise Qwen2VLDims.max_positions)")

        # per-row decode state, two sets (a row compaction gathers from one into the other):
        # [set][fed token, rope position, cache write index, key count, cache slot, key start, output row][B] + the done flags
        ar = np.arange(Bx, dtype=np.int64)
        st_host = np.zeros((7, Bx), np.int64)
        st_host[1, :B], st_host[2, :B], st_host[3, :B] = max_pos + 1, lens, lens + 1
        if NC:   # a carried sequence continues where it stood: pending token, its rope position, its keys
            st_host[0, B:], st_host[1, B:] = cin["tok"], cin["pos"]
            st_host[2, B:], st_host[3, B:] = cin["cached"], np.asarray(cin["cached"]) + 1
        st_host[4], st_host[5], st_host[6] = ar, ar * Hkv * s_max, ar
        state = torch.empty((2, 7, Bx), dtype=I32, device=self.device)
        state[0].copy_(self._i32(st_host.astype(np.int32)))
        done2 = torch.zeros((2, Bx), dtype=torch.uint8, device=self.device)
        cur = 0
        next_tok = state[0, 0]
        step_logits = torch.empty((max_new_tokens, B, d.vocab), dtype=BF16, device=self.device) if return_step_logits else None
        first_logits = step_logits[0] if return_step_logits else (
            torch.empty((B, d.vocab), dtype=BF16, device=self.device) if return_logits else None)
        forced = None
        if forced_tokens is not None:
            f_host = np.zeros((T_out, Bx), np.int64)
            f_host[:max_new_tokens, :B] = np.asarray(forced_tokens).reshape(B, max_new_tokens).T   # [T, B]: one contiguous row per step
            for i in range(NC):   # a carried sequence writes its true column c0 + j - 1 at this pass's step j
                fr = np.asarray(cin["forced_rest"][i])[: T_out - 1]
                f_host[1:1 + len(fr), B + i] = fr
            forced = self._i32(f_host)
        samp, samp_ref = None, None
        if sampling is not None:
            if not float(sampling["temperature"]) > 0:
                raise ValueError("sampling needs temperature > 0 (temperature 0 is greedy: pass sampling=None)")
            sid = np.zeros(Bx, np.int64)
            sid[:B] = np.asarray(sampling.get("stream_ids", np.arange(B)), dtype=np.int64).reshape(B)
            soff = np.zeros(Bx, np.int64)
            if NC:   # a carried sequence keeps its stream and goes on counting ITS steps: pass-local step j is its step c0 + j - 1
                sid[B:], soff[B:] = cin["stream"], np.asarray(cin["emitted"]) - 1
            streams, step_off = self._i32(sid), self._i32(soff)
            top_p = sampling.get("top_p")
            samp = _lib.Sampling(float(sampling["temperature"]), int(sampling.get("top_k") or 0), float(top_p) if top_p else 0.0,
                                 int(sampling.get("seed", 0)) & 0xFFFFFFFFFFFFFFFF, streams.data_ptr(), step_off.data_ptr())
            samp_ref = C.byref(samp)

        seen = None
        if rep != 1.0:   # one bitmap row of the vocabulary per cache slot: which ids the sequence has seen (prompt + fed tokens)
            wpr = (d.vocab + 31) // 32
            seen = torch.zeros((n_slots, wpr), dtype=I32, device=self.device)
            _lib.check(self._lib.owc_llm_set_repetition_penalty(self._ctx, rep, seen.data_ptr(), wpr), self.dev_index)
        # ---- prefill in chunks of whole prompts (chunk size counted in packed ROWS: with a shared prefix every
        # prompt contributes len - P rows, so more prompts fit the same GEMM M)
        p_all = self._common_prefix(prompts, 0, B)
        b0 = 0
        while b0 < B:
            b1, rows = b0, p_all
            while b1 < B and (b1 == b0 or rows + lens[b1] - p_all <= self.prefill_chunk_tokens):
                rows += int(lens[b1]) - p_all
                b1 += 1
            self._prefill_chunk(prompts, pos_list, img_index, img_embeds, lens, b0, b1, cache, next_tok, first_logits, samp_ref)
            b0 = b1

        # ---- greedy decode: the live rows of the batch per step
        if stats is not None:    # (events only: the caller reads `decode_ms` after it has synchronised)
            stats["decode_events"] = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            stats["decode_events"][0].record()
        out_tokens = torch.empty((Bx, T_out), dtype=I32, device=self.device)
        out_tokens.fill_(pad_token_id)              # rows dropped by a compaction keep pad behind their last column
        eos1 = -1
        rc = self._lib.owc_decode_update(self._ctx, next_tok.data_ptr(), done2[0].data_ptr(), out_tokens.data_ptr(),
                                         T_out, 0, B, eos_token_id, eos1, pad_token_id, None,
                                         forced[0].data_ptr() if forced is not None else None, _lib.stream_ptr())
        _lib.check(rc, self.dev_index)
        live_per_step = [Bx]
        steps_run = 0
        row_of = np.arange(Bx)                                     # compact row -> row of out_tokens (host mirror of state[6])
        if T_out > 1:
            q_start = self._i32(ar * (d.n_q_heads + 2 * Hkv))      # functions of the compact row index: prefixes stay valid
            o_start = self._i32(ar * d.n_q_heads)
            q_len = self._i32(np.full(Bx, G))
            nbytes = self._lib.owc_llm_workspace_bytes(C.byref(self.w.llm), Bx, Bx)
            ws = self._workspace(nbytes)
            col = torch.ones(1, dtype=I32, device=self.device)     # the step's output column, advanced on the device
            n = Bx                                                 # live rows
            budget = np.full(Bx, max_new_tokens - 1, np.int64)     # decode steps a row may take in this pass
            if NC:
                budget[B:] = rem

            def step(j):
                v, dn = state[cur], done2[cur]
                rc = self._lib.owc_llm_decode_step(
                    self._ctx, C.byref(self.w.llm), C.byref(cache), v[0].data_ptr(), v[1].data_ptr(), v[4].data_ptr(),
                    v[2].data_ptr(), v[5].data_ptr(), v[3].data_ptr(), q_start.data_ptr(), o_start.data_ptr(), q_len.data_ptr(),
                    dn.data_ptr(), out_tokens.data_ptr(), T_out, j, col.data_ptr(), n, eos_token_id, eos1, pad_token_id,
                    v[6].data_ptr(), forced[j].data_ptr() if forced is not None else None, samp_ref,
                    step_logits[j].data_ptr() if step_logits is not None else None, ws.data_ptr(), ws.numel(),
                    _lib.stream_ptr())
                _lib.check(rc, self.dev_index)

            watch = eos_token_id >= 0 and stop_check_every > 0
            compact = compact_rows and step_logits is None
            if self.graph_decode and carry is None and B <= self.graph_max_batch and max_new_tokens >= 4 and forced is None and step_logits is None:
                # Small batches are launch-bound (~250 tiny launches per step): steps 2.. replay ONE captured hipGraph.  The
                # step keeps its own rope position / write index / key count / output column on the device, so every replay
                # is the same launch sequence with the same arguments (no row compaction on this path).
                step(1)                                             # eager: loads code objects, sets kernel attributes
                graph = torch.cuda.CUDAGraph()
                torch.cuda.synchronize(self.device)
                with torch.cuda.graph(graph):
                    step(0)
                # the capture did not execute anything: replay for steps 2 .. T-1
                for j in range(2, max_new_tokens):
                    graph.replay()
                    live_per_step.append(B)
                    if watch and j % max(stop_check_every, 8) == 0 and int(done2[0].cpu().numpy().min()) == 1:
                        break
                del graph
            else:
                flags = torch.empty(Bx, dtype=torch.uint8).pin_memory() if watch else None
                pending = None                                      # (event, rows, step) of a done-flag snapshot in flight
                below = int(carry.get("below") or 0) if carry is not None else 0
                for j in range(1, T_out):
                    step(j)
                    steps_run = j
                    live_per_step.append(n)
                    if pending is not None:
                        # the flags as of the step BEFORE the one just enqueued: the GPU is busy while the host looks at them
                        pending[0].synchronize()
                        # (a row is also finished when its budget is spent: a carried sequence's cap, or this pass's own)
                        seen = pending[1]                           # rows of the batch when the flags were taken (= n: only this block compacts)
                        alive = np.flatnonzero((flags[:seen].numpy() == 0) & (budget[row_of[:seen]] > pending[2]))
                        pending = None
                        if len(alive) == 0:
                            break
                        if below and int((row_of[alive] < B).sum()) <= below:
                            break                                   # the stragglers go on inside the next pass (exported below)
                        # A finished row stays in the batch until enough of them make a compaction worth its launch, and every
                        # step advances its cache write index and rope position like a live row's.  Its slot holds `s_max` rows
                        # (sized for the LIVE rows' budgets), so a finished row must leave before it writes past its slot into
                        # the next one - with passes of different generation lengths or carried-in sequences T_out exceeds what
                        # an own row's slot was sized for.  The next step (j + 1) writes at w0 + j; the flags are one step old.
                        crowded = False
                        if compact and len(alive) < seen:
                            gone = np.setdiff1d(row_of[:seen], row_of[alive], assume_unique=True)
                            ahead_ = j + 1 + max(1, stop_check_every)             # rows linger until the next look at the flags
                            crowded = bool((st_host[2, gone] + ahead_ >= s_max).any() or (st_host[1, gone] + ahead_ >= d.max_positions).any())
                        if compact and (crowded or n - len(alive) >= max(1, n // 64)):
                            live = self._i32(alive)
                            a, b_ = state[cur], state[cur ^ 1]
                            rc = self._lib.owc_decode_compact(
                                self._ctx, live.data_ptr(), len(alive), *(a[i].data_ptr() for i in (0, 1, 2, 3, 4, 5, 6)),
                                done2[cur].data_ptr(), *(b_[i].data_ptr() for i in (0, 1, 2, 3, 4, 5, 6)),
                                done2[cur ^ 1].data_ptr(), _lib.stream_ptr())
                            _lib.check(rc, self.dev_index)
                            cur ^= 1
                            n = len(alive)
                            row_of = row_of[alive]
                    if watch and j % stop_check_every == 0 and j + 1 < T_out:
                        flags[:n].copy_(done2[cur, :n], non_blocking=True)
                        ev = torch.cuda.Event()
                        ev.record()
                        pending = (ev, n, j)
        if carry is not None:
            self._carry_export(carry, cin, B, NC, n if T_out > 1 else Bx, cur, state, done2, row_of, steps_run, out_tokens, kc, vc,
                               s_max, Bx, max_new_tokens, forced_tokens, pad_token_id, n_slots,
                               None if sampling is None else sid)
            if "_caller" in carry:   # (repetition penalty: the engine ran on a copy with the hand-over off - hand the results back)
                carry["_caller"].update({k: v for k, v in carry.items() if k not in ("_caller", "below")})
            out_tokens = out_tokens[:B, :max_new_tokens]
        if stats is not None:
            stats["live_rows_per_step"] = live_per_step
            stats["decode_events"][1].record()
        if return_step_logits:
            return out_tokens, step_logits
        return (out_tokens, first_logits) if return_logits else out_tokens

    def _positions_and_image_rows(self, prompts, grids_per_prompt, img_embeds, img_rows):
        """Per prompt: its M-RoPE position ids [3, len] (HF get_rope_index), the largest of them, and for every token the row of
        `img_embeds` it takes its embedding from (-1: the token table)."""
        d = self.d
        pos_list, max_pos = [], np.empty(len(prompts), dtype=np.int64)
        img_index = []
        img_cursor = 0
        for b, ids in enumerate(prompts):
            ids = np.asarray(ids)
            if grids_per_prompt[b]:
                p3, _ = positions.mrope_positions(ids, grids_per_prompt[b], d.image_token_id, d.merge)
            else:
                p3 = np.tile(np.arange(len(ids), dtype=np.int32)[None], (3, 1))
            pos_list.append(p3)
            max_pos[b] = int(p3.max())
            is_img = ids == d.image_token_id
            idx = np.full(len(ids), -1, dtype=np.int32)
            n_img = int(is_img.sum())
            if img_rows is not None:
                rows = np.asarray(img_rows[b], dtype=np.int32)
                if len(rows) != n_img or (n_img and (img_embeds is None or rows.min() < 0 or rows.max() >= img_embeds.shape[0])):
                    raise ValueError("image token count does not match the image feature rows")
                idx[is_img] = rows
            else:
                idx[is_img] = np.arange(img_cursor, img_cursor + n_img, dtype=np.int32)
            img_cursor += n_img
            img_index.append(idx)
        if img_rows is None and img_cursor and (img_embeds is None or img_embeds.shape[0] != img_cursor):
            raise ValueError("image token count does not match the image embeddings")
        return pos_list, max_pos, img_index

    def generate_beam(self, prompts: list, img_embeds: torch.Tensor | None, grids_per_prompt: list, max_new_tokens: int, num_beams: int,
                      *, eos_token_id: int, pad_token_id: int = 0, img_rows: list | None = None, length_penalty: float = 1.0,
                      early_stopping=False, return_scores: bool = False):
        """Beam search for a batch of prompts: what the reference asks HF for with `num_beams` > 1 in a request's gen_kwargs
        (/root/reference/src/models/_qwen2_vl.py:308-329, _llava_hf.py:365-376; `do_sample=False`, HF's `length_penalty=1.0`,
        `early_stopping=False`, the best hypothesis returned).  Every prompt is searched on its own, exactly as a batch-size-1
        `generate` call would (`engine/beam.BeamSearcher`: HF's bookkeeping, pinned on HF's own run in tests/test_oracle_beam.py);
        the B x num_beams running hypotheses share the decode steps.

        Device side per step: `owc_llm_decode_step` on the B x num_beams rows (fed token, position, cache slot per row), then
        `owc_beam_candidates` (log-sum-exp + the 2 x num_beams best logits per row) - 8 (1 + 4 num_beams) bytes per row come back.
        The KV cache follows the hypotheses by SLOT: a running beam that continues a parent takes over the parent's slot; when a
        parent is continued more than once, the extra children get a copy of its rows in the slot of a parent nobody continued.
        The prompt is prefilled once per prompt (the num_beams hypotheses start identical: HF gives all but the first a score of
        -1e9 at the first step).  Returns int32 [B, max_new_tokens] (EOS kept, pad behind it) and optionally the scores."""
        from .beam import BeamSearcher

        d = self.d
        B, k = len(prompts), int(num_beams)
        if k < 2:
            raise ValueError("generate_beam is for num_beams >= 2 (one beam is `generate`)")
        if float(self.repetition_penalty) != 1.0:
            raise NotImplementedError("beam search with a repetition penalty (generation_config.json `repetition_penalty` != 1) is not "
                                      "implemented by the HIP decoder; greedy and sampled one-beam generation are")
        if B == 0:
            out = torch.empty((0, max_new_tokens), dtype=I32, device=self.device)
            return (out, np.zeros(0, np.float32)) if return_scores else out
        R = B * k                                   # (decode steps are launched directly: the rows' slots change between steps)
        lens = np.array([len(p) for p in prompts], dtype=np.int64)
        Hkv, G = d.n_kv_heads, d.n_q_heads // d.n_kv_heads
        s_max = (int(lens.max()) + max_new_tokens + 15) // 16 * 16
        kc, vc = self.reserve_kv(d.n_layers * R * Hkv * s_max * d.head_dim)
        cache = _lib.KvCache(kc.data_ptr(), vc.data_ptr(), R, s_max)
        kv5 = (kc.view(d.n_layers, R, Hkv, s_max, d.head_dim), vc.view(d.n_layers, R, Hkv, s_max, d.head_dim))
        pos_list, max_pos, img_index = self._positions_and_image_rows(prompts, grids_per_prompt, img_embeds, img_rows)
        if int(max_pos.max()) + max_new_tokens + 1 > d.max_positions:
            raise ValueError("prompt + generation exceeds the rope table (raise Qwen2VLDims.max_positions)")
        # ---- prefill: prompt b into slot b * k (the slots of a prompt's beams are b * k .. b * k + k - 1)
        first_logits = torch.empty((B, d.vocab), dtype=BF16, device=self.device)
        next_tok = torch.empty(B, dtype=I32, device=self.device)
        share, self.share_prefix = self.share_prefix, False       # (the shared-prefix broadcast writes to CONSECUTIVE slots)
        try:
            b0 = 0
            while b0 < B:   # launch groups of whole prompts, as in `generate`
                b1, rows = b0, 0
                while b1 < B and (b1 == b0 or rows + lens[b1] <= self.prefill_chunk_tokens):
                    rows += int(lens[b1])
                    b1 += 1
                self._prefill_chunk(prompts, pos_list, img_index, img_embeds, lens, b0, b1, cache, next_tok, first_logits, None,
                                    slot_of=lambda i: i * k)
                b0 = b1
        finally:
            self.share_prefix = share
        bs = BeamSearcher(B, k, max_new_tokens, eos_token_id, pad_token_id, length_penalty, early_stopping)
        K2 = 2 * k
        # step 0: every beam of a prompt sees the prefill's logits
        logz, tv, ti = ops.beam_candidates(first_logits, K2)
        logz, tv, ti = (x.cpu().numpy() for x in (logz, tv, ti))
        parent, token, more = bs.step(np.repeat(logz[:, None], k, 1), np.repeat(tv[:, None], k, 1), np.repeat(ti[:, None], k, 1))
        # all k running beams continue "beam 0" = the prompt: its cache rows go to the prompt's other slots
        base = torch.arange(B, device=self.device) * k
        for j in range(1, k):
            for t5 in kv5:
                t5[:, base + j, :, : int(lens.max())] = t5[:, base, :, : int(lens.max())]
        slot = (np.arange(B)[:, None] * k + np.arange(k)[None, :]).astype(np.int64)            # slot of running beam [b, j]
        ar = np.arange(R, dtype=np.int64)
        q_start, o_start = self._i32(ar * (d.n_q_heads + 2 * Hkv)), self._i32(ar * d.n_q_heads)
        q_len = self._i32(np.full(R, G))
        done = torch.zeros(R, dtype=torch.uint8, device=self.device)
        scratch_tok = torch.empty((R, 1), dtype=I32, device=self.device)
        logits = torch.empty((R, d.vocab), dtype=BF16, device=self.device)
        ws = self._workspace(self._lib.owc_llm_workspace_bytes(C.byref(self.w.llm), R, R))
        g = 1                                                                                    # tokens every running beam holds
        while more:
            # state of the R rows for this step: the fed token is the beam's newest, at rope position max_pos + g, cache row len + g - 1
            st = np.empty((5, R), np.int64)
            st[0] = token.reshape(-1)
            st[1] = np.repeat(max_pos, k) + g
            st[2] = slot.reshape(-1)
            st[3] = np.repeat(lens, k) + g - 1
            st[4] = st[2] * Hkv * s_max
            dev = self._i32(st.astype(np.int32))
            klen = self._i32((st[3] + 1).astype(np.int32))
            rc = self._lib.owc_llm_decode_step(
                self._ctx, C.byref(self.w.llm), C.byref(cache), dev[0].data_ptr(), dev[1].data_ptr(), dev[2].data_ptr(),
                dev[3].data_ptr(), dev[4].data_ptr(), klen.data_ptr(), q_start.data_ptr(), o_start.data_ptr(), q_len.data_ptr(),
                done.data_ptr(), scratch_tok.data_ptr(), 1, 0, None, R, -1, -1, pad_token_id, None, None, None,
                logits.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr())
            _lib.check(rc, self.dev_index)
            logz, tv, ti = ops.beam_candidates(logits, K2)
            logz, tv, ti = (x.cpu().numpy() for x in (logz, tv, ti))
            parent, token, more = bs.step(logz.reshape(B, k), tv.reshape(B, k, K2), ti.reshape(B, k, K2))
            g += 1
            if not more:
                break
            # the cache follows the hypotheses: first child of a parent keeps its slot, further children copy it into a freed slot
            new_slot = np.empty_like(slot)
            src, dst = [], []
            for b in range(B):
                taken = set()
                free = [int(slot[b, j]) for j in range(k) if j not in set(parent[b].tolist())]
                for j in range(k):
                    p_ = int(parent[b, j])
                    if p_ not in taken:
                        taken.add(p_)
                        new_slot[b, j] = slot[b, p_]
                    else:
                        new_slot[b, j] = free.pop()
                        src.append(int(slot[b, p_]))
                        dst.append(int(new_slot[b, j]))
            if src:
                rows_now = int(lens.max()) + g - 1
                s_t = torch.tensor(src, device=self.device)
                d_t = torch.tensor(dst, device=self.device)
                for t5 in kv5:
                    t5[:, d_t, :, :rows_now] = t5[:, s_t, :, :rows_now]
            slot = new_slot
        toks, scores = bs.result()
        out = torch.from_numpy(toks.astype(np.int32)).to(self.device)
        return (out, scores) if return_scores else out

    def _carry_export(self, carry, cin, B, NC, n, cur, state, done2, row_of, steps_run, out_tokens, kc, vc, s_max, Bx,
                      max_new_tokens, forced_tokens, pad_token_id, n_slots, stream_of=None) -> None:
        """End of a pass with straggler hand-over: after the last enqueued step has run, split the rows that were still in the
        decode batch into finished ones and sequences that go on in the next pass (their K / V rows, pending token, position and
        remaining budget are copied out: the pass's own cache is released with the pass)."""
        d = self.d
        torch.cuda.current_stream().synchronize()
        st = state[cur, :, :n].cpu().numpy()                        # [7, n] after the last step (advanced in place)
        dn = done2[cur, :n].cpu().numpy()
        toks = out_tokens.cpu().numpy()
        own_tags = list(carry.get("tags") or range(B))
        c0 = np.zeros(Bx, np.int64)                                 # tokens a row had emitted BEFORE this pass
        cap = np.full(Bx, max_new_tokens, np.int64)                 # its total budget of new tokens
        if NC:
            c0[B:] = np.asarray(cin["emitted"])
            cap[B:] = c0[B:] + np.asarray(cin["remaining"])
        emitted = np.where(np.arange(Bx) < B, 1 + steps_run, c0 + steps_run)    # ... and has now (own rows: column 0 = the prefill's)
        emitted = np.minimum(emitted, cap)
        running = np.zeros(Bx, bool)
        running[row_of[:n][dn == 0]] = True
        running &= emitted < cap
        # carried-in sequences that ended in this pass: their earlier tokens + this pass's columns 1.. (pad behind EOS)
        finished = []
        for i in range(NC):
            r = B + i
            if running[r]:
                continue
            new = toks[r, 1:1 + int(emitted[r] - c0[r])]
            full = np.full(int(cap[r]), pad_token_id, np.int32)
            prev = np.asarray(cin["tokens"][i], np.int32)
            full[:len(prev)] = prev
            full[len(prev):len(prev) + len(new)] = new
            finished.append((cin["tags"][i], full))
        carry["finished"] = finished
        carry["unfinished_rows"] = [int(r) for r in np.flatnonzero(running[:B])]
        idx = np.flatnonzero(running)
        if len(idx) == 0:
            carry["out"] = None
            return
        pos_in_batch = {int(r): k for k, r in enumerate(row_of[:n])}
        comp = np.array([pos_in_batch[int(r)] for r in idx])        # compact row of every exported sequence
        cached = st[2, comp].astype(np.int64)                       # tokens in its cache = the next write index
        w_c = int(cached.max())
        sel = torch.from_numpy(idx.astype(np.int64)).to(self.device)
        kv = (kc.view(d.n_layers, n_slots, d.n_kv_heads, s_max, d.head_dim), vc.view(d.n_layers, n_slots, d.n_kv_heads, s_max, d.head_dim))
        k_out = kv[0][:, :, :, :w_c].index_select(1, sel)
        v_out = kv[1][:, :, :, :w_c].index_select(1, sel)
        tokens, tags, forced_rest = [], [], []
        for r in idx:
            r = int(r)
            if r < B:
                tokens.append(toks[r, :int(emitted[r])].copy())
                tags.append(own_tags[r])
                fr = None if forced_tokens is None else np.asarray(forced_tokens).reshape(B, max_new_tokens)[r, int(emitted[r]):]
            else:
                i = r - B
                prev = np.asarray(cin["tokens"][i], np.int32)
                tokens.append(np.concatenate([prev, toks[r, 1:1 + int(emitted[r] - c0[r])]]))
                tags.append(cin["tags"][i])
                fr = None if forced_tokens is None else np.asarray(cin["forced_rest"][i])[int(emitted[r] - c0[r]):]
            forced_rest.append(fr)
        carry["out"] = {"k": k_out, "v": v_out, "cached": cached, "tok": st[0, comp].astype(np.int64), "pos": st[1, comp].astype(np.int64),
                        "emitted": emitted[idx], "remaining": (cap - emitted)[idx], "tokens": tokens, "tags": tags,
                        "forced_rest": forced_rest, "stream": None if stream_of is None else np.asarray(stream_of)[idx]}

    def score(self, ids, img_embeds: torch.Tensor | None, grids: list, start: int, *, img_rows=None):
        """Teacher-forced scoring of ONE token sequence (prompt + continuation) in a single prefill: the logits of the positions
        start-1 .. S-1.  Returns (logprob float32 [S - start]: log p(ids[i] | ids[:i]) for i = start .. S-1, from the logits of
        position i-1; argmax int32 [S - start]: argmax of the logits AT positions start .. S-1) - the two things a loglikelihood
        request is made of (reference src/models/_llava_hf.py:236-252: HF's shifted cross-entropy and the reference's unshifted
        greedy comparison)."""
        d = self.d
        ids = np.asarray(ids, dtype=np.int32)
        S = len(ids)
        if not 1 <= start <= S:
            raise ValueError("score: need 1 <= start <= len(ids)")
        kc = torch.empty(d.n_layers * d.n_kv_heads * S * d.head_dim, dtype=BF16, device=self.device)
        vc = torch.empty_like(kc)
        cache = _lib.KvCache(kc.data_ptr(), vc.data_ptr(), 1, S)
        if grids:
            p3, _ = positions.mrope_positions(ids, grids, d.image_token_id, d.merge)
        else:
            p3 = np.tile(np.arange(S, dtype=np.int32)[None], (3, 1))
        if int(p3.max()) + 1 > d.max_positions:
            raise ValueError("sequence exceeds the rope table (raise Qwen2VLDims.max_positions)")
        is_img = ids == d.image_token_id
        iidx = np.full(S, -1, dtype=np.int32)
        n_img = int(is_img.sum())
        rows = np.arange(n_img, dtype=np.int32) if img_rows is None else np.asarray(img_rows, dtype=np.int32)
        if len(rows) != n_img or (n_img and (img_embeds is None or rows.min() < 0 or rows.max() >= img_embeds.shape[0])):
            raise ValueError("image token count does not match the image feature rows")
        iidx[is_img] = rows
        want = np.arange(start - 1, S, dtype=np.int32)            # rows whose logits are needed
        n_out = len(want)
        t_ids, t_pos3, t_iidx = self._i32(ids), self._i32(p3), self._i32(iidx)
        t_slot, t_idx = self._i32(np.zeros(S, np.int32)), self._i32(np.arange(S, dtype=np.int32))
        t_start, t_len, t_kstart, t_want = self._i32([0]), self._i32([S]), self._i32([0]), self._i32(want)
        logits = torch.empty((n_out, d.vocab), dtype=BF16, device=self.device)
        top = torch.empty(n_out, dtype=I32, device=self.device)
        ws = self._workspace(self._lib.owc_llm_workspace_bytes(C.byref(self.w.llm), S, max(n_out, 1)))
        rc = self._lib.owc_llm_prefill(
            self._ctx, C.byref(self.w.llm), C.byref(cache), t_ids.data_ptr(), t_iidx.data_ptr(), _lib.ptr(img_embeds),
            t_pos3.data_ptr(), t_slot.data_ptr(), t_idx.data_ptr(), t_start.data_ptr(), t_len.data_ptr(), t_len.data_ptr(),
            t_kstart.data_ptr(), t_want.data_ptr(), 1, n_out, S, S, 0, 1, _lib.PREFILL_SCORE_ROWS, None, 0, top.data_ptr(), logits.data_ptr(), ws.data_ptr(),
            ws.numel(), _lib.stream_ptr())
        _lib.check(rc, self.dev_index)
        if n_out == 1:      # only the last position was asked for: nothing to score
            return np.zeros(0, np.float32), np.zeros(0, np.int32)
        target = self._i32(np.concatenate([ids[start:], [-1]]).astype(np.int32))
        lp = ops.token_logprob_bf16(logits, target)
        return lp[:-1].cpu().numpy(), top[1:].cpu().numpy()

    def _common_prefix(self, prompts, b0: int, b1: int) -> int:
        """Length of the leading run of text tokens shared by every prompt of the chunk (system prompt, question
        preamble, <|vision_start|>): identical ids at identical positions give identical hidden states in every
        layer under causal attention, so they are prefilled once."""
        if not self.share_prefix or b1 - b0 < 2:
            return 0
        first = np.asarray(prompts[b0])
        p = int(min(len(prompts[b]) for b in range(b0, b1))) - 1  # every prompt keeps >= 1 own token
        img = np.flatnonzero(first[:p] == self.d.image_token_id)
        if len(img):
            p = int(img[0])
        for b in range(b0 + 1, b1):
            if p <= 0:
                break
            neq = np.flatnonzero(np.asarray(prompts[b][:p]) != first[:p])
            if len(neq):
                p = int(neq[0])
        return p if p >= self.min_shared_prefix else 0

    def _prefill_chunk(self, prompts, pos_list, img_index, img_embeds, lens, b0, b1, cache, next_tok, first_logits, sampling=None,
                       slot_of=None):
        """`slot_of` (optional): prompt index -> cache slot (default: the prompt's own index; beam search spaces them num_beams apart)."""
        d = self.d
        n = b1 - b0
        slot_of = slot_of or (lambda i: i)
        P = self._common_prefix(prompts, b0, b1)
        sl = lens[b0:b1] - P                       # rows each prompt contributes
        ids = [np.asarray(prompts[b], dtype=np.int32)[P:] for b in range(b0, b1)]
        pos3 = [pos_list[b][:, P:] for b in range(b0, b1)]
        iidx = [img_index[b][P:] for b in range(b0, b1)]
        tok_slot = [np.full(int(sl[i]), slot_of(b0 + i), dtype=np.int32) for i in range(n)]
        tok_idx = [np.arange(P, int(lens[b0 + i]), dtype=np.int32) for i in range(n)]
        starts = np.concatenate([[0], np.cumsum(sl)[:-1]])
        q_len, k_len = sl.copy(), lens[b0:b1].copy()
        k_start = np.array([slot_of(i) for i in range(b0, b1)], dtype=np.int64) * d.n_kv_heads * cache.s_max
        last = starts + sl - 1
        if P:  # the shared prefix rides along as one extra segment at the end of the packed rows
            ids.append(np.asarray(prompts[b0], dtype=np.int32)[:P])
            pos3.append(pos_list[b0][:, :P])
            iidx.append(np.full(P, -1, dtype=np.int32))
            tok_slot.append(np.full(P, -1, dtype=np.int32))   # -1: K/V rows are written to every slot of the chunk
            tok_idx.append(np.arange(P, dtype=np.int32))
            starts = np.concatenate([starts, [int(sl.sum())]])
            q_len, k_len = np.concatenate([q_len, [P]]), np.concatenate([k_len, [P]])
            k_start = np.concatenate([k_start, [k_start[0]]])
        T = int(sum(len(x) for x in ids))
        t_ids, t_pos3, t_iidx = self._i32(np.concatenate(ids)), self._i32(np.concatenate(pos3, axis=1)), self._i32(np.concatenate(iidx))
        t_slot, t_idx = self._i32(np.concatenate(tok_slot)), self._i32(np.concatenate(tok_idx))
        t_start, t_klen, t_qlen = self._i32(starts), self._i32(k_len), self._i32(q_len)
        t_kstart, t_last = self._i32(k_start), self._i32(last)
        n_seq = len(starts)
        nbytes = self._lib.owc_llm_workspace_bytes(C.byref(self.w.llm), T, n_seq)
        ws = self._workspace(nbytes)
        logits_ptr = first_logits[b0:b1].data_ptr() if first_logits is not None else None
        rc = self._lib.owc_llm_prefill(
            self._ctx, C.byref(self.w.llm), C.byref(cache), t_ids.data_ptr(), t_iidx.data_ptr(),
            _lib.ptr(img_embeds), t_pos3.data_ptr(), t_slot.data_ptr(), t_idx.data_ptr(), t_start.data_ptr(),
            t_klen.data_ptr(), t_qlen.data_ptr(), t_kstart.data_ptr(), t_last.data_ptr(), n_seq, n, T,
            int(q_len.max()), b0, n, _lib.PREFILL_LAST_TOKENS, sampling, b0, next_tok[b0:b1].data_ptr(), logits_ptr, ws.data_ptr(), ws.numel(),
            _lib.stream_ptr())
        _lib.check(rc, self.dev_index)
