#!/usr/bin/env python3
"""Throughput of the LLaVA row of the hot path (BASELINE.json config #4) on one MI355X.

  python tools/bench_llava.py --model llava-1.5-7b --batch 512
  python tools/bench_llava.py --model llava-next-34b --batch 64 --image-size 480x640

One step = uint8 views resident in HBM -> owc_clip_patchify_u8 -> CLIP tower + projector -> packed prefill ->
greedy decode -> token ids on host.  Not the headline metric (bench.py is); same measurement method.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

PEAK = 2500.0e12


def flops(d, n_views: int, S: int, T: int) -> float:
    E, F, D, t = d.v_embed, d.v_mlp, d.d_model, d.tokens
    f_clip = n_views * (2 * (t - 1) * 588 * E + d.v_run_layers * (8 * t * E * E + 4 * t * t * E + 4 * t * E * F) + 2 * t * (E * D + D * D))
    H, KV, hd, ff, L, V = d.n_q_heads, d.n_kv_heads, d.head_dim, d.d_ff, d.n_layers, d.vocab
    f_pre = L * (2 * S * D * (H + 2 * KV) * hd + 2 * S * H * hd * D + 2 * S * S * H * hd + 6 * S * D * ff) + 2 * D * V
    f_dec = sum(L * (2 * D * (H + 2 * KV) * hd + 2 * H * hd * D + 4 * (S + i) * H * hd + 6 * D * ff) + 2 * D * V for i in range(T - 1))
    return float(f_clip + f_pre + f_dec)


def not_executed(d, S: int, shared_rows: float) -> float:
    """Nominal-forward FLOPs the path skips (see bench.py pruned_flops_per_image): the last prefill layer's attention / o-proj /
    MLP for all but the last token, and the text tokens before the image that every prompt shares (prefilled once per group)."""
    D, H, KV, hd, ff, L = d.d_model, d.n_q_heads, d.n_kv_heads, d.head_dim, d.d_ff, d.n_layers
    last_layer = (S - 1) * (2 * H * hd * D + 6 * D * ff) + 2 * S * S * H * hd - 4 * S * H * hd
    per_row = (L - 1) * (2 * D * (H + 2 * KV) * hd + 2 * H * hd * D + 6 * D * ff) + 2 * D * (H + 2 * KV) * hd
    return float(last_layer + shared_rows * per_row)


def run(model: str, batch: int, steps: int = 2, warmup: int = 1, new_tokens: int = 16, image_size: str = "480x640",
        text_tokens: int = 48, decoder_dtype: str = "bf16") -> dict:
    """One measurement (also a leg of bench.py's default run: `llava_next_34b_leg`)."""
    from lmms_owc_amd import _lib
    from lmms_owc_amd.engine import anyres
    from lmms_owc_amd.engine.llava import DIMS, LlavaEngine, LlavaWeights
    from lmms_owc_amd.models import imageproc

    device = torch.device("cuda", torch.cuda.current_device())
    d = DIMS[model]
    if decoder_dtype != "bf16":
        import dataclasses

        d = dataclasses.replace(d, decoder_dtype=decoder_dtype)
    t_w = time.perf_counter()
    eng = LlavaEngine(LlavaWeights.random(d, device, seed=1234))
    torch.cuda.synchronize()
    t_w = time.perf_counter() - t_w
    h, w = (int(x) for x in image_size.split("x"))
    nv = anyres.num_views((h, w), d.grid_pinpoints, d.image_size) if d.grid_pinpoints else 1
    B, T = batch, new_tokens
    u8 = torch.randint(0, 256, (B * nv, 3, d.image_size, d.image_size), dtype=torch.uint8, device=device)
    rows = eng.feature_rows([nv] * B, [(h, w)] * B)
    r = np.random.default_rng(0)
    head, tail = r.integers(1000, 30000, text_tokens // 2), r.integers(1000, 30000, text_tokens - text_tokens // 2)
    prompts = [np.concatenate([head, np.full(len(rows[b]), d.image_token_id), tail]).astype(np.int32) for b in range(B)]
    S = len(prompts[0])

    def step():
        feats = eng.encode_views(eng.patchify(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD))
        return eng.generate_from_features(prompts, feats, rows, T, eos_token_id=-1, pad_token_id=0).cpu()

    lib, ctx = _lib.load(), _lib.ctx(device.index or 0)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    lib.owc_gemm_profile_enable(ctx, 1)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms2, fl2, n2 = (C.c_double * 2)(), (C.c_double * 2)(), (C.c_int64 * 2)()
    _lib.check(lib.owc_gemm_profile_read(ctx, ms2, fl2, n2), 0)
    ms, fl, n = C.c_double(ms2[0] + ms2[1]), C.c_double(fl2[0] + fl2[1]), C.c_int64(n2[0] + n2[1])
    fp8_tf = fl2[1] / (ms2[1] * 1e-3) / 1e12 if ms2[1] > 0 else None
    lib.owc_gemm_profile_enable(ctx, 0)
    assert out.shape == (B, T)
    ips = B * steps / dt
    f_model = flops(d, nv, S, T)
    per_group = max(1, min(B, 65536 // S))  # prompts per prefill launch group (engine default prefill_chunk_tokens)
    f = f_model - not_executed(d, S, len(head) * (1.0 - 1.0 / per_group))  # executed FLOPs: what the utilisation is priced on
    return {"metric": f"images/s {model} open-world classify (1 GPU)", "value": ips, "unit": "images/s", "dtype": "bf16" if decoder_dtype == "bf16" else "fp8-e4m3 decoder projections, bf16 elsewhere",
            "data": "synthetic", "ms_per_step": dt / steps * 1e3,
            "config": {"workload": f"{model}: {B} synthetic {h}x{w} images per step, {nv} CLIP view(s) of {d.image_size}px each, "
                                   f"prompt S={S} ({len(rows[0])} image tokens), {T} forced greedy tokens, random weights",
                       "views_per_image": nv, "prompt_tokens": S, "new_tokens": T},
            "model_flops_per_image": f_model, "executed_flops_per_image": f, "mfma_frac_end_to_end": ips * f / PEAK,
            "roofline": {"bound": "mfma", "achieved": fl.value / (ms.value * 1e-3) / 1e12, "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": fl.value / (ms.value * 1e-3) / PEAK, "share_of_step_time": ms.value * 1e-3 / dt,
                         "launches": int(n.value)},
            "decoder_dtype": decoder_dtype, "fp8_gemm_tflops": fp8_tf, "weights_gb": eng.w.nbytes() / 1e9, "weight_init_seconds": t_w}


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llava-1.5-7b")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--new-tokens", type=int, default=16)
    ap.add_argument("--image-size", default="480x640", help="HxW of the synthetic source images (anyres tiling depends on it)")
    ap.add_argument("--text-tokens", type=int, default=48)
    ap.add_argument("--decoder-dtype", default="bf16", choices=["bf16", "fp8"])
    args = ap.parse_args()
    torch.cuda.set_device(0)
    print(json.dumps(run(args.model, args.batch, args.steps, args.warmup, args.new_tokens, args.image_size, args.text_tokens,
                         args.decoder_dtype)), flush=True)


if __name__ == "__main__":
    main()
