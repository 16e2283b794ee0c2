#!/usr/bin/env python3
"""Headline benchmark of the MI355X open-world classification hot path (BASELINE.json metric).

One "step" = one pass of the hot path over one batch of synthetic input per GPU:
  pixel_values [B*1024, 1176] bf16 (B images of 448x448, resident in HBM)
    -> Qwen2-VL vision tower -> prompt prefill (S = 286) -> 16 greedy decode steps -> token ids on host.
Reported `value` = images/s over all ranks (weak scaling: per-GPU batch fixed).  The same JSON line carries
the label-cosine/s of the scorer leg, the MFMA roofline of the dominant kernel (bf16 GEMM, HIP events
around every launch of the timed region) and the reference's CPU HuggingFace path timed on the host cores.

  python bench.py --gpus 1 --steps 3 --warmup 1
  python bench.py --gpus 8 --steps 3 --warmup 1          # spawns its own 8 ranks (one process per GPU, RCCL)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
         bench.py --gpus 8 --steps 3 --warmup 1          # or under a launcher that already set RANK / WORLD_SIZE
  python bench.py --gpus 2 --dry-run                     # CPU check of the launcher + rendezvous (gloo), no HIP
"""

from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

T_PROCESS_START = time.perf_counter()   # `leg_seconds` / `--leg-budget-s` count from here (torch import included)

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("TOKENIZERS_PARALLELISM", "false")

import numpy as np  # noqa: E402
import torch  # noqa: E402

S_TEXT_BEFORE, S_IMG, S_TEXT_AFTER = 14, 256, 16  # 286-token prompt of the 448x448 classification query
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md (never the 2:1-sparsity figure)
# CPU affinity of the process as the launcher gave it to us, before anything pins a thread (lmms_owc_amd.models._base.pin_to_gpu_numa_node)
FULL_AFFINITY = frozenset(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else frozenset()


def restore_full_affinity() -> dict:
    """Before a CPU-baseline leg: EVERY thread of this process (/proc/self/task) must be allowed on every core the process started with.
    A thread created while the launch thread was pinned to its GPU's NUMA share keeps that mask for good (round 5's CPU baseline ran
    its 128 intra-op threads on one socket's share: 0.034 -> 0.020-0.031 images/s across records).  Threads found narrower are reset and
    COUNTED in the record - `narrow_threads_found` must read 0 when the plug-in's own `release_host_resources` did its job."""
    if not FULL_AFFINITY:
        return {"checked": 0}
    found, tids = 0, []
    try:
        tids = [int(t) for t in os.listdir("/proc/self/task")]
    except OSError:
        tids = [0]
    for tid in tids:
        try:
            if frozenset(os.sched_getaffinity(tid)) != FULL_AFFINITY:
                found += 1
                os.sched_setaffinity(tid, FULL_AFFINITY)
        except OSError:
            pass
    return {"checked": len(tids), "narrow_threads_found": found, "cpus_allowed": len(os.sched_getaffinity(0))}


FULLSIZE_LOGIT_BOUND = 0.03   # max |HIP - HF| / max |HF| per teacher-forced step of the full 7B model (observed 1.7-2.1 %)
TRAFFIC_KERNEL = "gemm_bf16_nt_256pp_kernel"   # what the default dispatch launches for the 7B gate/up prefill group
RENDEZVOUS_FAILED = 75   # exit code of a rank whose init_process_group failed (EX_TEMPFAIL): self_launch retries on a new port
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 (block-scaled f8f6f4 MFMA) peak, same source


def flops_per_image(d, new_tokens: int, patches: int = 1024) -> float:
    """SURVEY.md §8(d) algorithmic FLOPs per image of `patches` 14x14 patches (448x448: 1024; causal attention counted at 1/2)."""
    P, Dv, Lv, S = patches, d.v_embed, d.v_depth, S_TEXT_BEFORE + patches // 4 + S_TEXT_AFTER
    if getattr(d, "v_variant", 0) == 1:   # Qwen2.5-VL: gated MLP (3 matrices), window attention (<= 64 keys) except in the full-attention blocks
        n_full = len(d.v_fullatt)
        attn = n_full * 4 * P * P * Dv + (Lv - n_full) * 4 * P * min(P, 64) * Dv
        f_vit = 2 * P * d.patch_k * Dv + Lv * (8 * P * Dv * Dv + 6 * P * Dv * d.v_mlp) + attn
    else:
        f_vit = 2 * P * d.patch_k * Dv + Lv * (8 * P * Dv * Dv + 4 * P * P * Dv + 4 * P * Dv * d.v_mlp)
    f_vit += 2 * (P // 4) * (4 * Dv) ** 2 + 2 * (P // 4) * 4 * Dv * d.d_model
    H, KV, hd, dm, ff, L, V = d.n_q_heads, d.n_kv_heads, d.head_dim, d.d_model, d.d_ff, d.n_layers, d.vocab
    f_pre = L * (2 * S * dm * (H + 2 * KV) * hd + 2 * S * H * hd * dm + 2 * S * S * H * hd + 6 * S * dm * ff) + 2 * dm * V
    # the first new token comes out of the prefill logits: T new tokens need T - 1 single-token forwards
    f_dec = sum(L * (2 * dm * (H + 2 * KV) * hd + 2 * H * hd * dm + 4 * (S + i) * H * hd + 6 * dm * ff) + 2 * dm * V
                for i in range(new_tokens - 1))
    return float(f_vit + f_pre + f_dec)


def pruned_flops_per_image(d, prompts_per_chunk: int, S: int | None = None) -> float:
    """FLOPs of the model's nominal forward that the path does NOT execute, subtracted before any utilisation figure:
    * owc_llm_prefill runs the last decoder layer's attention, o-proj and MLP for the last token of each prompt only (the
      other rows' outputs feed nothing; logits bit-identical, tests/test_qwen2vl_gpu.py);
    * the S_TEXT_BEFORE leading text tokens are identical in every prompt of the task and are prefilled once per launch group
      (shared-prefix segment, bit-identical, same test file), so each prompt contributes S - 14 (1 - 1/n) rows."""
    S = S if S is not None else S_TEXT_BEFORE + S_IMG + S_TEXT_AFTER
    H, KV, hd, dm, ff, L = d.n_q_heads, d.n_kv_heads, d.head_dim, d.d_model, d.d_ff, d.n_layers
    last_layer = (S - 1) * (2 * H * hd * dm + 6 * dm * ff) + 2 * S * S * H * hd - 4 * S * H * hd
    shared_rows = S_TEXT_BEFORE * (1.0 - 1.0 / max(prompts_per_chunk, 1))
    per_row = (L - 1) * (2 * dm * (H + 2 * KV) * hd + 2 * H * hd * dm + 6 * dm * ff) + 2 * dm * (H + 2 * KV) * hd
    return float(last_layer + shared_rows * per_row)


def baseline_threads() -> int:
    """Intra-op threads of the CPU baselines: the CPUs the cgroup lets this process keep busy (one per physical core at most: 128
    of the host's 256 logical CPUs).  Rounds 1-5 took `os.cpu_count()` - 128 threads on a box whose cgroup grants 16 CPUs, which
    throttles the whole group: the baseline's dominant bf16 Linear measured 122 ms on 128 threads and 39 ms on 16
    (tools/probes/cpu_threads_under_quota.py), so those rounds under-reported the CPU baseline ~3x and spent 90-140 s on it."""
    from lmms_owc_amd.models._base import usable_cpus

    return max(1, min(usable_cpus()[0], 128))


def host_cpu_record() -> dict:
    from lmms_owc_amd.models._base import usable_cpus

    n, quota = usable_cpus()
    return {"logical_cpus": os.cpu_count(), "cgroup_cpu_quota": quota, "usable_cpus": n}


def cpu_baseline_lmm(dims, device, seed: int, new_tokens: int, pix_dev, hip_tokens, n_images: int, engine=None,
                     budget_s: float = 150.0) -> dict:
    """The reference's CPU path: HF Qwen2VLForConditionalGeneration.generate, batch 1, greedy (src/models/_qwen2_vl.py:308-329),
    on the SAME seeded weights (`random_param`, regenerated per HF parameter name and copied to the host) and the SAME
    pixel_values / prompt ids as the HIP run's first `n_images` images; its tokens are compared with the HIP tokens."""
    from transformers import Qwen2VLConfig, Qwen2VLForConditionalGeneration

    from lmms_owc_amd.engine.qwen2vl import hf_param_names, random_param

    d = dims
    affinity = restore_full_affinity()
    threads = baseline_threads()
    torch.set_num_threads(threads)
    cfg = Qwen2VLConfig(
        text_config=dict(hidden_size=d.d_model, num_hidden_layers=d.n_layers, num_attention_heads=d.n_q_heads,
                         num_key_value_heads=d.n_kv_heads, intermediate_size=d.d_ff, vocab_size=d.vocab,
                         rms_norm_eps=d.rms_eps, max_position_embeddings=4096, tie_word_embeddings=d.tie_embeddings,
                         rope_parameters=dict(rope_type="default", rope_theta=d.rope_theta, mrope_section=list(d.mrope_section))),
        vision_config=dict(depth=d.v_depth, embed_dim=d.v_embed, num_heads=d.v_heads, hidden_size=d.d_model, mlp_ratio=4,
                           patch_size=14, spatial_merge_size=2, temporal_patch_size=2),
        image_token_id=d.image_token_id, video_token_id=d.image_token_id + 1, vision_start_token_id=d.image_token_id - 3,
        vision_end_token_id=d.image_token_id - 2, tie_word_embeddings=d.tie_embeddings)
    cfg._attn_implementation = "sdpa"
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    ctx = None    # skip HF's own random init of 7.6 B parameters (minutes of host time): every parameter is overwritten below
    for mod in ("transformers.initialization", "transformers.modeling_utils"):   # (transformers 5.x / 4.x)
        try:
            ctx = ctx or getattr(__import__(mod, fromlist=["no_init_weights"]), "no_init_weights", None)
        except ImportError:
            pass
    try:
        if ctx is not None:
            with ctx():
                model = Qwen2VLForConditionalGeneration(cfg)
        else:
            model = Qwen2VLForConditionalGeneration(cfg)
    finally:
        torch.set_default_dtype(prev)
    model = model.to(torch.bfloat16).eval()
    sd = model.state_dict()
    with torch.no_grad():
        for name in hf_param_names(d):
            sd[name].copy_(random_param(d, name, device, seed).reshape(sd[name].shape).cpu())
        if d.tie_embeddings and "lm_head.weight" in sd:
            sd["lm_head.weight"].copy_(sd["model.language_model.embed_tokens.weight"])
    ids = prompt_ids(d.image_token_id)
    inp = torch.from_numpy(ids.astype(np.int64))[None]
    mm = (inp == d.image_token_id).int()
    grid = torch.tensor([[1, 32, 32]])
    times, same_first, same_all, logit_err, flips, forced_equal = [], 0, 0, [], [], 0
    t_leg = time.perf_counter()
    with torch.no_grad():
        for i in range(n_images):
            # a BOUNDED sample: on a slow or busy host (40-50 s per image seen) the leg stops after the image that crosses `budget_s`,
            # but never before two images are done (the first is the warm-up; `value` needs one after it)
            if i >= 2 and time.perf_counter() - t_leg > budget_s:
                n_images = i
                break
            pix = pix_dev[i * 1024:(i + 1) * 1024].cpu()
            t0 = time.perf_counter()
            gen = model.generate(input_ids=inp, attention_mask=torch.ones_like(inp), pixel_values=pix, image_grid_thw=grid,
                                 mm_token_type_ids=mm, do_sample=False, num_beams=1, max_new_tokens=new_tokens,
                                 min_new_tokens=new_tokens, use_cache=True, pad_token_id=0, return_dict_in_generate=True,
                                 output_logits=True)
            times.append(time.perf_counter() - t0)
            new = gen.sequences[0, inp.shape[1]:].tolist()
            hip = [int(t) for t in hip_tokens[i].tolist()]
            same_first += int(new[0] == hip[0])
            same_all += int(new == hip)
            if engine is not None and gen.logits is not None:
                # full-size logit parity: the HIP engine teacher-forced on HF's continuation, every step against HF's CPU logits
                emb = engine.encode_images(pix_dev[i * 1024:(i + 1) * 1024], [(1, 32, 32)])
                _, sl = engine.generate([ids], emb, [[(1, 32, 32)]], new_tokens, forced_tokens=np.asarray(new)[None],
                                        return_step_logits=True)
                sl = sl[:, 0].float().cpu()
                for j, ref in enumerate(gen.logits):
                    ref = ref[0].float()
                    scale = float(ref.abs().max())
                    err = float((sl[j] - ref).abs().max()) / scale
                    logit_err.append(err)
                    # teacher-forced token comparison: both sides conditioned on HF's continuation, so every step is comparable
                    hip_tok = int(sl[j].argmax())
                    if hip_tok == new[j]:
                        forced_equal += 1
                    else:   # a flip is legitimate only on a near-tie: show HF's top-2 margin next to the HIP error at this step
                        top2 = torch.topk(ref, 2).values
                        flips.append({"image": i, "step": j, "hf_token": int(new[j]), "hip_token": hip_tok,
                                      "hf_top2_margin_rel": float(top2[0] - top2[1]) / scale, "hip_logit_err_rel": err,
                                      "hf_logit_of_hip_token_below_hf_max_rel": float(top2[0] - ref[hip_tok]) / scale,
                                      "explained_by_error": bool(float(top2[0] - ref[hip_tok]) / scale <= 2.0 * err)})
    del model
    best = float(np.mean(times[1:])) if len(times) > 1 else float(times[0])
    import transformers

    return {"value": 1.0 / best, "unit": "images/s", "cores": threads, "kind": "reference", "thread_affinity": affinity, "host": host_cpu_record(),
            "sample": f"{n_images} image(s) 448x448 = the HIP run's first images (same pixel_values, prompt ids and seeded weights), batch 1, "
                      f"bf16, transformers {transformers.__version__} Qwen2VLForConditionalGeneration.generate on CPU (greedy, {new_tokens} "
                      f"new tokens); mean of images after the first; per-image s = {[round(t, 2) for t in times]}",
            "logits_vs_hip": {"steps_compared": len(logit_err), "worst_rel_err": max(logit_err) if logit_err else None,
                              "mean_rel_err": float(np.mean(logit_err)) if logit_err else None,
                              "what": "max |HIP - HF| / max |HF| per step, HIP engine teacher-forced on HF's tokens (full 7B, same weights)"},
            "tokens_vs_hip": {"images": n_images, "first_token_equal": same_first, "all_tokens_equal": same_all,
                              "teacher_forced_steps_equal": forced_equal, "teacher_forced_steps": len(logit_err), "flips": flips,
                              "note": "`all_tokens_equal` compares FREE-RUNNING continuations (after one near-tie flip they legitimately "
                                      "diverge); `teacher_forced_*` compares argmax per step with both sides conditioned on HF's tokens. "
                                      "Random weights give near-flat logits over a 152k vocabulary: every flip is listed with HF's top-2 "
                                      "margin and the HIP logit error at that step - a flip whose margin exceeds twice the error fails the run"}}


def cpu_baseline_scorer(n_labels: int, L: int) -> dict:
    """Reference scorer on CPU fp32: BertModel (MiniLM-L6 config) + mean pool + L2 + paired bmm (_text.py:175-202)."""
    from transformers import BertConfig, BertModel

    from lmms_owc_amd.engine.scorer import MINILM_L6

    threads = baseline_threads()
    torch.set_num_threads(threads)
    m = BertModel(BertConfig(**MINILM_L6), add_pooling_layer=False).eval()
    g = torch.Generator().manual_seed(0)
    ids = torch.randint(1000, 30000, (n_labels, L), generator=g)
    mask = torch.ones_like(ids)

    def run():
        with torch.no_grad():
            outs = []
            for i in range(0, n_labels, 1024):  # datasets.map(batch_size=1024), _group.py:524-535
                h = m(input_ids=ids[i:i + 1024], attention_mask=mask[i:i + 1024])[0]
                mk = mask[i:i + 1024].unsqueeze(-1).float()
                z = (h * mk).sum(1) / mk.sum(1).clamp(min=1e-9)
                outs.append(z / z.norm(p=2, dim=-1, keepdim=True))
            z = torch.cat(outs)
            return torch.bmm(z.unsqueeze(1), z.unsqueeze(2)).squeeze()

    run()
    t0 = time.perf_counter()
    run()
    dt = time.perf_counter() - t0
    return {"value": n_labels / dt, "unit": "labels/s", "cores": threads, "kind": "reference", "host": host_cpu_record(),
            "sample": f"{n_labels} labels x {L} tokens, HF BertModel fp32 CPU batches of 1024 + paired bmm"}


def prompt_ids(image_token_id: int, n_image_tokens: int = S_IMG) -> np.ndarray:
    r = np.random.default_rng(1234)
    return np.concatenate([r.integers(1000, 150000, S_TEXT_BEFORE), np.full(n_image_tokens, image_token_id),
                           r.integers(1000, 150000, S_TEXT_AFTER)]).astype(np.int32)


# ---- real-size workload (BASELINE config #3: Food-101 + DTD + Flowers-102; none of them is 448x448).  There are no dataset files
# offline, so the (height, width) distributions are modelled on the datasets' published sizing rules - stated here, seeded, and
# only ever used for the extra `--image-sizes` leg (never `value`):
#   food101:    "rescaled to a maximum side length of 512": mostly 512x512, the rest 4:3 / 3:2 landscape or portrait
#   dtd:        "sizes range between 300x300 and 640x640": both sides uniform in [300, 640]
#   flowers102: "smallest side 500": the other side uniform in [500, 1000], landscape or portrait
def _sizes_food101(r, n):
    table = [(512, 512)] * 12 + [(384, 512)] * 3 + [(512, 384)] * 2 + [(341, 512), (512, 341), (306, 512)]
    return [table[i] for i in r.integers(0, len(table), n)]


def _sizes_dtd(r, n):
    return [(int(h), int(w)) for h, w in r.integers(300, 641, (n, 2))]


def _sizes_flowers102(r, n):
    out = []
    for long_side, portrait in zip(r.integers(500, 1001, n), r.random(n) < 0.35):
        out.append((int(long_side), 500) if portrait else (500, int(long_side)))
    return out


# BASELINE.json configs[2] evaluates the three datasets together: their test splits hold 30 300 : 1 692 : 2 463 images
CONFIG3_MIX = (("food101", 30300), ("dtd", 1692), ("flowers102", 2463))


def _sizes_config3(r, n):
    total = sum(c for _, c in CONFIG3_MIX)
    counts = [int(round(n * c / total)) for _, c in CONFIG3_MIX]
    counts[0] += n - sum(counts)
    out = []
    for (name, _), c in zip(CONFIG3_MIX, counts):
        out += DATASET_SIZES[name](r, c)
    order = r.permutation(len(out))      # a task's documents arrive interleaved, not dataset by dataset
    return [out[i] for i in order]


def _sizes_max_pixels(r, n):
    """Every image lands on the reference's `max_pixels` cap (1024 x 28 x 28, `_qwen2_vl.py:64-65`): SUN397 / Stanford-Cars-like
    sources of >= 0.8 Mpx - half of them square (-> 896 x 896 = a 64 x 64 patch grid = 1024 image tokens), half 4:3 landscape
    (1024 x 768 -> 756 x 1036 = a 54 x 74 grid = 999 image tokens)."""
    return [(1024, 1024) if i % 2 == 0 else (768, 1024) for i in range(n)]


DATASET_SIZES = {"food101": _sizes_food101, "dtd": _sizes_dtd, "flowers102": _sizes_flowers102}
DATASET_SIZES["config3"] = _sizes_config3
DATASET_SIZES["max_pixels"] = _sizes_max_pixels


def gemm_roofline(p: dict, dt_total: float, what: str) -> dict:
    """`roofline` object of a leg from the library's bf16-GEMM launch class (HIP events around every launch of the leg)."""
    tf = p["work"] / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
    return {"bound": "mfma", "kernel": "gemm_bf16_nt_* (all epilogues) " + what, "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": tf / PEAK_BF16_TFLOPS, "traffic": None, "launches": p["launches"], "kernel_ms_total": p["ms"],
            "share_of_leg_time": p["ms"] * 1e-3 / time_or(dt_total)}


def ragged_leg(engine, dims, name: str, n: int, T: int, steps: int, device, sync, min_pixels: int = 4 * 784, max_pixels: int = 1024 * 784,
               profile=None) -> dict:
    """The reference resizes every image inside [min_pixels, max_pixels] (`_qwen2_vl.py:64-65, 299-305`) -> 64...1024 image tokens
    per image.  `n` images with the dataset's size distribution -> smart_resize (two stages, like `imageproc.prepare_image`) ->
    uint8 uniform pixels -> owc_patchify_u8 -> ragged vision launch groups -> prompts of 14 + n_tok + 16 tokens (unequal lengths:
    KV slots sized by the longest, shared 14-token prefix) -> `T` greedy tokens."""
    from lmms_owc_amd import ops as owc_ops
    from lmms_owc_amd.models import imageproc

    r = np.random.default_rng(4321)
    sizes = DATASET_SIZES[name](r, n)
    grids = []
    for h, w in sizes:
        h1, w1 = imageproc.smart_resize(h, w, 28, 4 * 28 * 28, 16384 * 28 * 28)         # qwen_vl_utils.fetch_image
        h2, w2 = imageproc.smart_resize(h1, w1, 28, min_pixels, max_pixels)             # the HF processor
        grids.append((1, h2 // 14, w2 // 14))
    gen = torch.Generator(device=device).manual_seed(99)
    by_size: dict = {}
    for i, g in enumerate(grids):
        by_size.setdefault(g, []).append(i)
    pieces = [None] * n
    for (_, gh, gw), idx in by_size.items():   # one patchify launch per distinct size (outside the timed region)
        u8 = torch.randint(0, 256, (len(idx), 3, gh * 14, gw * 14), generator=gen, device=device, dtype=torch.uint8)
        pv = owc_ops.patchify_u8(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD)
        for j, i in enumerate(idx):
            pieces[i] = pv[j * gh * gw:(j + 1) * gh * gw]
    pix = torch.cat(pieces)
    del pieces
    n_tok = [g[1] * g[2] // 4 for g in grids]
    prompts = [prompt_ids(dims.image_token_id, t) for t in n_tok]
    gpp = [[g] for g in grids]

    def step():
        emb = engine.encode_images(pix, grids)
        return engine.generate(prompts, emb, gpp, T, eos_token_id=-1, pad_token_id=0).cpu()

    out = step()
    sync()
    if profile:
        profile(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    sync()
    dt = (time.perf_counter() - t0) / steps
    prof = profile(False) if profile else None
    # property checks outside the timed region: determinism, and a short + a long image alone == inside the ragged batch
    again = step()
    lo, hi = int(np.argmin(n_tok)), int(np.argmax(n_tok))
    inv = bool(torch.equal(again, out))
    for i in (lo, hi):
        row0 = sum(g[1] * g[2] for g in grids[:i])
        emb1 = engine.encode_images(pix[row0:row0 + grids[i][1] * grids[i][2]], [grids[i]])
        solo = engine.generate([prompts[i]], emb1, [gpp[i]], T, eos_token_id=-1, pad_token_id=0).cpu()
        inv = inv and bool(torch.equal(solo[0], out[i]))
    flops = float(sum(flops_per_image(dims, T, g[1] * g[2]) for g in grids))
    # executed FLOPs (what `mfma_frac_end_to_end` is priced on, as in the headline): minus the last prefill layer's dead rows and
    # the shared 14-token prefix, per image at ITS prompt length
    mean_rows = float(np.mean(n_tok)) + S_TEXT_AFTER
    per_chunk = max(1, min(n, int((engine.prefill_chunk_tokens - S_TEXT_BEFORE) // mean_rows))) if engine.share_prefix else 1
    f_exec = flops - float(sum(pruned_flops_per_image(dims, per_chunk, S_TEXT_BEFORE + t + S_TEXT_AFTER) for t in n_tok))
    att = None
    if prof:
        P2 = float(sum((g[1] * g[2]) ** 2 for g in grids))
        a_fl = steps * dims.v_depth * 4.0 * P2 * dims.v_embed               # QK^T + PV, non-causal, every layer, every image
        pa = prof["attn_vision"]
        a_tf = a_fl / (pa["ms"] * 1e-3) / 1e12 if pa["ms"] > 0 else 0.0
        att = {"bound": "mfma", "kernel": "attn_fwd_kernel<80,false>", "achieved": a_tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
               "frac": a_tf / PEAK_BF16_TFLOPS, "traffic": None, "launches": pa["launches"], "kernel_ms_total": pa["ms"],
               "share_of_leg_time": pa["ms"] * 1e-3 / time_or(dt * steps),
               "patches_per_image": {"min": min(g[1] * g[2] for g in grids), "mean": float(np.mean([g[1] * g[2] for g in grids])),
                                     "max": max(g[1] * g[2] for g in grids)}}
    return {"dataset_size_model": name, "images": n, "seconds_per_pass": dt, "images_per_s": n / dt, "image_tokens_per_s": sum(n_tok) / dt,
            "image_tokens_per_image": {"min": min(n_tok), "mean": float(np.mean(n_tok)), "max": max(n_tok)},
            "distinct_grids": len(by_size), "prompt_tokens": {"min": 30 + min(n_tok), "max": 30 + max(n_tok)},
            "model_flops_per_image_mean": flops / n, "executed_flops_per_image_mean": f_exec / n,
            "mfma_frac_end_to_end": f_exec / dt / (PEAK_BF16_TFLOPS * 1e12),
            "mfma_frac_end_to_end_nominal": flops / dt / (PEAK_BF16_TFLOPS * 1e12),
            "deterministic_and_batch_invariant": inv,
            "roofline": gemm_roofline(prof["gemm_bf16"], dt * steps, "of the ragged launch groups") if prof else None,
            "roofline_attention_vision": att,
            "what": "seeded (height, width) model of the dataset's published sizing rule -> smart_resize within [3136, 802816] px -> "
                    "ragged cu_seqlens vision launch groups + unequal prompts (shared 14-token prefix); uniform-noise pixels; never `value`"}


class BoxCalibration:
    """How fast is THIS box?  MI355X devices differ by up to ~12 % on MFMA-dense loops (MI355X_MICROARCH.md, DVFS give-back item 5:
    the clock a device holds under matrix load), and the driver's headline moves with it.  The yardstick is the step's own largest
    launch class - the 7B gate/up projection with the SwiGLU epilogue at M = 65536 (`gemm_bf16_nt_256pp_kernel`, 17.8 TFLOP per
    launch) on random operands - run back to back for ~2 s BEFORE and AFTER the timed region, TFLOP/s from HIP events on the
    launch stream.  `value_per_calibration_tflops` = images/s per calibration TFLOP/s should be the same on every box."""

    M, N, K = 65536, 37888, 3584

    def __init__(self, device, seconds: float = 2.0):
        from lmms_owc_amd import ops as owc_ops

        self.ops, self.seconds, self.device = owc_ops, seconds, device
        self.a = self.w = self.c = None
        self.runs = []

    def run(self) -> float:
        """One burst.  The 3.2 GB of operands live only for the burst (nothing of the calibration stays resident
        through the timed region; rounds 1-4 had no calibration at all, so the line's `box_calibration.ran` says whether a record's
        timed steps were preceded by one)."""
        from lmms_owc_amd import ops as owc_ops

        g = torch.Generator(device=self.device).manual_seed(4242)
        self.a = torch.randn((self.M, self.K), generator=g, device=self.device, dtype=torch.bfloat16)
        self.w = torch.randn((self.N, self.K), generator=g, device=self.device, dtype=torch.bfloat16) * (self.K ** -0.5)
        self.c = torch.empty((self.M, self.N // 2), device=self.device, dtype=torch.bfloat16)
        try:
            return self._burst_pair()
        finally:
            self.a = self.w = self.c = None     # back to torch's caching allocator (NOT emptied: that would also drop the blocks the warm-up
                                                # steps left for the timed steps)

    def _burst_pair(self) -> float:
        from lmms_owc_amd import ops as owc_ops

        flop = 2.0 * self.M * self.N * self.K

        def burst(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                owc_ops.gemm_bf16(self.a, self.w, epilogue=owc_ops.EPI_SWIGLU, out=self.c)
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) * 1e-3

        burst(8)                                           # the clock settles under load within the first launches
        n = max(16, int(self.seconds / (burst(8) / 8)))
        t = burst(n)
        self.runs.append({"launches": n, "seconds": t, "tflops": n * flop / t / 1e12})
        return self.runs[-1]["tflops"]

    def report(self, images_per_s_per_gpu: float) -> dict:
        tf = [r["tflops"] for r in self.runs]
        mean = float(np.mean(tf)) if tf else 0.0
        return {"ran": bool(tf), "operands_resident_during_timed_region": False,
                "kernel": f"7B gate/up projection + SwiGLU epilogue, M = {self.M}, N = {self.N}, K = {self.K}, random operands, back to back",
                "tflops_before_timed_region": tf[0] if tf else None, "tflops_after_timed_region": tf[1] if len(tf) > 1 else None,
                "tflops": mean, "frac_of_peak": mean / PEAK_BF16_TFLOPS, "runs": self.runs,
                "value_per_calibration_tflops": images_per_s_per_gpu / mean if mean > 0 else None,
                "what": "images/s per GPU divided by the box's own GEMM rate: constant across boxes when a headline difference is the "
                        "box (DESIGN.md section 5 lists the pairs measured so far)"}


def config2_leg(device, T: int, sync, profile, images: int = 512, passes: int = 2) -> dict:
    """BASELINE.json configs[1] (never `value`): Qwen2-VL-2B bf16, 512 images (Caltech-101 --limit 512 at the synthetic 448x448 size),
    one GPU: its own weights and engine beside the headline model's, `passes` timed passes after one warm-up."""
    from lmms_owc_amd import ops as owc_ops
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights
    from lmms_owc_amd.models import imageproc

    d2 = DIMS["qwen2-vl-2b"]
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d2, device, seed=1234))
    gen = torch.Generator(device=device).manual_seed(77)
    u8 = torch.randint(0, 256, (images, 3, 448, 448), generator=gen, device=device, dtype=torch.uint8)
    pix = owc_ops.patchify_u8(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD)
    del u8
    ids = prompt_ids(d2.image_token_id)
    prompts, grids, flat = [ids] * images, [[(1, 32, 32)]] * images, [(1, 32, 32)] * images

    def step():
        return eng.generate(prompts, eng.encode_images(pix, flat), grids, T, eos_token_id=-1, pad_token_id=0).cpu()

    step()
    sync()
    profile(True)
    t0 = time.perf_counter()
    for _ in range(passes):
        out = step()
    sync()
    dt = (time.perf_counter() - t0) / passes
    prof = profile(False)
    solo = eng.generate(prompts[:1], eng.encode_images(pix[:1024], flat[:1]), grids[:1], T, eos_token_id=-1, pad_token_id=0).cpu()
    f = flops_per_image(d2, T)
    per_chunk = max(1, min(images, (eng.prefill_chunk_tokens - S_TEXT_BEFORE) // (S_IMG + S_TEXT_AFTER)))
    f_exec = f - pruned_flops_per_image(d2, per_chunk)
    return {"config": "BASELINE.json configs[1]: Qwen2-VL-2B bf16, 512 synthetic 448x448 images per pass, 1 GPU", "images": images, "new_tokens": T,
            "seconds_per_pass": dt, "images_per_s": images / dt, "model_flops_per_image": f, "executed_flops_per_image": f_exec,
            "mfma_frac_end_to_end": images / dt * f_exec / (PEAK_BF16_TFLOPS * 1e12),
            "batch_invariance_check": bool(torch.equal(solo[0], out[0])),
            "roofline": gemm_roofline(prof["gemm_bf16"], dt * passes, "of the 2B pass")}


def qwen72b_fp8_leg(device, T: int, profile, images: int = 256) -> dict:
    """BASELINE.json configs[4], LMM side (never `value`): Qwen2-VL-72B with the fp8 (e4m3fn, per-token x per-channel scales) decoder,
    `images` synthetic 448x448 images, one warm-up + one timed pass, 73 GB of weights initialised on the device."""
    import dataclasses

    from lmms_owc_amd import ops as owc_ops
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights
    from lmms_owc_amd.models import imageproc

    t_w = time.perf_counter()
    d = dataclasses.replace(DIMS["qwen2-vl-72b"], decoder_dtype="fp8")
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d, device, seed=1234))
    torch.cuda.synchronize()
    t_w = time.perf_counter() - t_w
    gen = torch.Generator(device=device).manual_seed(78)
    u8 = torch.randint(0, 256, (images, 3, 448, 448), generator=gen, device=device, dtype=torch.uint8)
    pix = owc_ops.patchify_u8(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD)
    del u8
    ids = prompt_ids(d.image_token_id)
    prompts, grids, flat = [ids] * images, [[(1, 32, 32)]] * images, [(1, 32, 32)] * images

    def step():
        return eng.generate(prompts, eng.encode_images(pix, flat), grids, T, eos_token_id=-1, pad_token_id=0).cpu()

    step()
    torch.cuda.synchronize()
    profile(True)
    t0 = time.perf_counter()
    out = step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    prof = profile(False)
    solo = eng.generate(prompts[:1], eng.encode_images(pix[:1024], flat[:1]), grids[:1], T, eos_token_id=-1, pad_token_id=0).cpu()
    f = flops_per_image(d, T)
    per_chunk = max(1, min(images, (eng.prefill_chunk_tokens - S_TEXT_BEFORE) // (S_IMG + S_TEXT_AFTER)))
    f_exec = f - pruned_flops_per_image(d, per_chunk)
    g8, g16 = prof["gemm_fp8"], prof["gemm_bf16"]
    tf8 = g8["work"] / (g8["ms"] * 1e-3) / 1e12 if g8["ms"] > 0 else 0.0
    tf16 = g16["work"] / (g16["ms"] * 1e-3) / 1e12 if g16["ms"] > 0 else 0.0
    return {"config": "BASELINE.json configs[4], LMM side: Qwen2-VL-72B, fp8-e4m3 decoder projections (bf16 elsewhere), synthetic 448x448 images, 1 GPU",
            "images": images, "new_tokens": T, "seconds_per_pass": dt, "images_per_s": images / dt, "weights_gb": eng.w.nbytes() / 1e9,
            "weight_init_seconds": t_w, "model_flops_per_image": f, "executed_flops_per_image": f_exec,
            "mfma_frac_end_to_end_vs_bf16_peak": images / dt * f_exec / (PEAK_BF16_TFLOPS * 1e12),
            "batch_invariance_check": bool(torch.equal(solo[0], out[0])),
            "roofline": {"bound": "mfma", "kernel": "gemm_fp8_nt_* (decoder projections, scaled f8f6f4 MFMA)", "achieved": tf8, "peak": PEAK_FP8_TFLOPS,
                         "unit": "TFLOP/s", "frac": tf8 / PEAK_FP8_TFLOPS, "traffic": None, "launches": g8["launches"],
                         "kernel_ms_total": g8["ms"], "share_of_leg_time": g8["ms"] * 1e-3 / time_or(dt)},
            "bf16_gemm_in_same_leg": {"achieved": tf16, "frac": tf16 / PEAK_BF16_TFLOPS, "kernel_ms_total": g16["ms"],
                                      "share_of_leg_time": g16["ms"] * 1e-3 / time_or(dt)}}


def llava_next_34b_leg(batch: int = 16) -> dict:
    """BASELINE.json configs[3] (never `value`): LLaVA-NeXT-34B (CLIP ViT-L/14-336 anyres + Yi-34B dims, 69.5 GB of bf16 weights
    initialised on the device), `batch` synthetic 480x640 images, one warm-up + one timed pass (tools/bench_llava.py)."""
    sys.path.insert(0, str(ROOT / "tools"))
    import bench_llava

    return bench_llava.run("llava-next-34b", batch, steps=1, warmup=1, new_tokens=16, image_size="480x640", text_tokens=48,
                           decoder_dtype="bf16")


def cosine_10k_leg(scorer, device, sync, profile, n_pred: int = 65536, n_cls: int = 10000, k: int = 5, passes: int = 5) -> dict:
    """BASELINE.json configs[4]'s scorer side (never `value`): class-name embedding of a ~10k-class vocabulary on the GPU + cosine
    top-k of `n_pred` predictions against it (the N x C similarity matrix is never materialised)."""
    r = np.random.default_rng(5)
    L = 16
    ids = r.integers(1000, 30000, (n_cls, L)).astype(np.int32)
    lens = r.integers(2, L + 1, n_cls)
    mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int32)
    sync()
    t0 = time.perf_counter()
    cls_z = scorer.embed(ids, mask)
    sync()
    t_cls = time.perf_counter() - t0
    g = torch.Generator(device=device).manual_seed(9)
    z = torch.randn((n_pred, 384), generator=g, device=device, dtype=torch.float32)
    z = z / z.norm(dim=1, keepdim=True)              # synthetic unit-norm prediction embeddings (input preparation, not the path)
    label = torch.from_numpy(r.integers(0, n_cls, n_pred).astype(np.int32)).to(device)
    scorer.topk(z, cls_z, k, label)
    sync()
    profile(True)
    t0 = time.perf_counter()
    for _ in range(passes):
        tv, ti, paired = scorer.topk(z, cls_z, k, label)
    sync()
    dt = (time.perf_counter() - t0) / passes
    ck = profile(False)["cosine_topk"]
    tf = ck["work"] / (ck["ms"] * 1e-3) / 1e12 if ck["ms"] > 0 else 0.0
    byt = passes * (4.0 * 384 * (n_pred + n_cls) + 8.0 * k * n_pred)
    # self-check on a slice (the full N x C matrix is exactly what the kernel avoids): top-1 of 256 predictions against torch
    ref = (z[:256] @ cls_z.T).max(dim=1)
    ok = bool(torch.equal(ref.indices.to(torch.int32), ti[:256, 0])) and bool(torch.allclose(ref.values, tv[:256, 0], atol=2e-6))
    return {"config": "BASELINE.json configs[4], scorer side: on-GPU class-name embed + cosine top-k at a ~10k-class vocabulary",
            "classes": n_cls, "predictions": n_pred, "top_k": k, "class_embed_seconds": t_cls, "class_labels_per_s": n_cls / t_cls,
            "label_cosine_per_sec": n_pred / dt, "top1_matches_dense_matmul_on_256_rows": ok,
            "roofline": {"bound": "mfma", "kernel": "cosine_topk_kernel (C = 10 000)", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s",
                         "frac": tf / 157.3, "traffic": None, "launches": ck["launches"], "kernel_ms_total": ck["ms"],
                         "hbm_gbs_on_algorithmic_bytes": byt / (ck["ms"] * 1e-3) / 1e9 if ck["ms"] > 0 else 0.0,
                         "note": "2 N C D on the f32-input MFMA (157 TF dense f32 matrix peak); algorithmic bytes 4 D (N + C) + 8 k N"}}


EOS_ID = 151645   # <|im_end|>: Qwen2-VL's EOS token id


def ragged_answer_lengths(B: int, cap: int, mean_len: float, cap_frac: float, seed: int):
    """Seeded per-sequence continuations for the `eos_terminated` leg: stop lengths geometric with mean `mean_len` (the answer + its
    EOS; classification answers are a few tokens), `cap_frac` of the sequences never stop inside the cap (a rambling answer), EOS at
    the stop column of the forced continuation."""
    r = np.random.default_rng(seed)
    lens = np.minimum(r.geometric(1.0 / mean_len, B), cap)
    lens[r.random(B) < cap_frac] = cap + 1
    forced = r.integers(1000, 150000, (B, cap)).astype(np.int32)
    for b in np.flatnonzero(lens <= cap):
        forced[b, lens[b] - 1] = EOS_ID
    return forced, lens


def eos_terminated_leg(engine, pix, flat_grids, prompts, grids, B: int, sync, headline_images_per_s: float, rank: int,
                       caps=(64, 256), mean_len: float = 8.0, cap_frac: float = 0.01, plain_caps=(64,), hand_over_passes: int = 3) -> dict:
    """Ragged answer lengths (never `value`).  In the reference every image is its own `generate` call that stops at its own EOS
    (/root/reference/src/models/_qwen2_vl.py:319-337; `max_new_tokens` 64 in src/data/tasks/_classification/caltech101/base.yaml:9-11,
    256 in the zero_shot_cot / llava_cot / llamav_o1 YAMLs).  The same B images as the main leg, EOS handling ON, seeded stop
    lengths injected through the forced-token path: one pass with the finished rows dropped from the following decode steps
    (`compact_rows`, the default) and one where all B rows decode until the last sequence stops - same tokens, bit for bit."""
    out = {"answer_lengths": f"geometric, mean {mean_len} tokens incl. EOS, {cap_frac:.0%} of the sequences never stop inside the cap; seeded per rank",
           "headline_images_per_s_forced_16": headline_images_per_s, "by_cap": []}
    d = engine.d
    for cap in caps:
        kv_elems = d.n_layers * B * d.n_kv_heads * d.head_dim * (len(prompts[0]) + cap)
        kv_bytes = 4.0 * kv_elems
        need = 1.15 * kv_bytes + (24 << 30)      # the two cache blocks + embeddings, workspace, logits of the pass
        held = 4 * engine._kv[0].numel() if getattr(engine, "_kv", None) else 0   # (the engine's current pair is released when it grows)
        if need > torch.cuda.mem_get_info()[0] + held:
            torch.cuda.empty_cache()
        if need > torch.cuda.mem_get_info()[0] + held:
            out["by_cap"].append({"max_new_tokens": cap, "skipped": f"KV cache of {kv_bytes / 2**30:.0f} GiB does not fit beside the weights at this batch"})
            continue
        # the K / V caches of this cap are larger than the main leg's: the engine's grow-only pair is grown once, outside the timed
        # passes (a task pays this once, whatever its length; measured: up to 1 s of hipMalloc for 2 x 20 GB)
        engine.reserve_kv(kv_elems)

        forced, lens = ragged_answer_lengths(B, cap, mean_len, cap_frac, 4242 + rank)
        row = {"max_new_tokens": cap, "mean_answer_tokens": float(np.minimum(lens, cap).mean()), "sequences_at_cap": int((lens > cap).sum())}
        toks = {}
        for compact in ((True, False) if cap in plain_caps else (True,)):
            st = {}
            sync()
            t0 = time.perf_counter()
            emb = engine.encode_images(pix, flat_grids)
            toks[compact] = engine.generate(prompts, emb, grids, cap, eos_token_id=EOS_ID, pad_token_id=0, forced_tokens=forced,
                                            compact_rows=compact, stats=st).cpu()
            sync()
            dt = time.perf_counter() - t0
            live = st["live_rows_per_step"]
            row["compacted" if compact else "all_rows_every_step"] = {
                "seconds": dt, "images_per_s": B / dt, "vs_headline": B / dt / headline_images_per_s, "decode_steps_run": len(live) - 1,
                "decode_loop_seconds": st["decode_events"][0].elapsed_time(st["decode_events"][1]) * 1e-3,
                "row_steps": int(sum(live[1:])), "live_rows_at_step": {str(j): int(live[j]) for j in (1, 4, 8, 16, 32, 63, 128, 255) if j < len(live)}}
            del emb
        if len(toks) == 2:
            row["tokens_identical_with_and_without_compaction"] = bool(torch.equal(toks[True], toks[False]))
        avail = torch.cuda.mem_get_info()[0] + torch.cuda.memory_reserved() - torch.cuda.memory_allocated() + 4 * engine._kv[0].numel()
        # slots for carried sequences, as `PassPipeline._reserve_kv` sizes them: half a pass if the cache with them plus their export
        # copy fits with room to spare, else an eighth
        capacity = next((c for c in (B // 2, B // 8) if (1.0 + 2.0 * c / B) * 1.05 * kv_bytes + (24 << 30) <= avail), None)
        if cap == max(caps) and hand_over_passes > 1 and capacity is not None:
            from lmms_owc_amd.models._base import hand_over_below

            capacity = (max(capacity, 256) + 255) // 256 * 256
            slots = B + capacity                                    # (the engine's cache pair grows once more, outside the timing)
            rows_kv = (len(prompts[0]) + cap + 2 + 15) // 16 * 16
            engine.reserve_kv(d.n_layers * slots * d.n_kv_heads * d.head_dim * rows_kv)
            # a TASK is several passes: with straggler hand-over a pass stops once its own live sequences are few and the rest finish
            # inside the following passes (`generate(..., carry=)`), so the long tail is paid once per task, not once per pass
            ref = toks[True].numpy()
            state, got, handed, below = None, {}, [], []
            sync()
            t0 = time.perf_counter()
            for k in range(hand_over_passes):
                n_in = 0 if state is None else len(state["tags"])
                c = {"in": state, "below": hand_over_below(B, n_in, capacity) if k + 1 < hand_over_passes else 0, "slots": capacity, "tags": [(k, i) for i in range(B)]}
                below.append(c["below"])
                emb = engine.encode_images(pix, flat_grids)
                out_k = engine.generate(prompts, emb, grids, cap, eos_token_id=EOS_ID, pad_token_id=0, forced_tokens=forced, carry=c).cpu().numpy()
                skip = set(c["unfinished_rows"])
                got.update({(k, i): out_k[i] for i in range(B) if i not in skip})
                got.update(dict(c["finished"]))
                state = c["out"]
                handed.append(0 if state is None else len(state["tags"]))
                del emb
            sync()
            dt = time.perf_counter() - t0
            same = len(got) == hand_over_passes * B and all(np.array_equal(got[(k, i)], ref[i]) for k in range(hand_over_passes) for i in range(0, B, 7))
            row["passes_with_straggler_hand_over"] = {
                "passes": hand_over_passes, "seconds": dt, "images_per_s": hand_over_passes * B / dt,
                "vs_headline": hand_over_passes * B / dt / headline_images_per_s, "sequences_handed_to_the_next_pass": handed,
                "hand_over_when_own_live_sequences_at_most": below, "slots_for_carried_sequences": capacity,
                "tokens_identical_to_a_pass_run_alone": bool(same)}
        pad_ok = all(bool((toks[True][b, min(int(lens[b]), cap):] == 0).all()) for b in range(0, B, max(1, B // 64)))
        row["pad_behind_stop_column"] = pad_ok
        out["by_cap"].append(row)
    return out


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: this (parent) process has not touched the GPU; it starts N children
    of the same command line with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set (one process per GPU, the analogue of the
    reference's `accelerate launch --num_processes N`, scripts/schedule_batch.sh:109-112), waits for all of them and
    returns the first non-zero exit code.  Rank 0 inherits stdout, so its ONE JSON line is this command's output; the other
    ranks' stdout goes to stderr.  No exec: the children are ordinary subprocesses."""
    import signal
    import socket
    import subprocess

    def free_port() -> int:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            return sk.getsockname()[1]

    procs: list = []

    def stop_children(sig=signal.SIGTERM) -> None:
        for q in procs:   # every rank leads its own process group (start_new_session): its prep threads / pools go with it
            if q.poll() is None:
                try:
                    os.killpg(q.pid, sig)
                except (ProcessLookupError, PermissionError):
                    pass

    def on_signal(signum, _frame):   # a driver timeout SIGTERMs this parent: never orphan N ranks that hold the GPUs
        stop_children(signal.SIGTERM)
        time.sleep(1.0)
        stop_children(signal.SIGKILL)
        raise SystemExit(128 + signum)

    for sg in (signal.SIGTERM, signal.SIGHUP, signal.SIGINT):
        signal.signal(sg, on_signal)
    rc = 0
    for attempt in range(3):   # the port is picked by bind-then-close: if another job takes it before rank 0 binds, try a new one
        port = free_port()
        procs.clear()
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                       MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OWC_BENCH_SELF_LAUNCHED="1")
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                          stdout=None if r == 0 else sys.stderr, start_new_session=True))
        rc = 0
        try:
            live = dict(enumerate(procs))
            while live:   # poll ALL ranks: a rank that died must not leave the others waiting in a collective
                for r, p in list(live.items()):
                    code = p.poll()
                    if code is None:
                        continue
                    del live[r]
                    if code != 0 and rc == 0:
                        rc = code
                        print(f"[bench] rank {r} exited with {code}; stopping the other ranks", file=sys.stderr)
                        stop_children(signal.SIGTERM)
                time.sleep(0.2)
        finally:
            stop_children(signal.SIGKILL)
        if rc != RENDEZVOUS_FAILED:
            break
        print(f"[bench] rendezvous on port {port} failed (attempt {attempt + 1}); retrying on a new port", file=sys.stderr)
    return rc


def dry_run(args, world: int, rank: int) -> None:
    """Launcher / rendezvous check that stops before HIP init: gloo all_reduce over the ranks, rank 0 prints a stub line."""
    seen = 1
    if os.environ.get("OWC_BENCH_DRYRUN_FAIL_RANK") == str(rank):   # test hook: exit-code propagation of the launcher
        raise SystemExit(7)
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        seen = int(t.item())
        assert seen == dist.get_world_size() == world
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": args.gpus, "world_size_seen": seen, "steps": args.steps,
                          "warmup": args.warmup, "self_launched": bool(os.environ.get("OWC_BENCH_SELF_LAUNCHED"))}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="7b", choices=["2b", "7b", "72b", "2.5-7b", "2.5-3b"],
                    help="Qwen2-VL size (default 7b = the model the metric is quoted on); 2.5-* = Qwen2.5-VL (window-attention vision tower)")
    ap.add_argument("--batch", type=int, default=2048, help="images per GPU per step")
    ap.add_argument("--new-tokens", type=int, default=16)
    ap.add_argument("--decoder-dtype", default="bf16", choices=["bf16", "fp8"],
                    help="fp8 = e4m3fn decoder projections (BASELINE config #5; never the default: the headline is bf16)")
    ap.add_argument("--scorer-labels", type=int, default=65536)
    ap.add_argument("--scorer-classes", type=int, default=397)
    ap.add_argument("--vit-chunk", type=int, default=None, help="vision-tower tokens per launch group (engine default if unset)")
    ap.add_argument("--prefill-chunk", type=int, default=None, help="packed prefill rows per launch group (engine default if unset)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pil-leg", action="store_true", help="skip the PIL -> generate_until -> strings leg")
    ap.add_argument("--no-decode-leg", action="store_true", help="skip the HBM-regime decode leg (decode step at batch 1 / 32 / 128)")
    ap.add_argument("--no-eos-leg", action="store_true", help="skip the EOS-terminated ragged-answer-length leg (max_new_tokens 64 / 256)")
    ap.add_argument("--eos-mean-len", type=float, default=8.0,
                    help="mean answer length (tokens incl. EOS, geometric) of the EOS-terminated leg: 8 = classification answers; ~100 = chain-of-thought answers")
    ap.add_argument("--leg-budget-s", type=float, default=420.0,
                    help="seconds since process start after which the remaining OPTIONAL legs (PIL, HBM-regime decode) are skipped with a "
                         "note, so that a long --steps run still ends within minutes; the timed region, the EOS / image-size / other-config "
                         "legs and the CPU baseline always run")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the other configs' short legs (Food-101 image sizes, Qwen2-VL-2B / 512 images, label-cosine at C = 10 000)")
    ap.add_argument("--image-sizes", default="config3", choices=sorted(DATASET_SIZES) + ["none"],
                    help="extra leg (never `value`): images of the dataset's real size distribution through smart_resize "
                         "(64...1024 image tokens per image, ragged vision / prefill launch groups); reports images/s and image-tokens/s; "
                         "config3 = BASELINE.json configs[2]'s mixture Food-101 : DTD : Flowers-102 = 30300 : 1692 : 2463")
    ap.add_argument("--ragged-images", type=int, default=1024, help="images of the --image-sizes leg")
    ap.add_argument("--cap-images", type=int, default=256,
                    help="images of the `max_pixels_images` leg (every image on the reference's max_pixels cap: ~1024 image tokens, "
                         "4096 patches - the SUN397 / Stanford-Cars regime); 0 skips it")
    ap.add_argument("--no-big-legs", action="store_true",
                    help="skip the Qwen2-VL-72B fp8 (256 images) and LLaVA-NeXT-34B (16 images) legs that run after the CPU baseline on one GPU")
    ap.add_argument("--big-leg-budget-s", type=float, default=600.0,
                    help="seconds since process start after which a big-model leg is not started any more")
    ap.add_argument("--shape-table", default=None, metavar="CSV",
                    help="write the timed region's bf16 GEMM launches grouped by (M, N, K, epilogue) - launches, total / avg / min / max "
                         "time, TFLOP/s - to this CSV (the line's roofline.by_shape holds the first 12 rows)")
    ap.add_argument("--total-budget-s", type=float, default=840.0,
                    help="seconds since process start after which a still-running big-model leg is abandoned: the line is printed "
                         "with that leg marked skipped and the process exits 0")
    ap.add_argument("--no-calibration", action="store_true", help="skip the box calibration GEMM bursts around the timed region")
    ap.add_argument("--nominal-forward", action="store_true",
                    help="run the model's nominal forward: full last prefill layer on every row and no shared-prefix segment "
                         "(same tokens bit for bit; shows what the two dead-work eliminations are worth)")
    ap.add_argument("--cpu-images", type=int, default=8,
                    help="images of the CPU baseline (HF generate on the CPUs the cgroup grants: ~3.5 s each for 7B on 16; rounds 1-5 "
                         "ran 128 throttled threads at 25-40 s each; SURVEY.md section 8d asks for 4 for 7B: 8 = ~30 s of CPU work and 128 "
                         "teacher-forced steps for the full-size parity gate); the first is the warm-up, `value` = mean of the rest")
    ap.add_argument("--cpu-budget-s", type=float, default=150.0,
                    help="the CPU baseline stops after the image that crosses this many seconds of generate time (at least two images run)")
    ap.add_argument("--tune", action="append", default=[], metavar="KNOB=VALUE",
                    help="owc_tuning_set(KNOB, VALUE) before anything runs (A-B experiments; recorded in config.tuning)")
    ap.add_argument("--dry-run", action="store_true", help="launcher + rendezvous check on CPU (gloo); stops before HIP init")
    ap.add_argument("--one-gpu-value", type=float, default=None,
                    help="images/s of the 1-GPU run: adds scaling_efficiency = value / (N * this) to the line")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))   # nothing above this line has touched the GPU
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    if args.dry_run:
        return dry_run(args, world, rank)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    # test hook (never set by the driver): OWC_BENCH_SHARE_GPU=1 lets the ranks of a multi-rank run share cuda:0 with gloo
    # collectives, so that the N > 1 code path can be exercised on a 1-GPU box (RCCL refuses two ranks on one device)
    share = os.environ.get("OWC_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    cdev = torch.device("cpu") if share else device   # where the timing scalars of the collectives live
    dist = None
    if world > 1:
        import torch.distributed as dist

        try:
            if share:
                dist.init_process_group("gloo")
            else:
                dist.init_process_group("nccl", device_id=device)  # RCCL over xGMI
        except (RuntimeError, OSError) as e:   # e.g. EADDRINUSE on the rendezvous port
            print(f"[bench] rank {rank}: rendezvous failed: {e}", file=sys.stderr)
            if os.environ.get("OWC_BENCH_SELF_LAUNCHED"):
                raise SystemExit(RENDEZVOUS_FAILED)
            raise

    from lmms_owc_amd import _lib
    from lmms_owc_amd import build as owc_build

    if rank == 0 or not _lib.lib_path().exists():
        if local == 0:
            owc_build.build(verbose=False)
    if dist is not None:
        dist.barrier()
    for kv in args.tune:
        name, val = kv.split("=")
        _lib.check(_lib.load().owc_tuning_set(name.encode(), int(val)), 0)
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights
    from lmms_owc_amd.engine.scorer import MINILM_L6, BertWeights, SentenceScorer

    key = f"qwen2.5-vl-{args.model[4:]}" if args.model.startswith("2.5-") else f"qwen2-vl-{args.model}"
    dims = DIMS[key]
    if dims.v_variant == 1:
        args.no_cpu_baseline = True   # the CPU leg builds HF's Qwen2-VL class; Qwen2.5-VL parity lives in tests/ (HF goldens)
    if args.decoder_dtype != "bf16":
        import dataclasses

        dims = dataclasses.replace(dims, decoder_dtype=args.decoder_dtype)
    weights = Qwen2VLWeights.random(dims, device, seed=1234)
    ekw = {k: v for k, v in (('vit_chunk_tokens', args.vit_chunk), ('prefill_chunk_tokens', args.prefill_chunk)) if v}
    if args.nominal_forward:
        ekw['share_prefix'] = False
        _lib.check(_lib.load().owc_tuning_set(b"prefill_prune_last", 0), 0)
    engine = Qwen2VLEngine(weights, **ekw)
    B, T = args.batch, args.new_tokens
    # SURVEY.md section 8(d) synthetic input: uint8 uniform [0, 255] images, seeded per rank, CLIP-normalised and patchified
    # on the GPU (owc_patchify_u8) OUTSIDE the timed region: the step starts from pixel_values resident in HBM
    from lmms_owc_amd import ops as owc_ops
    from lmms_owc_amd.models import imageproc

    gen = torch.Generator(device=device).manual_seed(1234 + rank)
    pix = torch.empty((B * 1024, 1176), device=device, dtype=torch.bfloat16)
    for i0 in range(0, B, 256):   # in groups: the uint8 source of 2048 images is 1.2 GB, no need to hold it
        n_i = min(256, B - i0)
        u8 = torch.randint(0, 256, (n_i, 3, 448, 448), generator=gen, device=device, dtype=torch.uint8)
        owc_ops.patchify_u8(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD, out=pix[i0 * 1024:(i0 + n_i) * 1024])
    del u8
    ids = prompt_ids(dims.image_token_id)
    prompts = [ids] * B
    grids = [[(1, 32, 32)]] * B
    flat_grids = [(1, 32, 32)] * B

    def step():
        emb = engine.encode_images(pix, flat_grids)
        toks = engine.generate(prompts, emb, grids, T, eos_token_id=-1, pad_token_id=0)
        return toks.cpu()  # last decoded token ids on the host

    def sync():
        torch.cuda.synchronize()
        t_own = time.perf_counter()      # this rank's own finish time, BEFORE it waits for the slowest rank
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        return t_own

    lib, ctx = _lib.load(), _lib.ctx(local)
    NK = len(_lib.PROF_KINDS)
    leg_seconds, lap_t = {}, [time.perf_counter()]
    leg_seconds["setup_weights_inputs"] = round(lap_t[0] - T_PROCESS_START, 1)

    def lap(name: str) -> None:
        """Wall seconds of every part of this run (rank 0's clock), reported as `leg_seconds`."""
        now = time.perf_counter()
        leg_seconds[name] = round(now - lap_t[0], 1)
        lap_t[0] = now

    def over_budget() -> bool:
        """Rank 0 decides for all ranks (the legs hold collectives)."""
        flag = torch.tensor([1.0 if time.perf_counter() - T_PROCESS_START > args.leg_budget_s else 0.0], device=cdev, dtype=torch.float64)
        if dist is not None:
            dist.broadcast(flag, 0)
        return bool(flag.item() > 0)

    def read_profile() -> dict:
        ms, wk, n = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
        _lib.check(lib.owc_profile_read(ctx, NK, ms, wk, n), local)
        return {k: {"ms": ms[i], "work": wk[i], "launches": int(n[i])} for i, k in enumerate(_lib.PROF_KINDS)}

    def read_shape_table(peak_tflops: float) -> list:
        """The bf16 GEMM launches of the profile just read, one row per (M, N, K, epilogue): what `roofline.achieved` is the sum of.
        A reader can redo the arithmetic: tflops = 2 M N K / avg_us / 1e6; sum(launches * 2MNK) / sum(total_us) = roofline.achieved."""
        cap = 512
        shp, st = (C.c_int32 * (4 * cap))(), (C.c_double * (4 * cap))()
        cnt = lib.owc_profile_shapes(ctx, cap, shp, st)
        epi = {0: "none", 1: "quick_gelu", 2: "gelu_erf", 3: "residual", 4: "swiglu", 5: "f32", 6: "vision_rope"}
        rows = []
        for i in range(max(0, min(cnt, cap))):
            m, n_, k, e = (int(shp[4 * i + j]) for j in range(4))
            launches, tot, mn, mx = int(st[4 * i]), st[4 * i + 1], st[4 * i + 2], st[4 * i + 3]
            flop = 2.0 * m * n_ * k
            rows.append({"M": m, "N": n_, "K": k, "epilogue": epi.get(e, str(e)), "launches": launches, "total_ms": tot,
                         "avg_us": tot / launches * 1e3, "min_us": mn * 1e3, "max_us": mx * 1e3, "flop_per_launch": flop,
                         "tflops": flop * launches / (tot * 1e-3) / 1e12 if tot > 0 else 0.0,
                         "frac_of_peak": flop * launches / (tot * 1e-3) / 1e12 / peak_tflops if tot > 0 else 0.0})
        rows.sort(key=lambda r: -r["total_ms"])
        return rows

    for _ in range(args.warmup):
        step()
    sync()
    calib = None
    if not args.no_calibration and args.model == "7b":   # (the yardstick is a 7B shape; every rank runs it: the ranks stay in step)
        try:
            calib = BoxCalibration(device)
            calib.run()
        except torch.OutOfMemoryError:
            calib = None
        sync()
    lib.owc_gemm_profile_enable(ctx, 1 if rank == 0 else 0)
    t0 = time.perf_counter()
    step_end = []
    for _ in range(args.steps):
        out = step()
        step_end.append(time.perf_counter())   # (a step ends with its token ids on the host: no extra synchronisation)
    t_own = sync()
    dt_local = time.perf_counter() - t0          # barrier-bracketed: what `value` is computed from (max over ranks)
    dt_own = t_own - t0                          # without the wait for the other ranks: the per-rank rates
    prof = read_profile() if rank == 0 else None
    shape_table = read_shape_table(PEAK_BF16_TFLOPS) if rank == 0 else None
    lib.owc_gemm_profile_enable(ctx, 0)
    if calib is not None:
        calib.run()
        sync()
    if rank == 0 and args.shape_table and shape_table:
        import csv

        with open(args.shape_table, "w", newline="") as fh:
            wr = csv.writer(fh)
            wr.writerow(["M", "N", "K", "epilogue", "launches", "total_ms", "avg_us", "min_us", "max_us", "flop_per_launch", "tflops", "frac_of_peak"])
            for r in shape_table:
                wr.writerow([r["M"], r["N"], r["K"], r["epilogue"], r["launches"], f"{r['total_ms']:.4f}", f"{r['avg_us']:.2f}", f"{r['min_us']:.2f}",
                             f"{r['max_us']:.2f}", f"{r['flop_per_launch']:.6g}", f"{r['tflops']:.1f}", f"{r['frac_of_peak']:.4f}"])
    lap("warmup_and_timed_steps")
    fp8_run = args.decoder_dtype == "fp8"
    assert out.shape == (B, T)
    tall = torch.tensor([dt_local, dt_own], device=cdev, dtype=torch.float64)
    per_rank_dt, per_rank_own = [dt_local], [dt_own]
    if dist is not None:
        gathered = torch.empty(2 * world, device=cdev, dtype=torch.float64)
        dist.all_gather_into_tensor(gathered, tall)
        per_rank_dt, per_rank_own = gathered.view(world, 2)[:, 0].tolist(), gathered.view(world, 2)[:, 1].tolist()
    dt = max(per_rank_dt)                       # max over ranks
    images_per_s = world * B * args.steps / dt

    # ---- outside the timed region: image 0's tokens inside the B-image batch == the same image run alone (batch invariance
    # across the chunked vision / prefill launch groups and the batch-size dependent decode kernels)
    emb0 = engine.encode_images(pix[:1024], flat_grids[:1])
    alone = engine.generate(prompts[:1], emb0, grids[:1], T, eos_token_id=-1, pad_token_id=0).cpu()
    invariant = bool(torch.equal(alone[0], out[0]))

    # ---- EOS-terminated leg (never `value`): seeded ragged answer lengths at the task configs' caps, with / without row compaction
    eos_leg = None
    if not args.no_eos_leg and T >= 2:
        try:
            eos_leg = eos_terminated_leg(engine, pix, flat_grids, prompts, grids, B, sync, B * args.steps / dt_own, rank,
                                         mean_len=args.eos_mean_len)
        except torch.OutOfMemoryError as e:   # (an extra leg must never sink the measurement: large models at the largest batch)
            torch.cuda.empty_cache()
            eos_leg = {"skipped": f"out of memory: {str(e)[:160]}", "by_cap": []}
        if dist is not None:   # whole-job rate of every pass = all ranks' images / the slowest rank's time
            for row in eos_leg["by_cap"]:
                for k in ("compacted", "all_rows_every_step", "passes_with_straggler_hand_over"):
                    if k in row:
                        t = torch.tensor([row[k]["seconds"]], device=cdev, dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        row[k]["seconds"] = float(t.item())
                        row[k]["images_per_s"] = world * B * row[k].get("passes", 1) / row[k]["seconds"]
                        row[k]["vs_headline"] = row[k]["images_per_s"] / images_per_s
            eos_leg["headline_images_per_s_forced_16"] = images_per_s

    lap("eos_terminated")

    def profile(on: bool):
        """Leg-level kernel profile: start (True) / stop and read (False) the library's per-launch-class HIP-event recording."""
        if on:
            lib.owc_gemm_profile_enable(ctx, 1)
            return None
        p = read_profile()
        lib.owc_gemm_profile_enable(ctx, 0)
        return p

    ragged = None
    if args.image_sizes != "none" and not (args.no_extra_legs and args.image_sizes == "food101"):
        ragged = ragged_leg(engine, dims, args.image_sizes, args.ragged_images, T, 1, device, sync, profile=profile)
        if dist is not None:
            t = torch.tensor([ragged["seconds_per_pass"]], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ragged["seconds_per_pass"] = float(t.item())
            ragged["images_per_s"] = world * ragged["images"] / ragged["seconds_per_pass"]

    lap("real_image_sizes")

    # ---- every image on the max_pixels cap (never `value`): P = 4096 patches, ~1024 image tokens, S ~ 1054 - vision attention is O(P^2)
    cap_leg = None
    if args.cap_images > 0 and not args.no_extra_legs:
        try:
            cap_leg = ragged_leg(engine, dims, "max_pixels", args.cap_images, T, 1, device, sync, profile=profile)
        except torch.OutOfMemoryError as e:
            torch.cuda.empty_cache()
            cap_leg = {"skipped": f"out of memory: {str(e)[:160]}", "deterministic_and_batch_invariant": True}
        if dist is not None and "seconds_per_pass" in cap_leg:
            t = torch.tensor([cap_leg["seconds_per_pass"]], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            cap_leg["seconds_per_pass"] = float(t.item())
            cap_leg["images_per_s"] = world * cap_leg["images"] / cap_leg["seconds_per_pass"]
    lap("max_pixels_images")

    # ---- PCIe-inclusive leg (never `value`): the same step fed from host uint8 images (what the boundary hands over in a
    # real run): pinned H2D copy + GPU rescale/normalise/patchify + the step above
    host_u8 = torch.randint(0, 256, (B, 3, 448, 448), dtype=torch.uint8).pin_memory()
    sync()
    p0 = time.perf_counter()
    dev_u8 = host_u8.to(device, non_blocking=True)
    pix_h = owc_ops.patchify_u8(dev_u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD)
    emb_h = engine.encode_images(pix_h, flat_grids)
    engine.generate(prompts, emb_h, grids, T, eos_token_id=-1, pad_token_id=0).cpu()
    sync()
    pcie_dt = torch.tensor([time.perf_counter() - p0], device=cdev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(pcie_dt, op=dist.ReduceOp.MAX)
    pcie_images_per_s = world * B / float(pcie_dt.item())
    del dev_u8, pix_h, emb_h
    lap("from_host_uint8")

    # ---- real-boundary leg (never `value`): PIL images -> Qwen2VL.generate_until -> strings, i.e. the reference's plug-in
    # contract end to end (JPEG round trip + bicubic resize + tokenise on the host worker pool, double-buffered against the GPU)
    pil = None
    if not args.no_pil_leg and over_budget():
        pil = {"skipped": f"--leg-budget-s {args.leg_budget_s:.0f} s reached before this leg (long --steps run); run it with --steps 3"}
    elif not args.no_pil_leg:
        pil = pil_leg(engine, dims, host_u8, B, T, device, sync)
        if dist is not None:
            t = torch.tensor([pil["seconds"]], device=cdev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pil["seconds"] = float(t.item())
        pil["images_per_s"] = world * pil["images"] / pil["seconds"]
    del host_u8
    lap("from_pil")

    # ---- HBM-regime decode leg (never `value`): ms per decode step at the reference's batch size and mid batches
    decode_leg = None
    if not args.no_decode_leg and over_budget():
        decode_leg = {"skipped": f"--leg-budget-s {args.leg_budget_s:.0f} s reached before this leg (long --steps run); run it with --steps 3"}
    elif rank == 0 and not args.no_decode_leg:
        decode_leg = decode_regime_leg(engine, dims)
    if dist is not None:
        dist.barrier()
    lap("roofline_decode")

    # ---- scorer leg: label-cosine/s (embed predictions + cosine top-5 against resident class embeddings)
    n_lab, L = args.scorer_labels, 16
    scorer = SentenceScorer(BertWeights.random(MINILM_L6, device, seed=7), max_batch=16384)
    # tokenised labels arrive on the HOST (a tokenizer's output): ids [n, 16], lengths uniform 2..16, right-padded
    lr = np.random.default_rng(99 + rank)
    lab_ids = lr.integers(1000, 30000, (n_lab, L)).astype(np.int32)
    lens = lr.integers(2, L + 1, n_lab)
    lab_mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int32)
    cls_z = scorer.embed(lab_ids[: args.scorer_classes], lab_mask[: args.scorer_classes])
    label = torch.from_numpy(lr.integers(0, args.scorer_classes, n_lab).astype(np.int32)).to(device)

    def score():
        z = scorer.embed(lab_ids, lab_mask)
        return scorer.topk(z, cls_z, 5, label)

    score()
    sync()
    lib.owc_gemm_profile_enable(ctx, 1 if rank == 0 else 0)
    s0 = time.perf_counter()
    for _ in range(args.steps):
        tv, ti, paired = score()
    sync()
    sdt = torch.tensor([time.perf_counter() - s0], device=cdev, dtype=torch.float64)
    sprof = read_profile() if rank == 0 else None
    lib.owc_gemm_profile_enable(ctx, 0)
    if dist is not None:
        dist.all_reduce(sdt, op=dist.ReduceOp.MAX)
    labels_per_s = world * n_lab * args.steps / float(sdt.item())
    lap("label_cosine")
    label_tokens = int(lens.sum())

    # ---- the other BASELINE configs in front of the driver (never `value`): configs[1] and the scorer side of configs[4]
    cfg2 = cos10k = None
    if not args.no_extra_legs and rank == 0:
        cos10k = cosine_10k_leg(scorer, device, sync if dist is None else torch.cuda.synchronize, profile)
        if args.model != "2b":
            cfg2 = config2_leg(device, T, sync if dist is None else torch.cuda.synchronize, profile)
    if dist is not None:
        dist.barrier()
    lap("config2_and_cosine_10k")

    parity_failure = None
    if rank == 0:
        f_model = flops_per_image(dims, T)
        # executed FLOPs: what every utilisation figure below is priced on
        s_prompt = S_TEXT_BEFORE + S_IMG + S_TEXT_AFTER
        per_chunk = max(1, min(B, (engine.prefill_chunk_tokens - S_TEXT_BEFORE) // (s_prompt - S_TEXT_BEFORE))) if engine.share_prefix else 1
        f_img = f_model if args.nominal_forward else f_model - pruned_flops_per_image(dims, per_chunk)
        g = prof["gemm_fp8" if fp8_run else "gemm_bf16"]
        gemm_tflops = g["work"] / (g["ms"] * 1e-3) / 1e12 if g["ms"] > 0 else 0.0
        peak = PEAK_FP8_TFLOPS if fp8_run else PEAK_BF16_TFLOPS
        traffic = None if fp8_run else load_traffic(args, engine, B)
        result = {
            "metric": f"images/sec (whole node) {'Qwen2.5-VL-' + args.model[4:].upper() if args.model.startswith('2.5-') else 'Qwen2-VL-' + args.model.upper()} open-world classify; label-cosine/sec",
            "value": images_per_s, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16" if not fp8_run else "fp8-e4m3 decoder projections (per-token / per-channel scales), bf16 elsewhere", "data": "synthetic",
            # (the driver's record keeps the scalars of `config` and `roofline` and the first 120 characters of a string)
            "config": {"workload": f"Qwen2-VL-{args.model.upper()}{' (fp8 decoder)' if fp8_run else ''}: {B} synthetic 448x448 images/GPU/step, timed from pixel_values "
                                   f"RESIDENT IN HBM, S=286, {T} forced greedy tokens; (1024 patches -> 256 image tokens per image, seeded random "
                                   "weights of the real architecture; images strided across ranks, no data-path collective; the same step fed "
                                   "from host uint8 / PIL images: images_per_s_from_host_uint8 / images_per_s_from_pil below)",
                       "images_per_gpu_per_step": B, "prompt_tokens": 286, "new_tokens": T, "parallelism": f"dp{world}",
                       "timed_region_starts_from": "pixel_values resident in HBM",
                       "images_per_s_from_host_uint8": pcie_images_per_s,
                       "images_per_s_from_pil": pil.get("images_per_s") if pil else None,
                       **({"tuning": args.tune} if args.tune else {})},
            "ms_of_each_step": [round((b - a) * 1e3, 1) for a, b in zip([t0] + step_end[:-1], step_end)],   # rank 0: drift under sustained load shows here
            "per_rank_images_per_s": [B * args.steps / t for t in per_rank_own],   # each rank's own clock, before the closing barrier
            "rccl_world_size": dist.get_world_size() if dist is not None else 1,
            "batch_invariance_check": "ok: image 0's tokens inside the batch == the same image run alone" if invariant else
                                      "MISMATCH: image 0's tokens differ between the batch and a single-image run",
            "images_per_s_from_host_uint8": pcie_images_per_s,  # PCIe-inclusive (H2D + GPU patchify + step), not `value`
            "images_per_s_from_pil": pil,                       # the plug-in boundary end to end, not `value`
            "label_cosine_per_sec": labels_per_s,
            "label_cosine_config": {"labels_per_gpu": n_lab, "tokens_per_label": f"uniform 2..{L} (mean {label_tokens / n_lab:.2f}), padded rows are not computed",
                                    "classes": args.scorer_classes, "top_k": 5,
                                    "encoder": "MiniLM-L6 (BERT 6x384) fp32; linears as 3-piece bf16 splits on the bf16 MFMA (fp32-level error, tests/test_scorer_gpu.py)"},
            "model_flops_per_image": f_model,      # the model's nominal forward
            "executed_flops_per_image": f_img,     # minus the last prefill layer's dead rows (see pruned_flops_per_image)
            "mfma_frac_end_to_end": images_per_s / world * f_img / (PEAK_BF16_TFLOPS * 1e12),  # always against the bf16 peak
            "roofline": {"bound": "mfma", "kernel": "gemm_fp8_nt_256_kernel (decoder projections)" if fp8_run else "gemm_bf16_nt_kernel (all epilogues)",
                         "achieved": gemm_tflops, "peak": peak, "unit": "TFLOP/s", "frac": gemm_tflops / peak,
                         "traffic": traffic["bytes_per_launch"] if traffic else None, "traffic_source": traffic,
                         "launches": g["launches"], "kernel_ms_total": g["ms"], "share_of_step_time": g["ms"] * 1e-3 / time_or(dt),
                         "method": "HIP events around every launch of the timed region on the launch stream; achieved = sum(2MNK) / sum(t)",
                         # the 12 shapes that hold the most time (the full table: --shape-table FILE); `shapes` = how many there are
                         "by_shape": [{k: (round(v, 3) if isinstance(v, float) else v) for k, v in r.items() if k != "flop_per_launch"}
                                      for r in (shape_table or [])[:12]],
                         "shapes": len(shape_table or []),
                         "by_shape_covers_ms": sum(r["total_ms"] for r in (shape_table or [])[:12])},
            "roofline_attention": attention_rooflines(prof, dims, B, T, args.steps, dt),
            "roofline_decode": decode_leg,
            "eos_terminated": eos_leg,
            "config2_qwen2vl_2b": cfg2,
            "label_cosine_10k_classes": cos10k,
            "roofline_label_cosine": scorer_rooflines(sprof, n_lab, args.scorer_classes, 5, args.steps, float(sdt.item())),
        }
        if calib is not None and calib.runs:
            cal = calib.report(images_per_s / world)
            result["roofline"]["box_calibration_tflops"] = cal["tflops"]
            result["config"]["box_calibration_tflops"] = cal["tflops"]
            result["config"]["value_per_calibration_tflops"] = cal["value_per_calibration_tflops"]
        if args.one_gpu_value:
            result["scaling_efficiency"] = images_per_s / (world * args.one_gpu_value)
        if fp8_run:
            b16 = prof["gemm_bf16"]
            result["bf16_gemm_in_same_run"] = {"tflops": b16["work"] / (b16["ms"] * 1e-3) / 1e12 if b16["ms"] > 0 else 0.0,
                                               "kernel_ms_total": b16["ms"], "launches": b16["launches"]}
        if world == 1 and not args.no_cpu_baseline:
            try:
                result["cpu_baseline"] = cpu_baseline_lmm(dims, device, 1234, T, pix, out, args.cpu_images, engine, args.cpu_budget_s)
                result["cpu_baseline_label_cosine"] = cpu_baseline_scorer(4096, L)
            except Exception as e:  # the baseline must never sink the measurement
                result["cpu_baseline"] = {"value": None, "unit": "images/s", "cores": os.cpu_count(), "kind": "reference",
                                          "sample": f"failed: {type(e).__name__}: {e}"}
            # full-size parity gate (bf16 only; the fp8 decoder has its own stated bound in tests/): the process exits non-zero
            # AFTER printing the line when the teacher-forced logits are off or a token flipped on a decisive margin
            lv, tv = result["cpu_baseline"].get("logits_vs_hip"), result["cpu_baseline"].get("tokens_vs_hip")
            if not fp8_run and lv and lv.get("worst_rel_err") is not None:
                if lv["worst_rel_err"] > FULLSIZE_LOGIT_BOUND:
                    parity_failure = f"full-size logit parity failed: worst step {lv['worst_rel_err']:.4f} of max|logit| > {FULLSIZE_LOGIT_BOUND}"
                bad = [f for f in tv["flips"] if not f["explained_by_error"]]
                if bad and not parity_failure:
                    parity_failure = f"full-size token parity failed: {len(bad)} teacher-forced flip(s) on a decisive margin: {bad[:2]}"
                result["cpu_baseline"]["parity_gate"] = {"logit_bound": FULLSIZE_LOGIT_BOUND, "passed": parity_failure is None}
        lap("cpu_baseline")
        # ---- BASELINE.json configs[3] / configs[4]'s LMM side on this GPU (never `value`): the headline model's weights, inputs and
        # caches go back to the driver first (73 GB of fp8 + bf16 weights, then 69.5 GB of bf16 weights)
        big = {"config5_qwen2vl_72b_fp8": None, "config4_llava_next_34b": None}

        # The line is owed from here on, whatever the big-model legs do (a hang, a driver time-out or a fault in a leg
        # that runs AFTER the measurement must not lose the measurement): `emit` assembles and prints it exactly once - from the normal
        # path below, from a SIGTERM / SIGINT handler, or from a watchdog thread once the process is `--total-budget-s` old (then the
        # process exits 0 with the unfinished leg marked as skipped).
        import signal
        import threading

        emit_lock, emitted = threading.RLock(), [False]   # re-entrant: the signal handler runs on the main thread, which may hold it

        def emit(why: str | None = None) -> None:
            with emit_lock:
                if emitted[0]:
                    return
                emitted[0] = True
                if why:
                    for name in big:
                        if big[name] is None:
                            big[name] = {"skipped": why}
                finish_line()

        def finish_line() -> None:
            # ---- the line ends with what round 5 added (the driver's record keeps the END of the line)
            result["images_per_s_from_host_uint8"] = result.pop("images_per_s_from_host_uint8")
            result["images_per_s_from_pil"] = result.pop("images_per_s_from_pil")
            result.update(big)
            result["real_image_sizes"] = ragged
            result["max_pixels_images"] = cap_leg
            if calib is not None and calib.runs:
                result["box_calibration"] = cal
            cfgd = result["config"]
            for k, leg in (("config3_mix", ragged), ("max_pixels", cap_leg)):
                if leg and "images_per_s" in leg:
                    cfgd[k + "_images_per_s"], cfgd[k + "_mfma_frac_end_to_end"] = leg["images_per_s"], leg.get("mfma_frac_end_to_end")
                    if leg.get("roofline_attention_vision"):
                        cfgd[k + "_vision_attn_tflops"] = leg["roofline_attention_vision"]["achieved"]
                        cfgd[k + "_vision_attn_share"] = leg["roofline_attention_vision"]["share_of_leg_time"]
            cfgd["qwen2vl_72b_fp8_images_per_s"] = (big["config5_qwen2vl_72b_fp8"] or {}).get("images_per_s")
            cfgd["llava_next_34b_images_per_s"] = (big["config4_llava_next_34b"] or {}).get("value")
            leg_seconds["total_since_process_start"] = round(time.perf_counter() - T_PROCESS_START, 1)
            result["leg_seconds"] = leg_seconds
            print(json.dumps(result), flush=True)

        def on_signal(signum, _frame):
            emit(f"signal {signum} during the big-model legs: the line was printed from the handler")
            os._exit(0)

        def watchdog():
            while not emitted[0]:
                if time.perf_counter() - T_PROCESS_START > args.total_budget_s:
                    emit(f"--total-budget-s {args.total_budget_s:.0f} s reached inside this leg: the line was printed by the watchdog")
                    os._exit(0)
                time.sleep(1.0)

        old_handlers = {}
        if world == 1 and not args.no_big_legs and not args.no_extra_legs:
            for sg in (signal.SIGTERM, signal.SIGINT):
                try:
                    old_handlers[sg] = signal.signal(sg, on_signal)
                except (ValueError, OSError):
                    pass
            threading.Thread(target=watchdog, daemon=True, name="owc-bench-line-watchdog").start()
        if world == 1 and not args.no_big_legs and not args.no_extra_legs:
            engine = weights = pix = scorer = cls_z = emb0 = None
            import gc

            for name, fn in (("config5_qwen2vl_72b_fp8", lambda: qwen72b_fp8_leg(device, T, profile)),
                             ("config4_llava_next_34b", lambda: llava_next_34b_leg())):
                if time.perf_counter() - T_PROCESS_START > args.big_leg_budget_s:
                    big[name] = {"skipped": f"--big-leg-budget-s {args.big_leg_budget_s:.0f} s reached before this leg"}
                    continue
                gc.collect()
                torch.cuda.empty_cache()
                try:
                    got = fn()
                except BaseException as e:   # an extra leg must never sink the measurement (KeyboardInterrupt / SystemExit included)
                    got = {"skipped": f"{type(e).__name__}: {str(e)[:200]}"}
                with emit_lock:
                    if not emitted[0]:
                        big[name] = got
                lap(name)
            gc.collect()
            torch.cuda.empty_cache()
        emit()
        for sg, h in old_handlers.items():
            signal.signal(sg, h)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for leg in (ragged, cap_leg):
        if leg is not None and not leg["deterministic_and_batch_invariant"]:
            raise SystemExit("ragged leg: determinism / batch invariance failed (the JSON line above carries the measurement)")
    if not invariant:
        raise SystemExit("batch invariance check failed (the JSON line above carries the measurement)")
    if parity_failure:
        raise SystemExit(parity_failure + " (the JSON line above carries the measurement)")


def pil_leg(engine, dims, host_u8, B: int, T: int, device, sync, lm=None) -> dict:
    """PIL images -> `Qwen2VL.generate_until` -> strings on the engine of the main leg (same weights): 2 x B images; the wrapper
    prepares them in units of B / 16 and assembles engine passes adaptively (the GPU starts after one unit's preparation, later
    passes grow to B), so the host preparation overlaps GPU work; the first unit's preparation has nothing to hide behind and is
    inside the timed region, as are the one-time costs of a call (allocations for each new pass size)."""
    from PIL import Image

    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL
    from lmms_owc_amd.tasks import ClassificationTask

    class BenchTokenizer(ByteTokenizer):
        """The main leg's 286-token prompt (14 text ids + image placeholders + 16 text ids): a byte tokenizer would spell the
        question out one token per character (~360 tokens) and the two legs would not do the same work per image."""

        def chat_ids(self, question: str, n_image_tokens: list[int]) -> list[int]:
            ids = prompt_ids(self.image_pad)
            assert n_image_tokens == [S_IMG]
            return [int(t) for t in ids]

    n, bs = 2 * B, B
    arr = host_u8[:B].permute(0, 2, 3, 1).contiguous().numpy()            # HWC uint8 (uniform noise: the slowest JPEG case)
    docs = [{"visual": Image.fromarray(arr[i % len(arr)], "RGB"), "target": f"class_{i % 10}"} for i in range(n)]
    task = ClassificationTask("bench", docs, generation_kwargs={"max_new_tokens": T, "do_sample": False})
    if lm is None:
        lm = Qwen2VL.from_engine(engine, BenchTokenizer(), batch_size=bs, eos_token_id=-1)
    lm.task_dict["bench"] = task.dataset
    task.build_all_requests(limit=None, rank=0, world_size=1)
    warm = task.instances[: min(8, n)]                                  # warm the worker pool / pinned allocator on their own
    lm.generate_until(warm)                                             # requests (generate_until pops `until` from a request's
    task.build_all_requests(limit=None, rank=0, world_size=1)           # gen_kwargs, which would split the timed call's grouping)
    sync()
    t0 = time.perf_counter()
    answers = lm.generate_until(task.instances)
    sync()
    dtp = time.perf_counter() - t0
    assert len(answers) == n and all(isinstance(a, str) for a in answers)
    first = lm.last_timing.get("first_chunk_prep_s", 0.0)
    pinned_to = lm._cpu_affinity
    lm.release_host_resources()     # the CPU baseline below runs in this process: its threads must see every core again
    return {"seconds": dtp, "images": n, "batch_size": bs, "prep_threads": lm._prep_threads, "host_cores": os.cpu_count(), "host": host_cpu_record(),
            "cpu_affinity": None if not pinned_to else f"{len(pinned_to)} CPUs of the GPU's NUMA node ({pinned_to[0]}..{pinned_to[-1]})",
            "chunks": lm.last_timing.get("chunks"),
            "first_chunk_prep_s": first,   # exposed once per generate_until call (a task), whatever its length
            "images_per_s_after_first_prep": n / max(dtp - first, 1e-9),   # what a long task converges to (per rank)
            "what": "PIL 448x448 (uniform-noise pixels) -> JPEG round trip + smart_resize/bicubic + tokenise on the host pool -> pinned "
                    "H2D -> patchify -> vision tower -> prefill -> 16 greedy tokens -> detokenised strings; chunk k+1 is prepared while "
                    "chunk k runs on the GPU"}


def load_traffic(args, engine, B: int) -> dict | None:
    """roofline.traffic = HBM bytes per launch of the step's largest launch class, READ FROM the committed rocprofv3 PMC summary
    (profiles/*_pmc_gemm_traffic*.json: separate FETCH_SIZE / WRITE_SIZE passes, gfx950 x2 correction on FETCH_SIZE) - never a
    literal.  Only when this run contains that launch class: 7B, one full 65536-row prefill group, no tuning knob, bf16."""
    if args.model != "7b" or B < 256 or engine.prefill_chunk_tokens != 65536 or args.nominal_forward:
        return None
    if any(k.startswith("OWC_GEMM_") for k in os.environ) or args.tune:
        return None
    # the summary is selected by FIELDS of the JSON (the kernel the default dispatch launches for this shape, highest `seq` =
    # the latest measurement of it), never by file-name order
    best = None
    for f in (ROOT / "profiles").glob("r*_pmc_gemm_traffic*.json"):
        try:
            d = json.loads(f.read_text())
            if d.get("kernel") != TRAFFIC_KERNEL or "bytes_per_launch" not in d:
                continue
            if best is None or d.get("seq", -1) > best[1].get("seq", -1):
                best = (f, d)
        except ValueError:
            continue
    if best is None:
        return None
    f, d = best
    return {"bytes_per_launch": d["bytes_per_launch"], "algorithmic_bytes_per_launch": d.get("algorithmic_bytes_per_launch"),
            "kernel": d.get("kernel"), "shape": d.get("shape"), "seq": d.get("seq"), "file": str(f.relative_to(ROOT)),
            "raw_counters": d.get("raw_csv")}


PEAK_HBM_GBS = 8000.0   # HBM3E peak, MI355X_MICROARCH.md (about 6.3 TB/s is what a pure stream achieves)


def attention_rooflines(prof: dict, d, B: int, T: int, steps: int, dt: float) -> dict:
    """Roofline objects of the THREE attention launch classes of the timed region (HIP events inside the library; the launcher
    cannot see the device-side lengths, so the algorithmic work is priced here from the workload's shapes):
    vision (non-causal, MFMA-bound), causal prefill (MFMA-bound), decode (one query row per q head over the sequence's whole
    KV cache: HBM-bound, priced in bytes = the K and V rows it streams)."""
    S = S_TEXT_BEFORE + S_IMG + S_TEXT_AFTER
    vis = steps * B * d.v_depth * 4.0 * 1024 * 1024 * d.v_embed                      # QK^T + PV, non-causal, per image per layer
    if getattr(d, "v_variant", 0) == 1:   # Qwen2.5-VL: 64-key windows except in the full-attention blocks
        vis = steps * B * 4.0 * 1024 * d.v_embed * (len(d.v_fullatt) * 1024 + (d.v_depth - len(d.v_fullatt)) * 64)
    pre = steps * B * d.n_layers * 2.0 * S * S * d.n_q_heads * d.head_dim            # causal: half of 4 S^2 H hd
    kv_row = 2.0 * d.n_kv_heads * d.head_dim * 2                                     # K + V bytes of one token in one layer
    dec_bytes = steps * B * d.n_layers * sum(kv_row * (S + i + 1) for i in range(T - 1))   # step i reads the S + i + 1 cached rows
    out = {}
    for name, kind, flops, kernel in (("vision", "attn_vision", vis, "attn_fwd_kernel<80,false>"),
                                      ("prefill", "attn_prefill", pre, "attn_fwd_kernel<128,true> (causal GQA prefill, S = 286)")):
        p = prof[kind]
        tf = flops / (p["ms"] * 1e-3) / 1e12 if p["ms"] > 0 else 0.0
        out[name] = {"bound": "mfma", "kernel": kernel, "achieved": tf, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                     "frac": tf / PEAK_BF16_TFLOPS, "traffic": None, "launches": p["launches"], "kernel_ms_total": p["ms"],
                     "share_of_step_time": p["ms"] * 1e-3 / time_or(dt)}
    p = prof["attn_decode"]
    gbs = dec_bytes / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else 0.0
    out["decode"] = {"bound": "hbm", "kernel": "attn_decode_fused_kernel<1|2> (rope + KV-cache write + attention of a decode step; a block = one (sequence, kv head), its 4 waves split the keys)",
                     "achieved": gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": gbs / PEAK_HBM_GBS, "traffic": None,
                     "algorithmic_bytes": "K + V rows of the sequence's cache, once per kv head per step: 2 * n_kv_heads * 128 * 2 B per token per layer",
                     "launches": p["launches"], "kernel_ms_total": p["ms"], "share_of_step_time": p["ms"] * 1e-3 / time_or(dt)}
    return out


def decode_regime_leg(engine, dims, batches=(1, 32, 128), s_prompt: int = 286, t_short: int = 2, t_long: int = 18) -> dict:
    """The HBM-bound regime of the greedy decode loop (SURVEY.md section 8d "Bound": a decode step streams every decoder weight once
    whatever the batch, so below the ridge B ~ 312 it is a weight stream): ms per token-step of the bench's own engine at the
    reference's batch size (1) and at mid batches, from the difference of a `t_long`- and a `t_short`-token generate on text-only
    prompts of S = 286 tokens.  Algorithmic bytes per step = the decoder layers' weights + lm_head (each read once) + the K/V
    rows of every sequence (2 KB per token per layer for 7B) - NOT the embedding table (B rows of it) and not the vision tower."""
    d = dims
    wbytes = 2.0 * (d.n_layers * (d.d_model * (d.n_q_heads + 2 * d.n_kv_heads) * d.head_dim + d.n_q_heads * d.head_dim * d.d_model
                                  + 3 * d.d_model * d.d_ff) + d.vocab * d.d_model)
    if getattr(d, "decoder_dtype", "bf16") == "fp8":   # fp8 decoder projections: 1 byte per layer weight, lm_head stays bf16
        wbytes = wbytes - (wbytes - 2.0 * d.vocab * d.d_model) / 2
    r = np.random.default_rng(0)
    rows = []
    for B in batches:
        prompts = [r.integers(1000, 30000, s_prompt).astype(np.int32) for _ in range(B)]
        none = [[] for _ in prompts]
        ts = {}
        for T in (t_short, t_long):
            engine.generate(prompts, None, none, T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                engine.generate(prompts, None, none, T)
            torch.cuda.synchronize()
            ts[T] = (time.perf_counter() - t0) / 3
        per_step = (ts[t_long] - ts[t_short]) / (t_long - t_short)
        kv = B * d.n_layers * 2.0 * d.n_kv_heads * d.head_dim * 2 * (s_prompt + (t_short + t_long) / 2.0)   # mean cached rows over the steps
        rows.append({"batch": B, "ms_per_step": per_step * 1e3, "bytes_per_step": wbytes + kv, "achieved": (wbytes + kv) / per_step / 1e9,
                     "frac": (wbytes + kv) / per_step / 1e9 / PEAK_HBM_GBS, "tokens_per_s": B / per_step})
    head = rows[0]
    return {"bound": "hbm", "kernel": "one greedy decode step (owc_llm_decode_step: 28 layers of weight-streaming GEMMs + KV-cache attention + lm_head)",
            "achieved": head["achieved"], "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": head["frac"], "traffic": None,
            "batch": head["batch"], "ms_per_step": head["ms_per_step"], "by_batch": rows,
            "weight_bytes_per_step": wbytes,
            "regime": "the REFERENCE's batch size (1) and mid batches: the HBM-bound side of the decode loop.  The product does not run "
                      "here - `engine_batch=auto` decodes a whole pass together (2048 rows, MFMA-bound; its measured step time is this record's `by_batch` entry of the largest batch when `--decode-batches` includes it, and profiles/*decode* otherwise) and reaches "
                      "these row counts only in the tail of a pass - so `achieved` says how good the weight stream is, not how the "
                      "headline is made",
            "method": f"(t({t_long} new tokens) - t({t_short} new tokens)) / {t_long - t_short} on text-only prompts of {s_prompt} tokens, 3 repeats each, "
                      "host clock around synchronised generates (every generate is >= 50 ms); `achieved` = the reference's own batch size (1)"}


def scorer_rooflines(sprof: dict, n_lab: int, n_cls: int, k: int, steps: int, sdt: float) -> dict:
    """Metric 2: the sentence encoder's linears (fp32 operands as 3 bf16 pieces, 6 piece-products per fp32 product on the bf16
    MFMA: effective peak 2.5 PF / 6) and the cosine top-k kernel (HBM bytes 4 D (N + C) + 8 k N)."""
    D = 384
    gm, ck = sprof["scorer_gemm"], sprof["cosine_topk"]
    eff_peak = PEAK_BF16_TFLOPS / 6.0
    g_tf = gm["work"] / (gm["ms"] * 1e-3) / 1e12 if gm["ms"] > 0 else 0.0
    ck_bytes = steps * (4.0 * D * (n_lab + n_cls) + 8.0 * k * n_lab)
    ck_gbs = ck_bytes / (ck["ms"] * 1e-3) / 1e9 if ck["ms"] > 0 else 0.0
    ck_tf = ck["work"] / (ck["ms"] * 1e-3) / 1e12 if ck["ms"] > 0 else 0.0
    return {
        "embed_gemm": {"bound": "mfma", "kernel": "gemm_f32x3_nt_kernel (BERT linears, rows = real tokens only)", "achieved": g_tf,
                       "peak": eff_peak, "unit": "TFLOP/s", "frac": g_tf / eff_peak, "traffic": None, "launches": gm["launches"],
                       "kernel_ms_total": gm["ms"], "share_of_leg_time": gm["ms"] * 1e-3 / time_or(sdt),
                       "peak_note": "fp32-equivalent FLOPs: each product costs six bf16 MFMA piece-products (2.5 PF / 6)"},
        "cosine_topk": {"bound": "mfma", "kernel": "cosine_topk_kernel", "achieved": ck_tf, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": ck_tf / 157.3, "traffic": None, "launches": ck["launches"], "kernel_ms_total": ck["ms"],
                        "share_of_leg_time": ck["ms"] * 1e-3 / time_or(sdt), "hbm_gbs_on_algorithmic_bytes": ck_gbs,
                        "note": "2 N C D on the f32-input MFMA (157 TF dense f32 matrix peak) + a register-resident top-k insertion per "
                                "lane; algorithmic bytes 4 D (N + C) + 8 k N move at the quoted GB/s (HBM is not the bound at C = 397); "
                                "the kernel is < 1 % of the scorer leg"},
    }


def time_or(x: float) -> float:
    return x if x > 0 else 1.0


if __name__ == "__main__":
    main()
