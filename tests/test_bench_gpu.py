"""bench.py's N > 1 code path on real hardware.

An 8-GPU node is the driver's to launch; a gpurun box has ONE GPU.  The test hook OWC_BENCH_SHARE_GPU=1 (bench.py) puts both ranks
of a --gpus 2 run on cuda:0 with gloo collectives (RCCL refuses two ranks on one device), so that everything after the rendezvous
- per-rank weights and inputs, the barrier-bracketed timed region, the gather of per-rank times, max-over-ranks, the whole-job
value, the scorer / PCIe legs' reductions and rank 0's JSON line - runs on the HIP engine exactly as it will with 8 ranks.
"""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_two_ranks_share_one_gpu(gpu):
    env = dict(os.environ, OWC_BENCH_SHARE_GPU="1", MASTER_ADDR="127.0.0.1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--model", "2b", "--batch", "64", "--steps", "2", "--warmup", "1",
           "--scorer-labels", "4096", "--no-cpu-baseline", "--no-pil-leg", "--cap-images", "12", "--ragged-images", "256"]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout       # rank 0 prints ONE line, rank 1 none
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and out["warmup"] == 1 and out["scaling"] == "weak"
    assert out["config"]["images_per_gpu_per_step"] == 64 and out["config"]["parallelism"] == "dp2" and out["rccl_world_size"] == 2
    # whole-job value = both ranks' images over the max-over-ranks time
    assert abs(out["value"] - 128 / (out["ms_per_step"] / 1e3)) <= 1e-3 * out["value"]
    assert len(out["per_rank_images_per_s"]) == 2 and all(v > 0 for v in out["per_rank_images_per_s"])
    assert out["value"] <= sum(out["per_rank_images_per_s"]) * (1 + 1e-6)
    assert out["batch_invariance_check"].startswith("ok")
    assert out["roofline"]["launches"] > 0 and 0 < out["roofline"]["frac"] < 1     # rank 0's launches only
    assert out["label_cosine_per_sec"] > 0 and out["images_per_s_from_host_uint8"] > 0
    # round 3: three attention objects (decode priced in bytes against HBM), the HBM-regime decode leg on rank 0
    att = out["roofline_attention"]
    assert set(att) == {"vision", "prefill", "decode"} and att["decode"]["bound"] == "hbm" and att["decode"]["unit"] == "GB/s"
    assert all(a["launches"] > 0 and 0 < a["frac"] < 1 for a in att.values())
    dec = out["roofline_decode"]
    assert dec["bound"] == "hbm" and dec["batch"] == 1 and [r["batch"] for r in dec["by_batch"]] == [1, 32, 128]
    assert all(0 < r["frac"] < 1 and r["ms_per_step"] > 0 for r in dec["by_batch"])
    # round 4: ragged answer lengths with / without row compaction (same tokens), the other configs' legs in the default line
    eos = out["eos_terminated"]
    assert [r["max_new_tokens"] for r in eos["by_cap"]] == [64, 256]
    c64 = eos["by_cap"][0]
    assert c64["tokens_identical_with_and_without_compaction"] is True and c64["pad_behind_stop_column"] is True
    assert c64["compacted"]["row_steps"] < 0.5 * c64["all_rows_every_step"]["row_steps"] and c64["compacted"]["images_per_s"] > 0
    assert eos["by_cap"][1]["compacted"]["decode_steps_run"] == 255
    rag = out["real_image_sizes"]
    assert rag["dataset_size_model"] == "config3" and rag["deterministic_and_batch_invariant"] and rag["roofline"]["launches"] > 0
    # round 5: config #3's size mixture; every image on the max_pixels cap (4096 / 3996 patches) with the vision attention's own
    # roofline object and share; executed-FLOP utilisation for both; the from-host rates and the leg summaries inside `config`
    assert 0 < rag["mfma_frac_end_to_end"] <= rag["mfma_frac_end_to_end_nominal"] < 1
    cap = out["max_pixels_images"]
    assert cap["images"] == 12 and cap["image_tokens_per_image"] == {"min": 999, "mean": 1011.5, "max": 1024}
    assert cap["deterministic_and_batch_invariant"] and cap["roofline_attention_vision"]["patches_per_image"]["max"] == 4096
    assert 0 < cap["roofline_attention_vision"]["frac"] < 1 and 0 < cap["roofline_attention_vision"]["share_of_leg_time"] < 1
    assert rag["roofline_attention_vision"]["launches"] > 0
    cfg = out["config"]
    assert cfg["timed_region_starts_from"].startswith("pixel_values") and cfg["images_per_s_from_host_uint8"] > 0
    assert cfg["max_pixels_images_per_s"] == cap["images_per_s"] and cfg["config3_mix_images_per_s"] == rag["images_per_s"]
    assert list(out)[-1] == "leg_seconds" and "max_pixels_images" in list(out)[-8:]      # the driver's record keeps the END of the line
    assert out["config5_qwen2vl_72b_fp8"] is None and out["config4_llava_next_34b"] is None   # (1-GPU legs: not in a 2-rank run)
    cos = out["label_cosine_10k_classes"]
    assert cos["classes"] == 10000 and cos["top1_matches_dense_matmul_on_256_rows"] and 0 < cos["roofline"]["frac"] < 1
    assert out["config2_qwen2vl_2b"] is None     # (this run IS the 2B model)
    # where the run's wall time went (rank 0's clock); the optional legs are skipped beyond --leg-budget-s - not here
    ls = out["leg_seconds"]
    assert {"setup_weights_inputs", "warmup_and_timed_steps", "eos_terminated", "real_image_sizes", "max_pixels_images", "roofline_decode"} <= set(ls)
    assert abs(sum(v for k, v in ls.items() if k != "total_since_process_start") - ls["total_since_process_start"]) < 2.0
