#!/usr/bin/env python3
"""Host-side soak of an N-rank node WITHOUT N GPUs (SURVEY.md section 8e "scaling risk"; reference `accelerate launch
--num_processes 8`, scripts/schedule_batch.sh:109-112, src/utils/_core_utils.py:53-69 strided shard).

N processes (LOCAL_WORLD_SIZE = N, no GPU touched) each run the REAL `Qwen2VL.generate_until` host pipeline of
lmms_owc_amd/models/_qwen2_vl.py - Collator grouping, chunking, the preparation thread + PIL worker pool (JPEG round trip,
two-stage smart_resize / bicubic, tokenisation, staging copies), look-ahead, EOS cut, detokenisation - with exactly one method
replaced: `_launch_chunk` (H2D + patchify + vision tower + prefill + decode) becomes a stand-in that occupies the launching
thread for `--launch-ms-per-image` (the measured host cost of enqueueing a chunk) and "finishes" the chunk `images /
--gpu-rate` seconds after the previous one (the measured engine rate), like a stream would.  Reported per rank: prepared
images/s, seconds the launching thread stood waiting for an unprepared chunk, and the whole node's aggregate against N x rate.

  python tools/soak_host_ranks.py --ranks 8 --images 6144 --gpu-rate 240 [--threads 8] [--size 448x448 | --sizes food101]
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def child(args) -> None:
    import numpy as np
    import torch
    from PIL import Image

    from bench import DATASET_SIZES
    from lmms_owc_amd.models._base import CacheHook
    from lmms_owc_amd.models._qwen2_vl import ByteTokenizer, Qwen2VL
    from lmms_owc_amd.tasks import ClassificationTask

    rank = int(os.environ["RANK"])
    torch.set_num_threads(1)

    class Dims:
        image_token_id = ByteTokenizer.image_pad

    class SoakQwen2VL(Qwen2VL):
        """The real plug-in minus the GPU: `_launch_chunk` is a clock."""

        gpu_free_at = 0.0

        def _pinned_take(self, shape):   # no GPU in this process: the staging copies go to ordinary memory (same memcpy work)
            return torch.empty(int(np.prod(shape)), dtype=torch.uint8).view(shape)

        _no_carry = True

        def _launch_chunk(self, prep, eos_token_id, pad, carry=None):
            n = prep["n"]
            t_end = time.perf_counter() + n * args.launch_ms_per_image * 1e-3
            while time.perf_counter() < t_end:      # the launching thread is busy (and holds the GIL part of the time)
                sum(range(2000))
            start = max(time.perf_counter(), type(self).gpu_free_at)
            type(self).gpu_free_at = start + n / args.gpu_rate
            done_at = type(self).gpu_free_at

            class Ev:
                def query(self_inner):
                    return time.perf_counter() >= done_at

                def synchronize(self_inner):
                    dt = done_at - time.perf_counter()
                    if dt > 0:
                        time.sleep(dt)

            r = np.random.default_rng(n)
            toks = r.integers(3, 250, (n, prep["max_new"])).astype(np.int32)
            toks[np.arange(n), r.integers(2, min(12, prep["max_new"]), n)] = eos_token_id   # answers of 2-11 tokens
            return torch.from_numpy(toks), Ev()

    lm = SoakQwen2VL.__new__(SoakQwen2VL)
    lm._engine_batch_arg = args.chunk
    lm._model_name_or_path, lm._decoder_dtype = "soak", "bf16"
    lm._max_pixels, lm._min_pixels = 1024 * 28 * 28, 4 * 28 * 28
    lm._device, lm._rank, lm._world_size = torch.device("cpu"), rank, args.ranks
    lm.batch_size_per_gpu = 1
    lm.cache_hook, lm.chat_template, lm.apply_chat_template, lm.task_dict = CacheHook(None), None, False, {}
    lm._tokenizer = lm._processor = ByteTokenizer()
    lm._dims, lm._model = Dims(), None
    if args.threads:
        os.environ["OWC_PREP_THREADS"] = str(args.threads)
    pinned = None
    if args.pin:   # what `pin_to_gpu_numa_node` does on a GPU box, with the GPU -> node map EMULATED: the ranks split evenly over the nodes
        from lmms_owc_amd.models._base import numa_share

        nodes = sorted(Path("/sys/devices/system/node").glob("node[0-9]*"), key=lambda p: int(p.name[4:]))
        lists = []
        for nd in nodes:
            cpus = []
            for part in filter(None, (nd / "cpulist").read_text().strip().split(",")):
                lo, _, hi = part.partition("-")
                cpus += list(range(int(lo), int(hi or lo) + 1))
            if cpus:
                lists.append(cpus)
        if len(lists) > 1:
            per_node = -(-args.ranks // len(lists))
            node, k = rank // per_node, rank % per_node
            here = min(per_node, args.ranks - node * per_node)
            pinned = numa_share(lists[node], k, here, set(os.sched_getaffinity(0)))
            if pinned:
                os.sched_setaffinity(0, pinned)
    if args.pillow_blocks:
        Image.core.set_blocks_max(args.pillow_blocks)
    if args.switch_interval_ms:
        sys.setswitchinterval(args.switch_interval_ms * 1e-3)
    lm._start_workers()

    r = np.random.default_rng(100 + rank)
    if args.sizes:
        sizes = DATASET_SIZES[args.sizes](r, 64)
    else:
        h, w = (int(x) for x in args.size.split("x"))
        sizes = [(h, w)]
    base = [Image.fromarray(r.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for h, w in sizes]   # noise: the slowest JPEG case
    n = args.images
    docs = [{"visual": base[i % len(base)], "target": f"class_{i % 10}"} for i in range(n)]
    task = ClassificationTask("soak", docs, generation_kwargs={"max_new_tokens": 64, "do_sample": False})
    lm.task_dict["soak"] = task.dataset
    task.build_all_requests(limit=None, rank=0, world_size=1)
    lm.generate_until(task.instances[:64])                       # warm the pools
    task.build_all_requests(limit=None, rank=0, world_size=1)
    SoakQwen2VL.gpu_free_at = 0.0
    # all ranks start together (file barrier: no process group needed for a host soak)
    Path(args.sync_dir, f"ready{rank}").touch()
    while len(list(Path(args.sync_dir).glob("ready*"))) < args.ranks:
        time.sleep(0.01)
    t0 = time.perf_counter()
    answers = lm.generate_until(task.instances)
    dt = time.perf_counter() - t0
    assert len(answers) == n and all(isinstance(a, str) for a in answers)
    lt = lm.last_timing
    print(json.dumps({"rank": rank, "images": n, "seconds": dt, "images_per_s": n / dt, "prep_threads": lm._prep_threads,
                      "chunks": lt.get("chunks"), "pass_sizes": lt.get("pass_sizes"), "first_chunk_prep_s": lt.get("first_chunk_prep_s"),
                      "prep_wait_s": lt.get("prep_wait_s", 0.0), "gpu_seconds_emulated": n / args.gpu_rate,
                      "pinned_cpus": None if not pinned else f"{len(pinned)}: {pinned[0]}..{pinned[-1]}"}), flush=True)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, default=8)
    ap.add_argument("--images", type=int, default=6144, help="images per rank")
    ap.add_argument("--chunk", type=int, default=2048, help="requests per engine pass (engine_batch)")
    ap.add_argument("--gpu-rate", type=float, default=240.0, help="images/s one GPU sustains (the measured engine rate)")
    ap.add_argument("--launch-ms-per-image", type=float, default=0.15,
                    help="host time the launching thread spends enqueueing a chunk, per image (measured: tools/profile_host_batch1.py)")
    ap.add_argument("--threads", type=int, default=0, help="OWC_PREP_THREADS per rank (0: the plug-in's default for LOCAL_WORLD_SIZE)")
    ap.add_argument("--size", default="448x448")
    ap.add_argument("--sizes", default=None, choices=["food101", "dtd", "flowers102"])
    ap.add_argument("--pin", action="store_true", help="pin every rank to its share of one NUMA node (emulated GPU -> node map: ranks split evenly)")
    ap.add_argument("--pillow-blocks", type=int, default=0, help="Pillow's arena cache (Image.core.set_blocks_max): freed image blocks are reused instead of unmapped")
    ap.add_argument("--switch-interval-ms", type=float, default=0.0, help="sys.setswitchinterval: how long a thread keeps the GIL while others wait (default 5 ms)")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--sync-dir", default=None)
    args = ap.parse_args()
    if args.child:
        return child(args)
    import tempfile

    with tempfile.TemporaryDirectory() as td:
        procs = []
        for rk in range(args.ranks):
            env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(args.ranks), LOCAL_WORLD_SIZE=str(args.ranks),
                       HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
            procs.append(subprocess.Popen([sys.executable, __file__, *sys.argv[1:], "--child", "--sync-dir", td], env=env,
                                          stdout=subprocess.PIPE, text=True))
        rows = []
        for p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                raise SystemExit(f"a soak rank failed with {p.returncode}")
            rows.append(json.loads([ln for ln in out.splitlines() if ln.startswith("{")][-1]))
    rows.sort(key=lambda x: x["rank"])
    wall = max(x["seconds"] for x in rows)
    total = sum(x["images"] for x in rows)
    from lmms_owc_amd.models._base import usable_cpus

    print(json.dumps({"ranks": args.ranks, "host_cores": os.cpu_count(), "usable_cpus": usable_cpus()[0], "cgroup_cpu_quota": usable_cpus()[1],
                      "prep_threads_per_rank": rows[0]["prep_threads"],
                      "images_per_rank": args.images, "emulated_gpu_rate_per_rank": args.gpu_rate,
                      "aggregate_images_per_s": total / wall, "target_images_per_s": args.ranks * args.gpu_rate,
                      "fraction_of_target": total / wall / (args.ranks * args.gpu_rate),
                      "per_rank_images_per_s": [round(x["images_per_s"], 1) for x in rows],
                      "per_rank_prep_wait_s": [round(x["prep_wait_s"], 3) for x in rows],
                      "per_rank_first_chunk_prep_s": [round(x["first_chunk_prep_s"], 3) for x in rows],
                      "rank0_pass_sizes": rows[0].get("pass_sizes"), "pinned_cpus_per_rank": [x.get("pinned_cpus") for x in rows],
                      "pillow_blocks_max": args.pillow_blocks, "switch_interval_ms": args.switch_interval_ms or 5.0, "image_sizes": args.sizes or args.size}))


if __name__ == "__main__":
    main()
