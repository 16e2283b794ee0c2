"""EXPERIMENT (round 6): do two engine passes IN FLIGHT on two HIP streams beat one after the other?

A pass of the 7B headline workload = vision tower + prefill (large GEMMs that fill the chip) + 16 decode steps at 2048 rows, whose
GEMMs leave CUs idle (qkv: 144 tiles of 256x256 on 256 CUs; o / down: one round of 224 tiles with a tail) and whose attention is a
KV-cache stream.  With two engines (same weights, own workspaces and KV caches) each on a stream of its own, the idle CUs of one
pass's decode steps could take blocks of the other pass's large GEMMs.  Measured here, same process:
  serial:    engine A runs 2 x N passes on one stream
  two-stream: engines A and B run N passes each, enqueued alternately on two streams
usage: python tools/exp_two_streams.py [passes=4] [batch=2048] [new_tokens=16]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402  (prompt_ids)
from lmms_owc_amd import ops as owc_ops  # noqa: E402
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402
from lmms_owc_amd.models import imageproc  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    dev = torch.device("cuda:0")
    dims = DIMS["qwen2-vl-7b"]
    weights = Qwen2VLWeights.random(dims, dev, seed=1234)
    engA, engB = Qwen2VLEngine(weights), Qwen2VLEngine(weights)
    gen = torch.Generator(device=dev).manual_seed(1234)
    pix = torch.empty((B * 1024, 1176), device=dev, dtype=torch.bfloat16)
    for i0 in range(0, B, 256):
        n_i = min(256, B - i0)
        u8 = torch.randint(0, 256, (n_i, 3, 448, 448), generator=gen, device=dev, dtype=torch.uint8)
        owc_ops.patchify_u8(u8, imageproc.OPENAI_CLIP_MEAN, imageproc.OPENAI_CLIP_STD, out=pix[i0 * 1024:(i0 + n_i) * 1024])
    ids = bench.prompt_ids(dims.image_token_id)
    prompts, grids, flat = [ids] * B, [[(1, 32, 32)]] * B, [(1, 32, 32)] * B

    def one_pass(eng):
        emb = eng.encode_images(pix, flat)
        return eng.generate(prompts, emb, grids, T, eos_token_id=-1, pad_token_id=0)

    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    # warm both engines (allocations, kernel attributes)
    for eng, st in ((engA, sA), (engB, sB)):
        with torch.cuda.stream(st):
            ref = one_pass(eng)
    torch.cuda.synchronize()
    ref = ref.cpu()

    for rnd in range(2):
        t0 = time.perf_counter()
        with torch.cuda.stream(sA):
            for _ in range(2 * N):
                out = one_pass(engA)
        torch.cuda.synchronize()
        t_serial = time.perf_counter() - t0
        assert torch.equal(out.cpu(), ref)
        t0 = time.perf_counter()
        outs = []
        for _ in range(N):
            with torch.cuda.stream(sA):
                outs.append(one_pass(engA))
            with torch.cuda.stream(sB):
                outs.append(one_pass(engB))
        torch.cuda.synchronize()
        t_two = time.perf_counter() - t0
        assert all(torch.equal(o.cpu(), ref) for o in outs)
        print(f"round {rnd}: serial {2 * N * B / t_serial:7.1f} images/s ({t_serial / (2 * N) * 1e3:7.1f} ms per pass)   "
              f"two streams {2 * N * B / t_two:7.1f} images/s ({t_two / (2 * N) * 1e3:7.1f} ms per pass)   ratio {t_serial / t_two:.3f}", flush=True)


if __name__ == "__main__":
    main()
