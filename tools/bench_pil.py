#!/usr/bin/env python3
"""The real-boundary leg of bench.py alone (PIL images -> Qwen2VL.generate_until -> strings) for host-side tuning:
   python tools/bench_pil.py [--batch 2048] [--threads 32] [--switch-interval 0.005]"""
import argparse
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=2048)
    ap.add_argument("--threads", default="32")
    ap.add_argument("--switch-interval", default="0.005")
    a = ap.parse_args()
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights

    device = torch.device("cuda", 0)
    dims = DIMS["qwen2-vl-7b"]
    engine = Qwen2VLEngine(Qwen2VLWeights.random(dims, device, seed=1234))
    B, T = a.batch, 16
    host_u8 = torch.randint(0, 256, (B, 3, 448, 448), dtype=torch.uint8)
    # engine-only reference: the same images as resident pixel_values
    pix = torch.randn((B * 1024, 1176), device=device, dtype=torch.bfloat16)
    ids = bench.prompt_ids(dims.image_token_id)

    def step():
        emb = engine.encode_images(pix, [(1, 32, 32)] * B)
        return engine.generate([ids] * B, emb, [[(1, 32, 32)]] * B, T, eos_token_id=-1, pad_token_id=0).cpu()

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    step()
    torch.cuda.synchronize()
    eng = 2 * B / (time.perf_counter() - t0)
    print(f"engine only: {eng:.1f} images/s", flush=True)
    for th in a.threads.split(","):
        for si in a.switch_interval.split(","):
            os.environ["OWC_PREP_THREADS"] = th
            sys.setswitchinterval(float(si))
            for rep in ("cold",):
                r = bench.pil_leg(engine, dims, host_u8, B, T, device, torch.cuda.synchronize)
                print(f"threads {th} switchinterval {si} [{rep}]: {r['images'] / r['seconds']:.1f} images/s ({r['images'] / r['seconds'] / eng:.3f} of "
                      f"engine), first prep {r['first_chunk_prep_s']:.2f} s, chunks {r['chunks']}", flush=True)


if __name__ == "__main__":
    main()
