#!/usr/bin/env python3
"""The plug-in boundary of the LLaVA path: PIL images -> LLaVA.generate_until -> strings (host: CLIP resize / crop or anyres tiling,
prompt ids; GPU: owc_clip_patchify_u8 + CLIP tower + projector + prefill + decode).  Not comparable 1:1 with tools/bench_llava.py's
engine-only rate: the synthetic byte tokenizer spells the Vicuna system prompt and the question out one token per character
(S ~ 820 against the bench's 624), i.e. ~30 % more prefill rows per image.
   python tools/bench_llava_pil.py [--model llava-1.5-7b] [--images 2048] [--batch 512] [--size 480x640]"""
import argparse
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="llava-1.5-7b")
    ap.add_argument("--images", type=int, default=2048)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--size", default="480x640")
    ap.add_argument("--new-tokens", type=int, default=16)
    a = ap.parse_args()
    from PIL import Image

    from lmms_owc_amd.models import get_model
    from lmms_owc_amd.tasks import ClassificationTask

    h, w = (int(x) for x in a.size.split("x"))
    r = np.random.default_rng(0)
    base = [Image.fromarray(r.integers(0, 256, (h, w, 3), dtype=np.uint8), "RGB") for _ in range(64)]
    docs = [{"visual": base[i % 64], "target": f"class_{i % 10}"} for i in range(a.images)]
    task = ClassificationTask("bench", docs, generation_kwargs={"max_new_tokens": a.new_tokens, "do_sample": False})
    lm = get_model("custom-model", model_type="llava", model_name_or_path=f"synthetic:{a.model}", batch_size=1, engine_batch=a.batch)
    lm._tokenizer.eos_token_id = -1          # forced length, like the engine-only bench
    lm.task_dict["bench"] = task.dataset
    task.build_all_requests(limit=None, rank=0, world_size=1)
    lm.generate_until(task.instances[:64])
    task.build_all_requests(limit=None, rank=0, world_size=1)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = lm.generate_until(task.instances)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert len(out) == a.images
    print(f"{a.model}: {a.images} PIL {a.size} images -> strings in {dt:.2f} s = {a.images / dt:.1f} images/s (engine batch {a.batch})")


if __name__ == "__main__":
    main()
