"""Where the HOST time of one batch-1 image goes (the reference's own batch size): cProfile around Qwen2VLEngine.encode_images +
generate for one 448x448 image, 16 new tokens, after two warm-up calls.  usage: python tools/profile_host_batch1.py [model]"""
import cProfile
import pstats
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402

dev = torch.device("cuda:0")
d = DIMS[sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-7b"]
eng = Qwen2VLEngine(Qwen2VLWeights.random(d, dev, seed=1))
r = np.random.default_rng(0)
grid = [(1, 32, 32)]
pix = (torch.randn(1024, d.patch_k if hasattr(d, "patch_k") else 1176, device=dev) * 0.5).to(torch.bfloat16)
ids = np.concatenate([r.integers(1000, 30000, 15), np.full(256, d.image_token_id), r.integers(1000, 30000, 15)]).astype(np.int32)


def one():
    emb = eng.encode_images(pix, grid)
    out = eng.generate([ids], emb, [grid], 16)
    return out.cpu()


for _ in range(3):
    one()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    one()
torch.cuda.synchronize()
print(f"{(time.perf_counter() - t0) * 100:.2f} ms per image (wall, 10 images)")
pr = cProfile.Profile()
pr.enable()
for _ in range(10):
    one()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
