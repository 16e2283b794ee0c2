"""Per-token decode latency of the 7B decoder at small batch sizes (the reference's default is batch size 1).
usage: bench_decode_latency.py <model> <batches> [bf16|fp8] [knob=value ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402

dev = torch.device("cuda:0")
d = DIMS[sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-7b"]
if len(sys.argv) > 3:  # decoder dtype: bf16 | fp8
    import dataclasses

    d = dataclasses.replace(d, decoder_dtype=sys.argv[3])
for kv in [a for a in sys.argv[4:] if "=" in a]:  # knob=value ... (owc_tuning_set); the word "graph" turns the hipGraph decode on
    from lmms_owc_amd import _lib

    name, val = kv.split("=")
    _lib.check(_lib.load().owc_tuning_set(name.encode(), int(val)), 0)
    print("knob", name, val, flush=True)
graph = any(a == "graph" for a in sys.argv[4:])
eng = Qwen2VLEngine(Qwen2VLWeights.random(d, dev, seed=1), graph_decode=graph)
print("graph_decode", graph, flush=True)
r = np.random.default_rng(0)
for B in ([int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else (1, 4, 16, 64, 256)):
    prompts = [r.integers(1000, 30000, 286).astype(np.int32) for _ in range(B)]
    ts = {}
    for T in (2, 34):
        eng.generate(prompts, None, [[] for _ in prompts], T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.generate(prompts, None, [[] for _ in prompts], T)
        torch.cuda.synchronize()
        ts[T] = (time.perf_counter() - t0) / 3
    per_tok = (ts[34] - ts[2]) / 32
    wbytes = eng.w.nbytes() - 2 * d.vocab * d.d_model * (0 if d.tie_embeddings else 1)
    print(f"B={B:4d}  {per_tok * 1e3:7.3f} ms/token-step  ({B / per_tok:9.1f} tok/s)  weight stream {wbytes / per_tok / 1e12:5.2f} TB/s", flush=True)
