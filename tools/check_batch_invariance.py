"""Tokens of prompt 0 inside a batch of B == the same prompt alone, over decode batch sizes (which GEMM / attention kernel the
dispatch picks must never show).  usage: python tools/check_batch_invariance.py [model] [B,...] [knob=value ...]"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib  # noqa: E402
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402

dev = torch.device("cuda:0")
d = DIMS[sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-2b"]
for kv in sys.argv[3:]:
    name, val = kv.split("=")
    _lib.check(_lib.load().owc_tuning_set(name.encode(), int(val)), 0)
    print("knob", name, val, flush=True)
eng = Qwen2VLEngine(Qwen2VLWeights.random(d, dev, seed=1234))
r = np.random.default_rng(0)
bad = 0
for B in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "2,33,64,65,128,256,512").split(",")]:
    prompts = [r.integers(1000, 30000, 286).astype(np.int32) for _ in range(B)]
    T = 8
    toks, logits = eng.generate(prompts, None, [[] for _ in prompts], T, eos_token_id=-1, forced_tokens=None, return_step_logits=True)
    alone_t, alone_l = eng.generate(prompts[:1], None, [[]], T, eos_token_id=-1, forced_tokens=toks[:1].cpu().numpy(), return_step_logits=True)
    # teacher-forced on the batch's tokens, so that every step's logits are comparable even after a flipped near-tie
    same = [bool(torch.equal(logits[j, 0], alone_l[j, 0])) for j in range(T)]
    print(f"B={B}: tokens equal {bool(torch.equal(toks[0], alone_t[0]))}, step logits bit-equal {same}", flush=True)
    bad += 0 if all(same) else 1
sys.exit(1 if bad else 0)
