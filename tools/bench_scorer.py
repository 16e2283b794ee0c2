"""The label-cosine leg alone, swept over the class count (C in {100, 397, 1000, 10000}) and the label count.
usage: python tools/bench_scorer.py [classes=100,397,1000,10000] [labels=65536] [knob=value ...]"""
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd.engine.scorer import MINILM_L6, BertWeights, SentenceScorer  # noqa: E402

args = dict(a.split("=") for a in sys.argv[1:])
for k in [k for k in args if k not in ("classes", "labels")]:   # anything else: owc_tuning_set(knob, value), e.g. bert_w3=0
    from lmms_owc_amd import _lib
    _lib.check(_lib.load().owc_tuning_set(k.encode(), int(args[k])), 0)
    print("knob", k, args[k], flush=True)
classes = [int(c) for c in args.get("classes", "100,397,1000,10000").split(",")]
n_lab, L = int(args.get("labels", 65536)), 16
dev = torch.device("cuda:0")
scorer = SentenceScorer(BertWeights.random(MINILM_L6, dev, seed=7), max_batch=16384)
lr = np.random.default_rng(99)
ids = lr.integers(1000, 30000, (max(n_lab, max(classes)), L)).astype(np.int32)
lens = lr.integers(2, L + 1, ids.shape[0])
mask = (np.arange(L)[None, :] < lens[:, None]).astype(np.int32)
for C in classes:
    cls_z = scorer.embed(ids[:C], mask[:C])
    label = torch.from_numpy(lr.integers(0, C, n_lab).astype(np.int32)).to(dev)
    z = scorer.embed(ids[:n_lab], mask[:n_lab])
    for what in ("embed + top-5", "top-5 only"):
        def run():
            zz = scorer.embed(ids[:n_lab], mask[:n_lab]) if what.startswith("embed") else z
            return scorer.topk(zz, cls_z, 5, label)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 3
        print(f"C={C:6d} N={n_lab}: {what:14s} {dt * 1e3:8.2f} ms  {n_lab / dt / 1e3:9.1f} k labels/s"
              + (f"  cosine {2.0 * n_lab * C * 384 / dt / 1e12:6.1f} TFLOP/s" if what.startswith("top") else ""), flush=True)
