"""Determinism under contention: P processes share cuda:0, each repeats the same vision tower / generate calls and compares every
repeat bit for bit with its own first result (a timing-dependent race shows as a repeat that differs).
usage: python tools/contention_stress.py [procs=2] [iters=6] [model=qwen2-vl-2b] [batch=64] [knob=value ...]"""
import multiprocessing as mp
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def worker(rank, args, knobs, q):
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch

    from lmms_owc_amd import _lib
    from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights

    dev = torch.device("cuda:0")
    for k, v in knobs:
        _lib.check(_lib.load().owc_tuning_set(k.encode(), int(v)), 0)
    d = DIMS[args["model"]]
    if "depth" in args:
        import dataclasses

        d = dataclasses.replace(d, v_depth=int(args["depth"]))
    eng = Qwen2VLEngine(Qwen2VLWeights.random(d, dev, seed=1234))
    B, T = int(args["batch"]), 8
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pix = torch.randn((B * 1024, 1176), generator=gen, device=dev, dtype=torch.bfloat16)
    r = np.random.default_rng(5)
    ids = np.concatenate([r.integers(1000, 30000, 14), np.full(256, d.image_token_id), r.integers(1000, 30000, 16)]).astype(np.int32)
    grids = [[(1, 32, 32)]] * B
    flat = [(1, 32, 32)] * B
    first = {}
    bad = {"emb": 0, "emb1": 0, "logits_batch": 0, "toks_batch": 0, "logits_alone": 0, "toks_alone": 0}
    if args["mode"] == "vit1":   # the single-image vision tower only, many repeats
        n_img = int(args.get("images", 1))
        from collections import Counter
        outcomes = Counter()
        wts = (torch.arange(256 * n_img * d.d_model, device=dev, dtype=torch.int64) % 65521 + 1)
        for it in range(int(args["iters"])):
            e = eng.encode_images(pix[:1024 * n_img], flat[:n_img])
            outcomes[int((e.view(torch.int16).flatten().to(torch.int64) * wts).sum())] += 1
        nbad = int(args["iters"]) - max(outcomes.values())   # runs that are not the majority result
        q.put((rank, {"vit1_bad": nbad, "iters": int(args["iters"]), "distinct": len(outcomes)}))
        return
    for it in range(int(args["iters"])):
        emb = eng.encode_images(pix, flat)
        toks, logits = eng.generate([ids] * B, emb, grids, T, eos_token_id=-1, return_step_logits=True)
        emb1 = eng.encode_images(pix[:1024], flat[:1])
        t1, l1 = eng.generate([ids], emb1, grids[:1], T, eos_token_id=-1, return_step_logits=True)
        torch.cuda.synchronize()
        cur = {"emb": emb, "emb1": emb1, "logits_batch": logits, "toks_batch": toks, "logits_alone": l1, "toks_alone": t1}
        for k, v in cur.items():
            if it == 0:
                first[k] = v.clone()
            elif not torch.equal(v, first[k]):
                bad[k] += 1
                if k.startswith("logits"):
                    steps = [j for j in range(T) if not torch.equal(v[j], first[k][j])]
                    rows = sorted(set((v != first[k]).nonzero()[:, 1].tolist()))[:8]
                    print(f"rank {rank} iter {it}: {k} differs at steps {steps}, batch rows {rows} ...", flush=True)
                elif k.startswith("emb"):
                    rows = (v != first[k]).any(dim=1).nonzero().flatten()
                    print(f"rank {rank} iter {it}: {k} differs in {rows.numel()} rows, first {rows[:6].tolist()} (image {int(rows[0]) // 256})", flush=True)
        if it == 0:
            inv = torch.equal(first["emb"][:256], first["emb1"]), torch.equal(first["logits_batch"][:, 0], first["logits_alone"][:, 0])
            print(f"rank {rank}: first pass, image 0 in batch == alone: emb {inv[0]}, logits {inv[1]}", flush=True)
    q.put((rank, bad))


if __name__ == "__main__":
    args = {"procs": "2", "iters": "6", "model": "qwen2-vl-2b", "batch": "64", "mode": "full", "images": "1"}
    knobs = []
    for a in sys.argv[1:]:
        k, v = a.split("=")
        if k in args or k in ("depth",):
            args[k] = v
        else:
            knobs.append((k, v))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, args, knobs, q)) for r in range(int(args["procs"]))]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
    while not q.empty():
        print(q.get())
