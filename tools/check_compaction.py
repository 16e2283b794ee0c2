"""EOS-aware row compaction must not change a token: `generate(..., compact_rows=True)` against `compact_rows=False` on seeded ragged
answer lengths (forced continuations), with the first differing (row, step) and the live-row count of that step when they differ.
usage: python tools/check_compaction.py [model] [B] [T] [prompt_len | img] [compact | plain]
(img: the bench's 286-token image prompt, random embeddings; a 5th argument runs ONE mode three times and exits - for
`rocprofv3 --kernel-trace --stats`, profiles/r04_decode_loop_*_kernel_stats.csv)"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from bench import EOS_ID, ragged_answer_lengths  # noqa: E402
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402

dev = torch.device("cuda:0")
d = DIMS[sys.argv[1] if len(sys.argv) > 1 else "qwen2-vl-7b"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
T = int(sys.argv[3]) if len(sys.argv) > 3 else 32
img = len(sys.argv) > 4 and sys.argv[4] == "img"
S = int(sys.argv[4]) if len(sys.argv) > 4 and not img else 40
eng = Qwen2VLEngine(Qwen2VLWeights.random(d, dev, seed=1234))
r = np.random.default_rng(0)
emb = None
if img:
    from bench import prompt_ids

    prompts = [prompt_ids(d.image_token_id)] * B
    none = [[(1, 32, 32)]] * B
    emb = (torch.randn((B * 256, d.d_model), device=dev, generator=torch.Generator(device=dev).manual_seed(3)) * 0.5).to(torch.bfloat16)
else:
    prompts = [r.integers(1000, 30000, S).astype(np.int32) for _ in range(B)]
    none = [[] for _ in prompts]
forced, lens = ragged_answer_lengths(B, T, 8.0, 0.01, 7)
kw = dict(eos_token_id=EOS_ID, pad_token_id=0, forced_tokens=forced)
if len(sys.argv) > 5:
    import time

    mode = sys.argv[5] == "compact"
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.generate(prompts, emb, none, T, compact_rows=mode, **kw).cpu()
        print(f"{sys.argv[5]}: {time.perf_counter() - t0:.3f} s per generate (prefill + {T - 1} decode steps)", flush=True)
    sys.exit(0)
sa, sb = {}, {}
a = eng.generate(prompts, emb, none, T, compact_rows=False, stats=sa, **kw).cpu().numpy()
a2 = eng.generate(prompts, emb, none, T, compact_rows=False, **kw).cpu().numpy()
b = eng.generate(prompts, emb, none, T, compact_rows=True, stats=sb, **kw).cpu().numpy()
b2 = eng.generate(prompts, emb, none, T, compact_rows=True, **kw).cpu().numpy()
print("plain deterministic:", np.array_equal(a, a2), " compacted deterministic:", np.array_equal(b, b2))
live = sb["live_rows_per_step"]
print("live rows per step:", live)
diff = np.argwhere(a != b)
print("differing (row, step) pairs:", len(diff))
by_step = {}
for row, step in diff:
    by_step.setdefault(int(step), []).append(int(row))
for step in sorted(by_step):
    rows = by_step[step]
    print(f"  step {step}: live rows {live[step] if step < len(live) else '?'}: {len(rows)} rows differ, e.g. rows {rows[:8]} plain {a[rows[0], step]} compact {b[rows[0], step]}")
sys.exit(1 if len(diff) else 0)
