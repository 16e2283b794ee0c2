"""CPU check of the width-slice parity cases (tests/test_decode_parity_gpu.py): how many steps of each checked sequence have a
DECISIVE top-2 margin in the ORACLE's logits (> 2 x the logit bound), i.e. how many token comparisons the GPU test will bind on.
Run after changing OUTLIER_* or the case seeds:   python tools/slice_margins.py [name ...]"""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from tests import test_decode_parity_gpu as T  # noqa: E402

CASES = [("2b", 8, 8, "bf16"), ("2b", 200, 8, "bf16"), ("2b", 1280, 8, "bf16"), ("7b", 8, 6, "bf16"), ("7b", 300, 6, "bf16"),
         ("yi34b", 8, 6, "bf16"), ("yi34b", 130, 6, "bf16"), ("72b", 8, 6, "bf16"), ("72b", 130, 6, "bf16"),
         ("72b", 8, T.FP8_STEPS, "fp8"), ("72b", 130, T.FP8_STEPS, "fp8"), ("72b", 8, T.FP8_STEPS, "fp8", 128.0),
         ("72b", 130, T.FP8_STEPS, "fp8", 128.0), ("72b", 130, T.FP8_STEPS, "fp8", 1024.0), ("q25_3b", 8, 6, "bf16"), ("q25_3b", 300, 6, "bf16")]
only = set(sys.argv[1:])
for name, B, steps, dt, *ch in CASES:
    if only and name not in only and dt not in only:
        continue
    kw = dict(channel_scale=ch[0] if ch else 0.0) if dt == "fp8" else {}
    t0 = time.time()
    frac = T.FP8_OUTLIER_BOUND if dt == "fp8" else T.OUTLIER_BOUND   # the decisive margin is 2 x the scaled rows' bound
    *_, check, forced, refs = T._slice_refs(name, B, steps, dt, **kw)
    for b in check:
        lg = np.asarray(refs[b][1], np.float32)
        m = [float((np.sort(x)[-1] - np.sort(x)[-2]) / np.abs(x).max()) for x in lg]
        print(f"{name:6s} {dt}{' channels x %g' % ch[0] if ch else ''} B={B:5d} seq {b:5d}: decisive (> {2 * frac:.2f}) {sum(x > 2 * frac for x in m)} / {steps}   margins {np.round(m, 3)}  tokens {list(refs[b][0])}", flush=True)
    print(f"   ({time.time() - t0:.1f} s)")
