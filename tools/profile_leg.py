#!/usr/bin/env python3
"""One leg of bench.py on its own, for rocprofv3 (`--kernel-trace --stats`): the kernels of the headline step and of the other
legs stay out of the summary.

  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cap -- python3 tools/profile_leg.py max_pixels 256
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_mix -- python3 tools/profile_leg.py config3 1024
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_dec -- python3 tools/profile_leg.py decode 2048     (N decode steps at that many rows)
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from lmms_owc_amd import _lib  # noqa: E402
from lmms_owc_amd.engine.qwen2vl import DIMS, Qwen2VLEngine, Qwen2VLWeights  # noqa: E402


def main() -> None:
    what, n = sys.argv[1], int(sys.argv[2])
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    dims = DIMS["qwen2-vl-7b"]
    engine = Qwen2VLEngine(Qwen2VLWeights.random(dims, device, seed=1234))
    lib, ctx = _lib.load(), _lib.ctx(0)
    NK = len(_lib.PROF_KINDS)
    import ctypes as C

    def profile(on: bool):
        if on:
            lib.owc_gemm_profile_enable(ctx, 1)
            return None
        ms, wk, cnt = (C.c_double * NK)(), (C.c_double * NK)(), (C.c_int64 * NK)()
        _lib.check(lib.owc_profile_read(ctx, NK, ms, wk, cnt), 0)
        lib.owc_gemm_profile_enable(ctx, 0)
        return {k: {"ms": ms[i], "work": wk[i], "launches": int(cnt[i])} for i, k in enumerate(_lib.PROF_KINDS)}

    if what == "decode":
        r = np.random.default_rng(0)
        prompts = [r.integers(1000, 30000, 286).astype(np.int32) for _ in range(n)]
        none = [[] for _ in prompts]
        engine.generate(prompts, None, none, 2)
        torch.cuda.synchronize()
        engine.generate(prompts, None, none, 10)     # 1 prefill + 9 decode steps
        torch.cuda.synchronize()
        print(json.dumps({"leg": "decode", "rows": n, "decode_steps": 9}))
        return
    out = bench.ragged_leg(engine, dims, what, n, 16, 1, device, torch.cuda.synchronize, profile=profile)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
