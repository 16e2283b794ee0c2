"""Per-tile fixed cost of the 256x256 GEMM: time vs K at fixed M = N (4 full rounds of 256 tiles) -> t = rounds * (a + b * nk)."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402
from lmms_owc_amd import build as _owc_build  # noqa: E402

_owc_build.build(verbose=False, timing=True)   # `gemm_dbg` exists only in the -DOWC_TIMING_KNOBS build
_lib.use_timing_library()

dev = torch.device("cuda:0")
lib = _lib.load()
M = N = 8192
for dbg in [int(x) for x in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["0", "4"])]:
    lib.owc_tuning_set(b"gemm_dbg", dbg)
    xs, ts = [], []
    for K in (512, 1024, 2048, 4096):
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        best = []
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for r in range(5):
            for _ in range(3):
                ops.gemm_bf16(a, w, out=out)
            e0.record()
            for _ in range(20):
                ops.gemm_bf16(a, w, out=out)
            e1.record()
            torch.cuda.synchronize()
            best.append(e0.elapsed_time(e1) / 20 * 1e3)
        t = float(np.median(best))
        xs.append(K // 64)
        ts.append(t)
        print(f"dbg{dbg} K={K:5d} nk={K // 64:4d}  {t:8.1f} us  {2.0 * M * N * K / t / 1e6:7.1f} TF", flush=True)
    b, a0 = np.polyfit(xs, ts, 1)
    print(f"dbg{dbg}: per launch fixed {a0:.1f} us (= {a0 / 4:.2f} us per tile round), per K-tile {b / 4:.3f} us -> asymptote {2.0 * 256 * 256 * 64 * 256 / (b / 4) / 1e6:.0f} TF", flush=True)
lib.owc_tuning_set(b"gemm_dbg", 0)
