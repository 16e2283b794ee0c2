"""Does the C row stride (ldc) change the epilogue cost?  Same GEMM, output rows padded by 0 / 64 / 128 / 192 elements."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
for name, m, n, k, epi in (("vit.fc1", 32768, 5120, 1280, ops.EPI_NONE), ("vit.qkv", 32768, 3840, 1280, ops.EPI_NONE),
                           ("vit.proj", 32768, 1280, 1280, ops.EPI_NONE), ("7b.gateup", 18304, 37888, 3584, ops.EPI_SWIGLU),
                           ("7b.qkv", 18304, 4608, 3584, ops.EPI_NONE), ("7b.o", 18304, 3584, 3584, ops.EPI_NONE)):
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    nout = n // 2 if epi == ops.EPI_SWIGLU else n
    res = {}
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bufs = {pad: torch.empty(m, nout + pad, dtype=torch.bfloat16, device=dev)[:, :nout] for pad in (0, 64, 128, 192, 8)}
    for rnd in range(5):
        for pad, out in bufs.items():
            for _ in range(2):
                ops.gemm_bf16(a, w, out=out, epilogue=epi)
            e0.record()
            for _ in range(10):
                ops.gemm_bf16(a, w, out=out, epilogue=epi)
            e1.record()
            torch.cuda.synchronize()
            if rnd:
                res.setdefault(pad, []).append(2.0 * m * n * k / (e0.elapsed_time(e1) / 10) / 1e9)
    print(f"{name:10s} ldc = N + pad: " + "  ".join(f"+{p}: {sorted(r)[len(r) // 2]:7.1f}" for p, r in res.items()), flush=True)
