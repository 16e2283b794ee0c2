"""Row 0 of a GEMM must not depend on how many other rows ride along (which kernel the dispatch picks): the decoder / vision
shapes of a model at a list of M.  usage: python tools/check_gemm_invariance.py [2b|7b]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
model = sys.argv[1] if len(sys.argv) > 1 else "2b"
d, ff, kv, vocab = {"2b": (1536, 8960, 256, 151936), "7b": (3584, 18944, 512, 152064)}[model]
shapes = [("qkv", d + 2 * kv, d, "bias"), ("o", d, d, "residual"), ("gateup", 2 * ff, d, "swiglu"), ("down", d, ff, "residual"),
          ("lm_head", vocab, d, "none"), ("vit.qkv", 3840, 1280, "bias"), ("vit.fc1", 5120, 1280, "quick_gelu"),
          ("vit.fc2", 1280, 5120, "residual"), ("vit.patch", 1280, 1176, "none")]
Ms = [1, 8, 32, 33, 64, 65, 128, 286, 512, 1024, 2048, 18304]
if len(sys.argv) > 2:   # e.g. 1,700,1025,1424,1793,2048: the row counts a compacting decode batch walks through
    Ms = [int(x) for x in sys.argv[2].split(",")]
g = torch.Generator(device=dev).manual_seed(5)
bad = 0
for name, n, k, epi in shapes:
    a = torch.randn(max(Ms), k, device=dev, generator=g).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(n, device=dev, generator=g).to(torch.bfloat16)
    r = torch.randn(max(Ms), n, device=dev, generator=g).to(torch.bfloat16)
    E = {"bias": _lib.EPI_NONE, "none": _lib.EPI_NONE, "residual": _lib.EPI_RESIDUAL, "swiglu": _lib.EPI_SWIGLU, "quick_gelu": _lib.EPI_QUICK_GELU}[epi]
    ref = None
    for m in Ms:
        out = ops.gemm_bf16(a[:m], w, None if epi in ("none", "swiglu") else b, epilogue=E, residual=r[:m] if epi == "residual" else None)
        row = out[0].clone()
        if ref is None:
            ref = row
            full = ops.gemm_bf16(a, w, None if epi in ("none", "swiglu") else b, epilogue=E, residual=r if epi == "residual" else None)
        elif not torch.equal(row, ref):
            bad += 1
            print(f"{model} {name} N={n} K={k} {epi}: row 0 at M={m} differs from M={Ms[0]} in {(row != ref).sum().item()} of {row.numel()} elements", flush=True)
        if not torch.equal(out, full[:m]):   # EVERY row of the M-row launch against the same rows of the largest launch
            bad += 1
            rows = (out != full[:m]).any(dim=1).nonzero().flatten()
            print(f"{model} {name} N={n} K={k} {epi}: M={m}: {len(rows)} rows differ from the M={max(Ms)} launch, first {rows[:6].tolist()}", flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
