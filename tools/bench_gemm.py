"""Time owc_gemm_bf16 on the Qwen2-VL shapes (HIP events on torch's current stream)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402

SHAPES = [
    ("vit.qkv", 32768, 3840, 1280),
    ("vit.proj", 32768, 1280, 1280),
    ("vit.fc1", 32768, 5120, 1280),
    ("vit.fc2", 32768, 1280, 5120),
    ("vit.patch", 32768, 1280, 1176),
    ("7b.qkv", 18304, 4608, 3584),
    ("7b.o", 18304, 3584, 3584),
    ("7b.gateup", 18304, 37888, 3584),
    ("7b.down", 18304, 3584, 18944),
    ("vit.qkv.x8", 32768, 4096, 1280),
    ("vit.qkv.k5120", 32768, 4096, 5120),
    ("sq4096", 4096, 4096, 4096),
    ("sq8192", 8192, 8192, 8192),
    ("7b.dec2k.qkv", 2048, 4608, 3584),
    ("7b.dec2k.o", 2048, 3584, 3584),
    ("7b.dec2k.gateup", 2048, 37888, 3584),
    ("7b.dec2k.down", 2048, 3584, 18944),
    ("7b.dec4k.qkv", 4096, 4608, 3584),
    ("7b.dec4k.down", 4096, 3584, 18944),
    ("7b.decode.qkv", 512, 4608, 3584),
    ("7b.decode.gateup", 512, 37888, 3584),
    ("7b.decode.down", 512, 3584, 18944),
]


def main():
    dev = torch.device("cuda:0")
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, m, n, k in SHAPES:
        if only and name != only:
            continue
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm_bf16(a, w, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 10
        e0.record()
        for _ in range(iters):
            ops.gemm_bf16(a, w, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * m * n * k / ms / 1e9
        print(f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  {ms:8.3f} ms  {tf:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
