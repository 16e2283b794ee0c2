"""Time owc_gemm_fp8 next to owc_gemm_bf16 on decoder shapes (one process, interleaved rounds)."""
import statistics
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import ops  # noqa: E402

SHAPES = [("7b.qkv", 18304, 4608, 3584), ("7b.gateup", 18304, 37888, 3584), ("7b.down", 18304, 3584, 18944),
          ("72b.qkv", 16384, 10240, 8192), ("72b.o", 16384, 8192, 8192), ("72b.gateup", 16384, 59136, 8192),
          ("72b.down", 16384, 8192, 29568), ("sq8192", 8192, 8192, 8192), ("72b.dec2k.gateup", 2048, 59136, 8192)]


def main():
    dev = torch.device("cuda:0")
    from lmms_owc_amd import _lib

    for a in sys.argv[1:]:   # --set=gemm_pingpong=0
        if a.startswith("--set="):
            k, v = a[6:].split("=")
            assert _lib.load().owc_tuning_set(k.encode(), int(v)) == 0
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    only = args[0] if args else None
    for name, m, n, k in SHAPES:
        if only and name != only:
            continue
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        a8, sa = ops.quantize_rows_fp8(a)
        w8, sw = ops.quantize_rows_fp8(w)
        res = {"bf16": [], "fp8": [], "quant": []}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rnd in range(6):
            for kind in ("bf16", "fp8", "quant"):
                fn = {"bf16": lambda: ops.gemm_bf16(a, w, out=out), "fp8": lambda: ops.gemm_fp8(a8, sa, w8, sw, out=out),
                      "quant": lambda: ops.quantize_rows_fp8(a, out=a8, scale=sa)}[kind]
                for _ in range(2):
                    fn()
                e0.record()
                for _ in range(10):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[kind].append(e0.elapsed_time(e1) / 10)
        fl = 2.0 * m * n * k / 1e9
        t16, t8, tq = (statistics.median(res[x]) for x in ("bf16", "fp8", "quant"))
        print(f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  bf16 {t16:7.3f} ms {fl / t16:7.1f} TF | fp8 {t8:7.3f} ms {fl / t8:7.1f} TF "
              f"(x{t16 / t8:.2f}) | quant A {tq * 1e3:7.1f} us = {m * k * 3 / tq / 1e6:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    main()
