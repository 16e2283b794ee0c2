"""The four decoder projections of ONE layer at decode-step row counts, as weight STREAMS: every launch reads a different copy of
the weight matrix (enough copies to exceed the 256 MB Infinity Cache several times over, like the 28 layers of a real step), with
the epilogue the decode step really uses.  Prints microseconds per launch and TB/s of weight bytes.

  python tools/bench_decode_gemms.py [--model 7b] [--m 1,8,32,128,256] [--dtype=bf16|fp8] [--set knob=v ...] [--ab knob=v0,v1,...]
"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402
from lmms_owc_amd.engine.qwen2vl import DIMS  # noqa: E402

EPI_RESIDUAL = 3


def arg(name, default):
    return next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith(f"--{name}=")), default)


def main():
    dev = torch.device("cuda:0")
    d = DIMS[f"qwen2-vl-{arg('model', '7b')}"]
    ms = [int(x) for x in arg("m", "1,8,32,128,256").split(",")]
    lib = _lib.load()
    for kv in [a[6:] for a in sys.argv[1:] if a.startswith("--set=")]:
        k, v = kv.split("=")
        _lib.check(lib.owc_tuning_set(k.encode(), int(v)), 0)
        print("set", k, v)
    ab = arg("ab", None)
    ab_knob, ab_vals = (ab.split("=")[0], [int(x) for x in ab.split("=")[1].split(",")]) if ab else (None, [None])
    qkv_n = (d.n_q_heads + 2 * d.n_kv_heads) * d.head_dim
    shapes = [("qkv", qkv_n, d.d_model, ops.EPI_NONE), ("o", d.d_model, d.n_q_heads * d.head_dim, EPI_RESIDUAL),
              ("gateup", 2 * d.d_ff, d.d_model, ops.EPI_SWIGLU), ("down", d.d_model, d.d_ff, EPI_RESIDUAL)]
    only = arg("only", None)
    fp8 = arg("dtype", "bf16") == "fp8"
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for name, n, k, epi in shapes:
        if only and name not in only.split(","):
            continue
        wbytes = n * k * 2
        copies = max(4, int(1.5e9 // wbytes))
        if fp8:
            wbytes //= 2
            copies = max(4, int(1.5e9 // wbytes))
        ws = [(torch.randn(n, k, device=dev) * 0.02).to(torch.bfloat16) for _ in range(copies)]
        if fp8:
            ws = [ops.quantize_rows_fp8(w) for w in ws]
        bias = torch.zeros(n, device=dev, dtype=torch.bfloat16) if name == "qkv" else None
        for m in ms:
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            res = torch.randn(m, n, device=dev).to(torch.bfloat16) if epi == EPI_RESIDUAL else None
            out = torch.empty(m, n // 2 if epi == ops.EPI_SWIGLU else n, dtype=torch.bfloat16, device=dev)
            if fp8:
                a8, sa = ops.quantize_rows_fp8(a)

            def gemm(w):
                if fp8:
                    return ops.gemm_fp8(a8, sa, w[0], w[1], bias, epilogue=epi, residual=res, out=out)
                return ops.gemm_bf16(a, w, bias, epilogue=epi, residual=res, out=out)

            line = f"{name:7s} M={m:4d} N={n:6d} K={k:6d} "
            for v in ab_vals:
                if ab_knob:
                    _lib.check(lib.owc_tuning_set(ab_knob.encode(), v), 0)
                best = []
                for rnd in range(4):
                    for w in ws[:2]:
                        gemm(w)
                    e0.record()
                    for i in range(2 * copies):
                        gemm(ws[i % copies])
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        best.append(e0.elapsed_time(e1) * 1e3 / (2 * copies))
                us = sorted(best)[len(best) // 2]
                line += f" | {ab_knob + '=' + str(v) if ab_knob else ''} {us:7.1f} us {wbytes / us / 1e6:5.2f} TB/s"
            print(line, flush=True)
        del ws


if __name__ == "__main__":
    main()
