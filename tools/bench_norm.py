"""Time owc_layernorm_bf16 / owc_rmsnorm_bf16 on the path's shapes (HIP events; GB/s = rows * d * 2 B read + written).
usage: bench_norm.py [--sweep=<knob>:v1,v2,...]   (the operand of every launch is the same buffer: up to ~250 MB it is served by the
Infinity Cache, so these rates are upper bounds of what the kernel reaches inside the model)"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402

SHAPES = [("vit.ln", "ln", 65536, 1280), ("vit.ln.16k", "ln", 16384, 1280), ("7b.rms prefill", "rms", 32604, 3584),
          ("7b.rms decode 2048", "rms", 2048, 3584), ("7b.rms decode 128", "rms", 128, 3584), ("72b.rms", "rms", 32604, 8192),
          ("2b.rms", "rms", 32604, 1536)]


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    sweep = next((a[8:] for a in sys.argv[1:] if a.startswith("--sweep=")), None)
    knob, vals = sweep.split(":") if sweep else ("", "0")
    vals = [int(v) for v in vals.split(",")]
    for name, kind, rows, d in SHAPES:
        x = torch.randn(rows, d, device=dev).to(torch.bfloat16)
        w = torch.randn(d, device=dev).to(torch.bfloat16)
        b = torch.randn(d, device=dev).to(torch.bfloat16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        res, outs = {}, {}
        for rnd in range(5):
            for v in vals:
                assert not knob or lib.owc_tuning_set(knob.encode(), v) == 0
                f = (lambda: ops.layernorm(x, w, b, 1e-6)) if kind == "ln" else (lambda: ops.rmsnorm(x, w, 1e-6))
                for _ in range(3):
                    y = f()
                e0.record()
                for _ in range(20):
                    y = f()
                e1.record()
                torch.cuda.synchronize()
                outs[v] = y
                if rnd:
                    res.setdefault(v, []).append(e0.elapsed_time(e1) / 20 * 1e3)
        same = all(torch.equal(outs[vals[0]], outs[v]) for v in vals)
        print(f"{name:20s} {rows:6d} x {d:5d}  [{knob}] " + "  ".join(
            f"{v}: {sorted(r)[len(r) // 2]:7.1f} us ({rows * d * 4 / sorted(r)[len(r) // 2] / 1e6:5.2f} TB/s)" for v, r in res.items())
            + f"  same bits: {same}", flush=True)
    if knob:
        lib.owc_tuning_set(knob.encode(), -1)


if __name__ == "__main__":
    main()
