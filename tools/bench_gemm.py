"""Time owc_gemm_bf16 on the Qwen2-VL shapes (HIP events on torch's current stream)."""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402

# timing experiments (`gemm_dbg` / `attn_dbg`: parts of a kernel switched off) exist only in the -DOWC_TIMING_KNOBS build
if any(a.startswith(("--dbg", "--timing")) or "_dbg" in a for a in sys.argv[1:]):
    from lmms_owc_amd import build as _owc_build

    _owc_build.build(verbose=False, timing=True)
    _lib.use_timing_library()

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
    ("vit25.gateup", 32768, 6912, 1280),     # Qwen2.5-VL vision MLP: gate / up of 3420 -> 3456 rows each (zero-padded at load), SwiGLU pairs
    ("vit25.down", 32768, 1280, 3456),       # ... and its down projection: K = 3420 padded to 3456 = 27 x 128 (the ping-pong kernel's K % 128)
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


def ab(dev, only, knob=b"gemm_big_min_m"):
    """Interleaved A/B of one owc_tuning_set knob (0 vs 1) in ONE process (median / best of 7 rounds each)."""
    import statistics

    lib = _lib.load()
    for name, m, n, k in SHAPES:
        if only and name != only:
            continue
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        res = {0: [], 1: []}
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for rnd in range(8):
            for v in (0, 1):
                lib.owc_tuning_set(knob, v)
                for _ in range(2):
                    ops.gemm_bf16(a, w, out=out)
                e0.record()
                for _ in range(10):
                    ops.gemm_bf16(a, w, out=out)
                e1.record()
                torch.cuda.synchronize()
                if rnd:
                    res[v].append(2.0 * m * n * k / (e0.elapsed_time(e1) / 10) / 1e9)
        lib.owc_tuning_set(knob, 0)
        print(f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  [{knob.decode()}] off med {statistics.median(res[0]):7.1f} max {max(res[0]):7.1f} | "
              f"on med {statistics.median(res[1]):7.1f} max {max(res[1]):7.1f}  ratio {statistics.median(res[1]) / statistics.median(res[0]):.3f}", flush=True)


def main():
    dev = torch.device("cuda:0")
    for a in sys.argv[1:]:   # --set=gemm_pingpong=0 : owc_tuning_set before anything runs
        if a.startswith("--set="):
            k, v = a[6:].split("=")
            assert _lib.load().owc_tuning_set(k.encode(), int(v)) == 0, k
    sweep = next((a[8:] for a in sys.argv[1:] if a.startswith("--sweep=")), None)
    if sweep:  # --sweep=<knob>:v1,v2,... [--m=<rows>] [prefix]: the values of one knob interleaved in one process, median of 5 rounds
        lib = _lib.load()
        knob, vals = sweep.split(":")
        vals = [int(v) for v in vals.split(",")]
        only = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
        m_over = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--m=")), 0)
        # --bias: with a bias vector; --epi=residual|quick_gelu: that epilogue (residual: added into the output in place)
        with_bias = "--bias" in sys.argv
        epi_name = next((a[6:] for a in sys.argv[1:] if a.startswith("--epi=")), "none")
        epi = {"none": ops.EPI_NONE, "quick_gelu": _lib.EPI_QUICK_GELU, "residual": _lib.EPI_RESIDUAL, "swiglu": ops.EPI_SWIGLU}[epi_name]
        for name, m, n, k in SHAPES:
            if only and not name.startswith(only):
                continue
            m = m_over or m
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            out = torch.zeros(m, n // 2 if epi_name == "swiglu" else n, dtype=torch.bfloat16, device=dev)
            bias = torch.randn(n, device=dev).to(torch.bfloat16) if with_bias else None
            kw = dict(epilogue=epi, residual=out if epi_name == "residual" else None)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            res = {}
            for rnd in range(6):
                for v in vals:
                    assert lib.owc_tuning_set(knob.encode(), v) == 0
                    for _ in range(2):
                        ops.gemm_bf16(a, w, bias, out=out, **kw)
                    e0.record()
                    for _ in range(10):
                        ops.gemm_bf16(a, w, bias, out=out, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        res.setdefault(v, []).append(e0.elapsed_time(e1) / 10 * 1e3)
            print(f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  [{knob}] " + "  ".join(
                f"{v}: {sorted(r)[len(r) // 2]:7.1f} us ({2.0 * m * n * k / sorted(r)[len(r) // 2] / 1e6:6.0f} TF)" for v, r in res.items()), flush=True)
        return
    if "--dbg" in sys.argv:  # timing experiments (--vals=0,512,4): 0 normal, 1 no DMA, 2 DMA re-reads K-tiles 0/1 (L2 hits), 4 no epilogue, 512 direct epilogue stores
        lib = _lib.load()
        only = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
        m_dbg = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--m=")), 0)
        epi_dbg = {"none": ops.EPI_NONE, "quick_gelu": _lib.EPI_QUICK_GELU}[next((a[6:] for a in sys.argv[1:] if a.startswith("--epi=")), "none")]
        for name, m, n, k in SHAPES:
            if only and name != only:
                continue
            m = m_dbg or m
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            res = {}
            for rnd in range(4):
                for v in [int(x) for x in next((a.split('=', 1)[1] for a in sys.argv[1:] if a.startswith('--vals=')), '0,4').split(',')]:
                    lib.owc_tuning_set(b"gemm_dbg", v)
                    for _ in range(2):
                        ops.gemm_bf16(a, w, out=out, epilogue=epi_dbg)
                    e0.record()
                    for _ in range(10):
                        ops.gemm_bf16(a, w, out=out, epilogue=epi_dbg)
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        res.setdefault(v, []).append(2.0 * m * n * k / (e0.elapsed_time(e1) / 10) / 1e9)
            lib.owc_tuning_set(b"gemm_dbg", 0)
            print(f"{name:18s} " + "  ".join(f"dbg{v}: {sorted(r)[len(r) // 2]:7.1f}" for v, r in res.items()), flush=True)
        return
    if "--vs-vendor" in sys.argv:
        # Interleaved A/B against the vendor library (hipBLASLt through torch.matmul) on the SAME operands in ONE process: rounds
        # alternate ours / vendor (rule 24 of the HIP guide: no ranking across processes or boxes).  A yardstick only - the vendor
        # library is never on the product path.  Output per shape: median and best TFLOP/s of both and the ratio of the medians.
        import statistics

        only = [a for a in sys.argv[1:] if not a.startswith("--")]
        m_over = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--m=")), 0)
        rounds = next((int(a[9:]) for a in sys.argv[1:] if a.startswith("--rounds=")), 7)
        for name, m, n, k in SHAPES:
            if only and name not in only:
                continue
            m = m_over if (m_over and not name.startswith("vit")) else m
            if name.startswith("vit") and m_over:
                m = 2 * m_over   # the vision launch group is twice the prefill group (131072 / 65536 rows)
            a = torch.randn(m, k, device=dev).to(torch.bfloat16)
            w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
            out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            wt = w.t()
            fns = {"ours": lambda: ops.gemm_bf16(a, w, out=out), "vendor": lambda: torch.matmul(a, wt, out=out)}
            res = {"ours": [], "vendor": []}
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = max(3, min(20, int(2e-2 / (2.0 * m * n * k / 1.4e15))))   # ~20 ms per timed burst
            for rnd in range(rounds + 1):
                for who in ("ours", "vendor"):
                    for _ in range(2):
                        fns[who]()
                    e0.record()
                    for _ in range(reps):
                        fns[who]()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        res[who].append(2.0 * m * n * k / (e0.elapsed_time(e1) / reps) / 1e9)
            mo, mv = statistics.median(res["ours"]), statistics.median(res["vendor"])
            print(f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  ours med {mo:7.1f} max {max(res['ours']):7.1f} | vendor med {mv:7.1f} "
                  f"max {max(res['vendor']):7.1f} TFLOP/s | ours / vendor {mo / mv:.3f}", flush=True)
        return
    if "--ab" in sys.argv:
        knob = next((a.split("=", 1)[1] for a in sys.argv[1:] if a.startswith("--knob=")), "gemm_dbg")
        return ab(dev, next((a for a in sys.argv[1:] if not a.startswith("--")), None), knob.encode())
    only = next((a for a in sys.argv[1:] if not a.startswith("--")), None)
    m_over = next((int(a[4:]) for a in sys.argv[1:] if a.startswith("--m=")), 0)  # e.g. --m=131072: the vision launch group
    swiglu = "--epi=swiglu" in sys.argv   # the epilogue the model's gate/up launch really uses (C is [M, N/2])
    epi = ops.EPI_SWIGLU if swiglu else ops.EPI_NONE
    iters = next((int(a[8:]) for a in sys.argv[1:] if a.startswith("--iters=")), 10)
    for name, m, n, k in SHAPES:
        if only and not name.startswith(only):
            continue
        m = m_over or m
        a = torch.randn(m, k, device=dev).to(torch.bfloat16)
        w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
        out = torch.empty(m, n // 2 if swiglu else n, dtype=torch.bfloat16, device=dev)
        for _ in range(3):
            ops.gemm_bf16(a, w, out=out, epilogue=epi)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.gemm_bf16(a, w, out=out, epilogue=epi)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        tf = 2.0 * m * n * k / ms / 1e9
        line = f"{name:18s} M={m:6d} N={n:6d} K={k:6d}  {ms:8.3f} ms  {tf:8.1f} TFLOP/s"
        if "--yardstick" in sys.argv:
            # vendor library (hipBLASLt through torch) on the same operands: a yardstick for the headroom, never on the product path
            for _ in range(3):
                torch.matmul(a, w.t(), out=out)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(iters):
                torch.matmul(a, w.t(), out=out)
            e1.record()
            torch.cuda.synchronize()
            ms2 = e0.elapsed_time(e1) / iters
            line += f"   | torch.matmul {ms2:8.3f} ms {2.0 * m * n * k / ms2 / 1e9:8.1f} TFLOP/s"
        print(line, flush=True)


if __name__ == "__main__":
    main()
