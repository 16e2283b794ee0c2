"""Time owc_attention_bf16 on the Qwen2-VL shapes."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402

# timing experiments (`gemm_dbg` / `attn_dbg`: parts of a kernel switched off) exist only in the -DOWC_TIMING_KNOBS build
if any(a.startswith(("--dbg", "--timing")) or "_dbg" in a for a in sys.argv[1:]):
    from lmms_owc_amd import build as _owc_build

    _owc_build.build(verbose=False, timing=True)
    _lib.use_timing_library()


def i32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)     # (seeded inputs: the output hashes printed at the end are comparable between builds - tools/ab_libs.sh)
    # vision: 64 images x 1024 tokens, 16 heads x 80
    n, L, H, hd = 64, 1024, 16, 80
    T, E = n * L, H * hd
    qkv = torch.randn(T, 3 * E, device=dev).to(torch.bfloat16)
    out = torch.empty(T, E, device=dev, dtype=torch.bfloat16)
    starts, lens = i32(np.arange(n) * L, dev), i32(np.full(n, L), dev)

    def vit():
        ops.attention(qkv, 3 * E, hd, qkv[:, E:], 3 * E, hd, qkv[:, 2 * E:], 3 * E, hd, out, E, hd, starts, starts, lens,
                      n_seq=n, n_heads=H, kv_group=1, head_dim=hd, max_q_len=L, causal=False, scale=hd ** -0.5)

    # the same tokens cut into fewer, longer images: 32 x 2048 and 16 x 4096 patches (4096 = the reference's max_pixels cap,
    # /root/reference/src/models/_qwen2_vl.py:64-65), and the cap leg's other grid, 54 x 74 = 3996 patches (not a multiple of 64)
    big = {}
    for n_b, L_b in ((32, 2048), (16, 4096), (16, 3996)):
        big[(n_b, L_b)] = (i32(np.arange(n_b) * L_b, dev), i32(np.full(n_b, L_b), dev))

    def vit_at(n_b, L_b):
        st_b, ln_b = big[(n_b, L_b)]

        def fn():
            ops.attention(qkv, 3 * E, hd, qkv[:, E:], 3 * E, hd, qkv[:, 2 * E:], 3 * E, hd, out, E, hd, st_b, st_b, ln_b,
                          n_seq=n_b, n_heads=H, kv_group=1, head_dim=hd, max_q_len=L_b, causal=False, scale=hd ** -0.5)
        return fn

    # decoder prefill: 114 prompts x 286, 28 q heads / 4 kv heads x 128
    nb, S, Hq, Hkv, smax = 114, 286, 28, 4, 302
    q = torch.randn(nb * S, (Hq + 2 * Hkv) * 128, device=dev).to(torch.bfloat16)
    kc = torch.randn(nb, Hkv, smax, 128, device=dev).to(torch.bfloat16)
    vc = torch.randn(nb, Hkv, smax, 128, device=dev).to(torch.bfloat16)
    o2 = torch.empty(nb * S, Hq * 128, device=dev, dtype=torch.bfloat16)
    st2, kst2, ln2 = i32(np.arange(nb) * S, dev), i32(np.arange(nb) * Hkv * smax, dev), i32(np.full(nb, S), dev)

    def prefill():
        ops.attention(q, (Hq + 2 * Hkv) * 128, 128, kc, 128, smax * 128, vc, 128, smax * 128, o2, Hq * 128, 128, st2, kst2, ln2,
                      n_seq=nb, n_heads=Hq, kv_group=Hq // Hkv, head_dim=128, max_q_len=S, causal=True, scale=128 ** -0.5)

    # LLaVA-NeXT-34B prefill: 24 prompts x 2388, 56 q heads / 8 kv heads x 128
    nb3, S3, Hq3, Hkv3 = 24, 2388, 56, 8
    q3 = torch.randn(nb3 * S3, (Hq3 + 2 * Hkv3) * 128, device=dev).to(torch.bfloat16)
    kc3 = torch.randn(nb3, Hkv3, S3 + 16, 128, device=dev).to(torch.bfloat16)
    vc3 = torch.randn(nb3, Hkv3, S3 + 16, 128, device=dev).to(torch.bfloat16)
    o3 = torch.empty(nb3 * S3, Hq3 * 128, device=dev, dtype=torch.bfloat16)
    st3, kst3, ln3 = i32(np.arange(nb3) * S3, dev), i32(np.arange(nb3) * Hkv3 * (S3 + 16), dev), i32(np.full(nb3, S3), dev)

    def prefill_long():
        ops.attention(q3, (Hq3 + 2 * Hkv3) * 128, 128, kc3, 128, (S3 + 16) * 128, vc3, 128, (S3 + 16) * 128, o3, Hq3 * 128, 128, st3, kst3,
                      ln3, n_seq=nb3, n_heads=Hq3, kv_group=Hq3 // Hkv3, head_dim=128, max_q_len=S3, causal=True, scale=128 ** -0.5)

    from lmms_owc_amd import _lib
    for kv in [a[6:] for a in sys.argv[1:] if a.startswith("--set=")]:     # --set=attn_waves=4 ...
        k, v = kv.split("=")
        _lib.check(_lib.load().owc_tuning_set(k.encode(), int(v)), 0)
        print("set", k, v)
    sweep = next((a[8:] for a in sys.argv[1:] if a.startswith("--sweep=")), None)
    if sweep:   # --sweep=<knob>:v0,v1,...  the vision shapes with the values of one knob INTERLEAVED in one process (median of 6 rounds)
        knob, vs = sweep.split(":")
        vs = [int(v) for v in vs.split(",")]
        shapes = [("vit hd80 64x1024", vit, 4.0 * n * H * L * L * hd)] + [(f"vit hd80 {n_b}x{L_b}", vit_at(n_b, L_b), 4.0 * n_b * H * L_b * L_b * hd)
                                                                          for n_b, L_b in big]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for name, fn, flops in shapes:
            res = {v: [] for v in vs}
            for rnd in range(7):
                for v in vs:
                    _lib.check(_lib.load().owc_tuning_set(knob.encode(), v), 0)
                    for _ in range(2):
                        fn()
                    e0.record()
                    for _ in range(10):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    if rnd:
                        res[v].append(e0.elapsed_time(e1) / 10)
            print(f"{name:24s} [{knob}] " + "  ".join(f"{v}: {sorted(r)[len(r) // 2]:7.3f} ms ({flops / sorted(r)[len(r) // 2] / 1e9:7.1f} TF)" for v, r in res.items()), flush=True)
        return
    vals = [0]
    for a in sys.argv[1:]:
        if a.startswith("--dbg="):
            vals = [int(v, 0) for v in a[6:].split(",")]
    for rep in range(2):
      for v in vals:
        if v or len(vals) > 1:
            _lib.load().owc_tuning_set(b"attn_dbg", v)
        print(f"-- attn_dbg = {v:#x}")
        bench_all(vit, prefill, prefill_long, n, H, L, hd, nb, Hq, S, nb3, Hq3, S3,
                  extra=[(f"vit hd80 {n_b}x{L_b}", vit_at(n_b, L_b), 4.0 * n_b * H * L_b * L_b * hd) for n_b, L_b in big])
    import hashlib

    torch.cuda.synchronize()
    for name, t in (("vit", out), ("prefill", o2), ("prefill_long", o3)):
        print(f"sha256 {name:13s} {hashlib.sha256(t.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]}")


def bench_all(vit, prefill, prefill_long, n, H, L, hd, nb, Hq, S, nb3, Hq3, S3, extra=()):
    for name, fn, flops in [("vit hd80 64x1024", vit, 4.0 * n * H * L * L * hd), *extra, ("prefill hd128 114x286 causal", prefill, 2.0 * nb * Hq * S * S * 128),
                            ("prefill hd128 24x2388 causal (llava-34b)", prefill_long, 2.0 * nb3 * Hq3 * S3 * S3 * 128)]:
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print(f"{name:42s} {ms:8.3f} ms  {flops / ms / 1e9:8.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
