"""Op-level determinism under contention: P processes share cuda:0 and repeat each op of the single-image vision tower
(T = 1024 tokens, E = 1280, 16 heads x 80, F = 5120); a repeat that differs from the first result bit for bit is a race.
usage: python tools/contention_ops.py [procs=2] [iters=400] [T=1024]"""
import multiprocessing as mp
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def worker(rank, args, q):
    sys.path.insert(0, str(ROOT))
    import torch

    from lmms_owc_amd import _lib, ops

    dev = torch.device("cuda:0")
    T, E, H, hd, F = int(args["T"]), 1280, 16, 80, 5120
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    x, w_qkv, b_qkv = rn(T, E), rn(3 * E, E, sc=0.03), rn(3 * E)
    w_proj, b_proj = rn(E, E, sc=0.03), rn(E)
    w_fc1, b_fc1, w_fc2, b_fc2 = rn(F, E, sc=0.03), rn(F), rn(E, F, sc=0.02), rn(E)
    pix, w_patch = rn(T, 1176), rn(E, 1176, sc=0.03)
    ln_w, ln_b = rn(E), rn(E)
    qkv = ops.gemm_bf16(x, w_qkv, b_qkv)
    mlp = ops.gemm_bf16(x, w_fc1, b_fc1, epilogue=_lib.EPI_QUICK_GELU)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)  # noqa: E731
    n_seq = max(1, T // 1024)
    starts, lens = i32([i * (T // n_seq) for i in range(n_seq)]), i32([T // n_seq] * n_seq)
    attn_out = torch.empty(T, E, dtype=torch.bfloat16, device=dev)

    def attention():
        o = torch.empty_like(attn_out)
        ops.attention(qkv, 3 * E, hd, qkv[:, E:], 3 * E, hd, qkv[:, 2 * E:], 3 * E, hd, o, E, hd, starts, starts, lens,
                      n_seq=n_seq, n_heads=H, kv_group=1, head_dim=hd, max_q_len=T // n_seq, causal=False, scale=hd ** -0.5)
        return o

    def proj_inplace():
        xx = x.clone()
        return ops.gemm_bf16(qkv[:, :E].contiguous(), w_proj, b_proj, epilogue=_lib.EPI_RESIDUAL, residual=xx, out=xx)

    def fc2_inplace():
        xx = x.clone()
        return ops.gemm_bf16(mlp, w_fc2, b_fc2, epilogue=_lib.EPI_RESIDUAL, residual=xx, out=xx)

    table = {
        "layernorm": lambda: ops.layernorm(x, ln_w, ln_b, 1e-6),
        "gemm qkv (bias)": lambda: ops.gemm_bf16(x, w_qkv, b_qkv),
        "attention hd80": attention,
        "gemm proj (residual, in place)": proj_inplace,
        "gemm fc1 (quick_gelu)": lambda: ops.gemm_bf16(x, w_fc1, b_fc1, epilogue=_lib.EPI_QUICK_GELU),
        "gemm fc2 (residual, in place)": fc2_inplace,
        "gemm patch (K=1176)": lambda: ops.gemm_bf16(pix, w_patch),
    }
    res = {}
    for name, fn in table.items():
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        burst = int(args["burst"])   # launches queued back to back before the host looks (so that the processes really contend)
        for it in range(int(args["iters"]) // burst):
            outs = [fn() for _ in range(burst)]
            torch.cuda.synchronize()
            for o in outs:
                if not torch.equal(o, ref):
                    bad += 1
                    if bad == 1:
                        d = (o != ref)
                        print(f"rank {rank} {name}: round {it}: {int(d.sum())} elements in {int(d.any(dim=1).sum())} rows differ, max |d| "
                              f"{(o.float() - ref.float()).abs().max().item():.4g}", flush=True)
        res[name] = bad
    q.put((rank, res))


if __name__ == "__main__":
    args = {"procs": "2", "iters": "4000", "T": "1024", "burst": "100"}
    args.update(a.split("=") for a in sys.argv[1:])
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=worker, args=(r, args, q)) for r in range(int(args["procs"]))]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
    while not q.empty():
        print(q.get())
