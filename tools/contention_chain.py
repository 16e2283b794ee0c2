"""Chains of DEPENDENT ops under contention (P processes share cuda:0): which dependency pattern loses determinism?
usage: python tools/contention_chain.py [procs=2] [iters=200] [layers=32] [T=1024] [variants=a,b,...]"""
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
    NL = int(args["layers"])
    g = torch.Generator(device=dev).manual_seed(7 + rank)
    rn = lambda *s, sc=1.0: (torch.randn(*s, device=dev, generator=g) * sc).to(torch.bfloat16)  # noqa: E731
    x0 = rn(T, E)
    w_qkv, b_qkv = rn(3 * E, E, sc=0.03), rn(3 * E)
    w_proj, b_proj = rn(E, E, sc=0.01), rn(E, sc=0.1)
    w_fc1, b_fc1, w_fc2, b_fc2 = rn(F, E, sc=0.03), rn(F), rn(E, F, sc=0.005), rn(E, sc=0.1)
    ln_w, ln_b = rn(E), rn(E)
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=dev)  # noqa: E731
    starts, lens = i32([0]), i32([T])
    h, attn = torch.empty_like(x0), torch.empty_like(x0)
    qkv = torch.empty(T, 3 * E, dtype=torch.bfloat16, device=dev)
    mlp = torch.empty(T, F, dtype=torch.bfloat16, device=dev)

    def att():
        ops.attention(qkv, 3 * E, hd, qkv[:, E:], 3 * E, hd, qkv[:, 2 * E:], 3 * E, hd, attn, E, hd, starts, starts, lens,
                      n_seq=1, n_heads=H, kv_group=1, head_dim=hd, max_q_len=T, causal=False, scale=hd ** -0.5)

    def full(inplace=True, with_attn=True, with_ln=True):
        x = x0.clone()
        y = torch.empty_like(x)
        for _ in range(NL):
            if with_ln:
                ops.layernorm(x, ln_w, ln_b, 1e-6, out=h)
            else:
                h.copy_(x)
            ops.gemm_bf16(h, w_qkv, b_qkv, out=qkv)
            if with_attn:
                att()
                a = attn
            else:
                a = qkv[:, :E]
            if inplace:
                ops.gemm_bf16(a if a.is_contiguous() else a.contiguous(), w_proj, b_proj, epilogue=_lib.EPI_RESIDUAL, residual=x, out=x)
            else:
                ops.gemm_bf16(a if a.is_contiguous() else a.contiguous(), w_proj, b_proj, epilogue=_lib.EPI_RESIDUAL, residual=x, out=y)
                x, y = y, x
            ops.layernorm(x, ln_w, ln_b, 1e-6, out=h)
            ops.gemm_bf16(h, w_fc1, b_fc1, epilogue=_lib.EPI_QUICK_GELU, out=mlp)
            if inplace:
                ops.gemm_bf16(mlp, w_fc2, b_fc2, epilogue=_lib.EPI_RESIDUAL, residual=x, out=x)
            else:
                ops.gemm_bf16(mlp, w_fc2, b_fc2, epilogue=_lib.EPI_RESIDUAL, residual=x, out=y)
                x, y = y, x
        return x

    def ln_only():     # x -> LN -> h -> LN -> x ... (elementwise-row kernels only)
        x = x0.clone()
        for _ in range(NL * 4):
            ops.layernorm(x, ln_w, ln_b, 1e-6, out=h)
            ops.layernorm(h, ln_w, ln_b, 1e-6, out=x)
        return x

    def gemm_chain():  # h = gemm(x) ; x = gemm(h) out of place, no residual
        x = x0.clone()
        for _ in range(NL * 2):
            ops.gemm_bf16(x, w_proj, b_proj, out=h)
            ops.gemm_bf16(h, w_proj, b_proj, out=x)
        return x

    def torch_chain():  # the same dependency pattern with torch's own kernels (is it this library at all?)
        x = x0.clone().float()
        wp = w_proj.float()
        for _ in range(NL * 2):
            hh = torch.nn.functional.layer_norm(x, (E,))
            x = x + hh @ wp.t() * 0.1
        return x

    table = {"full in place": lambda: full(True), "full out of place": lambda: full(False), "no attention": lambda: full(True, False),
             "no LN1": lambda: full(True, True, False), "layernorm chain": ln_only, "gemm chain": gemm_chain, "torch chain": torch_chain}
    names = args["variants"].split(",") if args["variants"] else list(table)
    res = {}
    for name in names:
        fn = table[name]
        ref = fn().clone()
        torch.cuda.synchronize()
        bad = 0
        for it in range(int(args["iters"])):
            o = fn()
            torch.cuda.synchronize()
            if not torch.equal(o, ref):
                bad += 1
                if bad == 1:
                    d = (o != ref)
                    print(f"rank {rank} {name}: iter {it}: {int(d.sum())} elements in {int(d.any(dim=1).sum())} rows differ", flush=True)
        res[name] = bad
    q.put((rank, res))


if __name__ == "__main__":
    args = {"procs": "2", "iters": "200", "T": "1024", "layers": "32", "variants": ""}
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
