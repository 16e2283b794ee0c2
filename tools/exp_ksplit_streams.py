"""EXPERIMENT (round 6): what would a fixed-order K split buy the decode steps' down projection?  

The 7B down projection (N = 3584, K = 18944 = 4 x 37 K-tiles of 128) at M rows of a decode step, three ways, per launch:
  one:      the shipped launch (residual epilogue) - at 512 rows 224 blocks of 128 x 64 tiles, 452 TFLOP/s
  split-1s: the four K segments as four fp32-output launches on ONE stream (no concurrency: what the split costs by itself)
  split-4s: the same four launches on FOUR streams (fork / join with events), `gemm_pp128` lowered so that each takes 256 x 128
            tiles (56 blocks each at 512 rows), + the reduce out = bf16(r + (((p0 + p1) + p2) + p3)) as torch ops (a stand-in
            for one fused kernel: its own time is printed)
usage: python tools/exp_ksplit_streams.py [rows,...] [pp128_min_tiles=32]"""
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from lmms_owc_amd import _lib, ops  # noqa: E402


def main():
    rows = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "128,256,512,768,1024,2048").split(",")]
    pp = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    dev = torch.device("cuda:0")
    lib = _lib.load()
    N, K, S = 3584, 18944, 4
    Ks = K // S
    torch.manual_seed(0)
    w = (torch.randn(N, K, device=dev) * 0.02).to(torch.bfloat16)
    main_s = torch.cuda.current_stream()
    side = [torch.cuda.Stream() for _ in range(S - 1)]
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for M in rows:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        r = torch.randn(M, N, device=dev).to(torch.bfloat16)
        parts = [torch.empty(M, N, device=dev, dtype=torch.float32) for _ in range(S)]
        fork, joins = torch.cuda.Event(), [torch.cuda.Event() for _ in range(S - 1)]

        def one():
            return ops.gemm_bf16(a, w, None, epilogue=_lib.EPI_RESIDUAL, residual=r)

        def seg(s):
            ops.gemm_bf16(a[:, s * Ks:(s + 1) * Ks], w[:, s * Ks:(s + 1) * Ks], None, epilogue=_lib.EPI_F32, out=parts[s])

        def split_1s():
            for s in range(S):
                seg(s)

        def split_4s():
            fork.record(main_s)
            for i, st in enumerate(side):
                st.wait_event(fork)
                with torch.cuda.stream(st):
                    seg(i + 1)
                    joins[i].record(st)
            seg(0)
            for j in joins:
                main_s.wait_event(j)

        def reduce():
            return (r.float() + (((parts[0] + parts[1]) + parts[2]) + parts[3])).to(torch.bfloat16)

        def timed(fn, n=28, reps=5):
            best = []
            for _ in range(reps):
                fn()
                fn()
                e0.record()
                for _ in range(n):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best.append(e0.elapsed_time(e1) / n * 1e3)
            return sorted(best)[len(best) // 2]

        flops = 2.0 * M * N * K
        t_one = timed(one)
        lib.owc_tuning_set(b"gemm_pp128", pp)
        t_1s = timed(split_1s)
        t_4s = timed(split_4s)
        t_red = timed(reduce)
        want = one()
        split_4s()
        got = reduce()
        torch.cuda.synchronize()
        err = (got.float() - want.float()).abs().max().item() / want.float().abs().max().item()
        lib.owc_tuning_set(b"gemm_pp128", -1)
        print(f"M={M:5d}  one {t_one:7.1f} us ({flops / t_one / 1e6:6.0f} TF)   split-1s {t_1s:7.1f} us   split-4s {t_4s:7.1f} us ({flops / t_4s / 1e6:6.0f} TF)"
              f"   torch reduce {t_red:6.1f} us   max |split - one| / max |one| = {err:.2e}", flush=True)


if __name__ == "__main__":
    main()
