"""Vendor-library yardstick (measurement aid only, never on the product path): a few torch.matmul calls so that
`rocprofv3 --kernel-trace --stats -- python3 tools/yardstick_matmul.py` shows which hipBLASLt kernel configuration
(macro tile, wave tile, staging mode) the library picks for this path's GEMM shapes."""
import torch

dev = torch.device("cuda:0")
for m, n, k in [(8192, 8192, 8192), (18304, 37888, 3584), (18304, 3584, 18944), (32768, 3840, 1280), (2048, 37888, 3584)]:
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    w = (torch.randn(n, k, device=dev) * 0.05).to(torch.bfloat16)
    out = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    for _ in range(5):
        torch.matmul(a, w.t(), out=out)
    torch.cuda.synchronize()
    del a, w, out
