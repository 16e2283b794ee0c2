#!/usr/bin/env python3
"""bf16 nn.Linear on the host (the CPU baseline's dominant op: 1089 prompt rows x the 7B MLP) with N intra-op threads - on a box whose
cgroup gives fewer CPUs than `os.cpu_count()` says (tools/probe_cpu_quota.py: 16 of 256), how many threads should the baseline use?"""
import sys
import time

import torch

x = torch.randn(1089, 3584).bfloat16()
w = torch.randn(18944, 3584).bfloat16()
for n in [int(a) for a in sys.argv[1:]] or [128, 64, 32, 16, 8]:
    torch.set_num_threads(n)
    with torch.no_grad():
        torch.nn.functional.linear(x, w)
        t0 = time.perf_counter()
        for _ in range(5):
            torch.nn.functional.linear(x, w)
        dt = (time.perf_counter() - t0) / 5
    print(f"{n:4d} threads: {dt * 1e3:8.1f} ms  {2 * 1089 * 3584 * 18944 / dt / 1e12:6.2f} TFLOP/s", flush=True)
