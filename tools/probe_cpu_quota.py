#!/usr/bin/env python3
"""How many cores does this host actually GIVE us?  `os.cpu_count()` / the affinity mask say 256 on the 1-GPU boxes, but every
multi-process host measurement there (tools/soak_host_ranks.py, tools/probe_host_prep_scaling.py) stops scaling at ~16 busy workers.
Prints the cgroup CPU quota (v2 `cpu.max`, v1 `cpu.cfs_quota_us`) and the aggregate rate of N spinning processes (pure integer
loops: no memory, no kernel) for N = 1 .. 128, plus `nr_throttled` before / after."""
from __future__ import annotations

import multiprocessing as mp
import os
import time
from pathlib import Path


def spin(seconds: float) -> int:
    n, t_end = 0, time.perf_counter() + seconds
    while time.perf_counter() < t_end:
        for _ in range(20000):
            n += 1
    return n


def read(p: str) -> str:
    try:
        return Path(p).read_text().strip().replace("\n", " | ")
    except OSError as e:
        return f"({type(e).__name__})"


def main() -> None:
    print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu.stat", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpu/cpu.stat", "/proc/self/cgroup", "/sys/fs/cgroup/cpuset.cpus.effective"):
        print(f"{p}: {read(p)}")
    base = None
    for n in (1, 4, 8, 16, 24, 32, 64, 128):
        with mp.Pool(n) as pool:
            pool.map(spin, [0.05] * n)
            t0 = time.perf_counter()
            total = sum(pool.map(spin, [1.0] * n))
            dt = time.perf_counter() - t0
        base = base or total / dt
        print(f"{n:4d} spinning processes: {total / dt / base:7.2f} x one process   ({dt:.2f} s)", flush=True)
    print("/sys/fs/cgroup/cpu.stat:", read("/sys/fs/cgroup/cpu.stat"))


if __name__ == "__main__":
    main()
