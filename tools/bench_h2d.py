import time, numpy as np, torch
dev = torch.device("cuda:0")
for nbytes in (512, 8192, 600_000, 2_400_000):
    a = np.random.randint(0, 1000, nbytes // 4).astype(np.int32)
    def pageable(): return torch.from_numpy(a).to(dev, non_blocking=True)
    def pinned_cache(): return torch.from_numpy(a).pin_memory().to(dev, non_blocking=True)
    buf = torch.empty(a.size, dtype=torch.int32).pin_memory()
    def pinned_persistent():
        buf.numpy()[:] = a
        return buf.to(dev, non_blocking=True)
    def blocking(): return torch.from_numpy(a).to(dev)
    for name, fn in (("pageable non_blocking", pageable), ("pin_memory() + non_blocking", pinned_cache), ("persistent pinned + non_blocking", pinned_persistent), ("pageable blocking", blocking)):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"{nbytes:8d} B  {name:34s} host {1e6 * (t1 - t0) / 200:8.1f} us/call   incl. drain {1e6 * (t2 - t0) / 200:8.1f} us/call", flush=True)
