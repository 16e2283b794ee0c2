"""Per-kernel time of ONE decode step from a rocprofv3 --kernel-trace database of tools/bench_decode_latency.py <model> <B>.
usage: python tools/decode_kernels.py <results.db> [n_layers=28]   (the last 34-token generate's decode launches)"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
L = int(sys.argv[2]) if len(sys.argv) > 2 else 28
rows = c.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
# the decode steps of the last generate: everything after the last causal-prefill attention launch
last_prefill = max(i for i, r in enumerate(rows) if "attn_fwd_kernel" in r[0] and "true" in r[0].split("attn_fwd_kernel")[1][:20])
tail = rows[last_prefill + 1:]
agg = {}
for name, s, e, gx, wx in tail:
    m = re.search(r"(\w+_kernel)(<[^>]*>)?", name)
    key = (m.group(0) if m else name[:40], gx // max(wx, 1))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
n_steps = sum(v[0] for k, v in agg.items() if "argmax" in k[0])
tot = sum(v[1] for v in agg.values())
print(f"{n_steps} decode steps, {tot / n_steps / 1e3:.3f} ms of kernel time per step; wall {(tail[-1][2] - tail[0][1]) / 1e6 / n_steps:.3f} ms per step")
for (k, blocks), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:44s} blocks {blocks:6d}  x{n / n_steps:6.1f}/step  avg {us / n:8.1f} us  {100 * us / tot:5.1f} %")
