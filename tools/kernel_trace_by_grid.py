#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> one row per (kernel name, grid size): calls, total / avg / min / max microseconds.

rocprofv3's own `--stats` summary aggregates by kernel NAME, i.e. over every shape a template instantiation is launched with
(`gemm_bf16_nt_256pp_kernel<4>`: avg 3.59 ms, min 0.076, max 40.6 in round 5's record) - a roofline fraction cannot be recomputed
from that.  Grouping by grid size as well separates the shapes (a 256x256-tile GEMM's grid is tiles_m x tiles_n x 512 threads);
two shapes that share kernel AND grid (the 7B o / down projections: both 256 x 14 tiles of the residual epilogue, K 3584 / 18944)
stay merged here - bench.py's `roofline.by_shape` (HIP events per launch, keyed by M, N, K, epilogue) separates those.

    python tools/kernel_trace_by_grid.py <rocprofv3 output dir> [--out profiles/r06_kernel_by_grid.csv] [--min-total-us 1000]
           [--shape NAME:M:N:K:GRID ...]   adds a 2MNK / avg column for the rows whose grid matches
"""
import argparse
import collections
import csv
import re
from pathlib import Path


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("trace_dir")
    ap.add_argument("--out", default=None)
    ap.add_argument("--min-total-us", type=float, default=0.0)
    ap.add_argument("--shape", action="append", default=[], help="NAME:M:N:K:GRID - annotate the row(s) of that grid with TFLOP/s")
    a = ap.parse_args()
    shapes = {}
    for sp in a.shape:
        nm, m, n, k, g = sp.split(":")
        shapes[int(g)] = (nm, 2.0 * int(m) * int(n) * int(k))
    acc = collections.defaultdict(list)
    for f in sorted(Path(a.trace_dir).rglob("*kernel_trace.csv")):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if "Grid_Size" in r:      # (counter-collection CSVs)
                    grid, wg = int(r["Grid_Size"]), int(r["Workgroup_Size"])
                else:                     # kernel-trace CSVs give the three dimensions
                    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
                    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
                acc[(short(r["Kernel_Name"]), grid, wg)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    rows = []
    for (name, grid, wg), ts in acc.items():
        tot = sum(ts)
        if tot < a.min_total_us:
            continue
        note, tf = "", ""
        if grid in shapes and "gemm" in name:
            note, flop = shapes[grid]
            tf = f"{flop / (tot / len(ts)) / 1e6:.1f}"
        rows.append((tot, [name, grid, wg, len(ts), f"{tot:.1f}", f"{tot / len(ts):.2f}", f"{min(ts):.2f}", f"{max(ts):.2f}", note, tf]))
    rows.sort(key=lambda x: -x[0])
    header = ["kernel", "grid_threads", "workgroup", "calls", "total_us", "avg_us", "min_us", "max_us", "shape", "tflops_from_avg"]
    out = [header] + [r for _, r in rows]
    if a.out:
        with open(a.out, "w", newline="") as fh:
            csv.writer(fh).writerows(out)
    for r in out[:40]:
        print(",".join(str(x) for x in r))


if __name__ == "__main__":
    main()
