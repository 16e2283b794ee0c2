#!/usr/bin/env python3
"""rocprofv3 --pmc passes -> committed evidence for `roofline.traffic` (bench.py reads the JSON; never a literal).

    python tools/pmc_traffic_summary.py --kernel gemm_bf16_nt_256_kernel --tag r02_gateup_swiglu \
        --shape "M 65536 N 37888 K 3584, SwiGLU epilogue" --alg-bytes A=469762048,W=271581184,C=2483027968 \
        gpurun_out/pmc_fetch gpurun_out/pmc_write [gpurun_out/pmc_tcc gpurun_out/pmc_sq ...]

Every argument directory is one rocprofv3 output tree (one pass: FETCH_SIZE costs 3 TCC slots, WRITE_SIZE 2, so they cannot
share a pass - MI355X_MICROARCH.md "rocprofv3 PMC slots").  For each pass the rows of the named kernel are copied verbatim
into profiles/<tag>_pmc_raw.csv (the raw counters, reproducible evidence) and averaged per launch into
profiles/<tag>_pmc_gemm_traffic.json:
  bytes_per_launch = 2 x FETCH_SIZE (gfx950 tallies the 128-byte requests of wide streaming reads at 64 bytes, same guide)
                     + WRITE_SIZE, both KiB -> bytes.
"""
import argparse
import collections
import csv
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True, help="substring of Kernel_Name")
    ap.add_argument("--tag", required=True)
    ap.add_argument("--shape", default="")
    ap.add_argument("--alg-bytes", default="", help="A=..,W=..,C=.. algorithmic bytes of one launch")
    ap.add_argument("--command", default="")
    ap.add_argument("--seq", type=int, default=None, help="measurement sequence number (bench.py reads the highest one of a kernel); "
                                                          "default: one more than the highest under profiles/")
    ap.add_argument("--grid", type=int, default=None,
                    help="only dispatches with this Grid_Size (threads): picks ONE shape out of a whole-program trace - the 7B gate/up "
                         "launch of bench.py's timed region is gemm_bf16_nt_256pp_kernel<4> with 256 x 148 blocks x 512 threads = 19398656")
    ap.add_argument("passes", nargs="+")
    a = ap.parse_args()
    acc, cnt = collections.defaultdict(float), collections.Counter()
    raw_rows, header = [], None
    for d in a.passes:
        for f in sorted(Path(d).rglob("*counter_collection.csv")):
            with open(f) as fh:
                rd = csv.DictReader(fh)
                header = header or rd.fieldnames
                for r in rd:
                    if a.kernel in r["Kernel_Name"] and (a.grid is None or int(r["Grid_Size"]) == a.grid):
                        acc[r["Counter_Name"]] += float(r["Counter_Value"])
                        cnt[r["Counter_Name"]] += 1
                        raw_rows.append([r.get(k, "") for k in header])
    if not cnt:
        raise SystemExit("no rows of that kernel found")
    per = {c: acc[c] / cnt[c] for c in acc}
    raw = ROOT / "profiles" / f"{a.tag}_pmc_raw.csv"
    with open(raw, "w", newline="") as fh:
        wr = csv.writer(fh)
        wr.writerow(header)
        wr.writerows(raw_rows)
    seq = a.seq
    if seq is None:
        seq = 1 + max([json.loads(f.read_text()).get("seq", -1) for f in (ROOT / "profiles").glob("r*_pmc_gemm_traffic*.json")] + [-1])
    out = {"seq": seq, "kernel": a.kernel, "shape": a.shape, "command": a.command, "launches_per_counter": dict(cnt),
           "per_launch": per, "raw_csv": str(raw.relative_to(ROOT))}
    if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
        fetch, write = 2.0 * per["FETCH_SIZE"] * 1024.0, per["WRITE_SIZE"] * 1024.0
        out.update({"fetch_bytes_per_launch_x2_gfx950_correction": fetch, "write_bytes_per_launch": write,
                    "bytes_per_launch": fetch + write})
    if a.alg_bytes:
        alg = {k: float(v) for k, v in (kv.split("=") for kv in a.alg_bytes.split(","))}
        alg["total"] = sum(alg.values())
        out["algorithmic_bytes_per_launch"] = alg
    if "TCC_HIT_sum" in per and "TCC_MISS_sum" in per:
        out["l2_hit_rate"] = per["TCC_HIT_sum"] / (per["TCC_HIT_sum"] + per["TCC_MISS_sum"])
    if "SQ_VALU_MFMA_BUSY_CYCLES" in per and "SQ_BUSY_CYCLES" in per:
        out["note_sq"] = "SQ_* are sums over all shader engines; ratios between them are meaningful, absolutes are not"
    dst = ROOT / "profiles" / f"{a.tag}_pmc_gemm_traffic.json"
    dst.write_text(json.dumps(out, indent=1))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
