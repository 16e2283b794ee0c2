#!/bin/bash
# Round 6: the measurement records behind bench.py's `roofline` object, all from ONE command line - bench.py's timed region.
# Run on the GPU box from the repo root: bash tools/profile_r6.sh [tag] [trace|pmc|all]
#   trace: rocprofv3 --kernel-trace --stats of the bench (3 timed steps) -> per-kernel stats, the trace grouped by (kernel, grid size)
#          (tools/kernel_trace_by_grid.py), and bench.py's own per-(M, N, K, epilogue) table of the timed region (HIP events).
#   pmc:   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (separate runs: MI355X_MICROARCH.md "rocprofv3 PMC slots") over ONE timed
#          step of the same bench, counters collected for the 256x256 ping-pong GEMM only (--kernel-include-regex); the rows of the
#          7B gate/up launch (gemm_bf16_nt_256pp_kernel<4>, 256 x 148 blocks) are averaged into
#          profiles/<tag>_gateup_timed_region_pmc_gemm_traffic.json - the file bench.py reads `roofline.traffic` from.
# Every step runs under its own `timeout`; the raw output trees (hundreds of MB) are removed whatever happens.
set -u
TAG=${1:-r06}
WHAT=${2:-all}
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out profiles
COMMON="--no-cpu-baseline --no-big-legs --no-pil-leg"
cleanup() { rm -rf /tmp/prof_bench /tmp/pmc_${TAG}_bench_FETCH_SIZE /tmp/pmc_${TAG}_bench_WRITE_SIZE; }
trap cleanup EXIT
if [ $WHAT = trace ] || [ $WHAT = all ]; then
  rm -rf /tmp/prof_bench
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 $COMMON \
     --shape-table $ROOT/profiles/${TAG}_gemm_by_shape_timed_region.csv) > gpurun_out/bench_${TAG}_profiled.json 2> gpurun_out/prof_bench_${TAG}.err
  echo "trace run rc=$?"
  f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f profiles/${TAG}_kernel_stats_bench.csv
  ls -la $(find /tmp/prof_bench -name "*kernel_trace.csv" | head -1)
  # gate/up at the prefill launch group (M = 65294 -> 256 tile rows): 256 x 148 blocks x 512 threads; vision fc1 (M = 131072): 512 x 20
  timeout 300 python3 tools/kernel_trace_by_grid.py /tmp/prof_bench --out profiles/${TAG}_kernel_by_grid.csv --min-total-us 2000 \
     --shape 7b.gateup+swiglu:65294:37888:3584:19398656 --shape vit.fc1:131072:5120:1280:5242880 --shape vit.qkv:131072:3840:1280:3932160 \
     --shape 7b.qkv:65294:4608:3584:2359296 | head -30
  grep "^{" gpurun_out/bench_${TAG}_profiled.json | tail -1 > profiles/bench_${TAG}_profiled.json
  rm -rf /tmp/prof_bench
fi
if [ $WHAT = pmc ] || [ $WHAT = all ]; then
  DIRS=""
  for G in FETCH_SIZE WRITE_SIZE; do
    OUT=/tmp/pmc_${TAG}_bench_${G}
    rm -rf $OUT
    (cd /tmp && timeout 600 rocprofv3 --pmc $G --kernel-trace --kernel-include-regex "gemm_bf16_nt_256pp_kernel" --output-format csv -d $OUT -- \
       python3 $ROOT/bench.py --steps 1 --warmup 0 --no-extra-legs --no-calibration $COMMON) > gpurun_out/pmc_${TAG}_${G}.log 2>&1
    echo "pmc $G rc=$?"; du -sh $OUT
    DIRS="$DIRS $OUT"
  done
  timeout 300 python3 tools/pmc_traffic_summary.py --kernel "gemm_bf16_nt_256pp_kernel<4>" --grid 19398656 --tag ${TAG}_gateup_timed_region \
    --shape "7B gate/up INSIDE bench.py's timed region: M 65294 (256 tile rows) N 37888 K 3584, SwiGLU epilogue (C = [M, N/2] bf16)" \
    --alg-bytes A=468027392,W=271581184,C=2473859072 \
    --command "rocprofv3 --pmc <FETCH_SIZE | WRITE_SIZE> --kernel-trace --kernel-include-regex gemm_bf16_nt_256pp_kernel -- python3 bench.py --steps 1 --warmup 0 --no-extra-legs --no-calibration $COMMON (tools/profile_r6.sh)" $DIRS | tail -25
  python3 - <<PY
import json, pathlib
p = pathlib.Path("profiles/${TAG}_gateup_timed_region_pmc_gemm_traffic.json")
if p.exists():   # the summary's \`kernel\` field is what bench.py matches on
    d = json.loads(p.read_text()); d["kernel_filter"] = d["kernel"]; d["kernel"] = "gemm_bf16_nt_256pp_kernel"; p.write_text(json.dumps(d, indent=1))
p = pathlib.Path("profiles/${TAG}_gateup_timed_region_pmc_raw.csv")
if p.exists():   # 2000+ launches per counter: keep 200 of each as the committed raw evidence
    rows = p.read_text().splitlines()
    head, body = rows[0], rows[1:]
    f = [r for r in body if ",FETCH_SIZE," in r][:200]
    w = [r for r in body if ",WRITE_SIZE," in r][:200]
    p.write_text("\n".join([head] + f + w) + "\n")
PY
fi
du -sh gpurun_out
