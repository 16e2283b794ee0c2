#!/bin/bash
# Round 5 evidence run on one GPU box (from the repo root): rocprofv3 kernel stats of the max_pixels leg, of the config #3 size
# mixture, of decode steps at 2048 rows, and of the default bench command; PMC traffic passes on the gate/up GEMM.
# Only the *_kernel_stats.csv summaries are kept (gpurun brings back at most 64 MiB; a kernel trace of the bench is larger).
set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
for leg in "max_pixels 256" "config3 1024" "decode 2048"; do
  tag=$(echo $leg | cut -d' ' -f1)
  rm -rf /tmp/prof_$tag
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -- python3 $ROOT/tools/profile_leg.py $leg) > gpurun_out/prof_$tag.log 2>&1
  f=$(find /tmp/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp $f gpurun_out/r05_kernel_stats_leg_$tag.csv
  grep "^{" gpurun_out/prof_$tag.log | tail -1 > gpurun_out/r05_leg_${tag}_profiled.json
  rm -rf /tmp/prof_$tag
done
rm -rf /tmp/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big-legs --no-pil-leg) > gpurun_out/bench_r05_profiled.json 2> gpurun_out/prof_bench.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r05_kernel_stats_bench_default.csv
rm -rf /tmp/prof_bench
bash tools/pmc_gemm.sh r05 pp > gpurun_out/pmc_r05.log 2>&1
cp profiles/r05_gateup* gpurun_out/ 2>/dev/null
rm -rf gpurun_out/pmc_r05_pp_*
tail -5 gpurun_out/pmc_r05.log
du -sh gpurun_out
