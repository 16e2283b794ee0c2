#!/bin/bash
# rocprofv3 kernel stats of the default bench command on the round's final tree (from the repo root; only the summary is kept)
set -u
ROOT=$(pwd); export TMPDIR=/tmp; mkdir -p gpurun_out
rm -rf /tmp/prof_bench
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_bench -- python3 $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-big-legs --no-pil-leg) > gpurun_out/bench_r05_profiled_final_tree.json 2> gpurun_out/prof_bench_final.err
f=$(find /tmp/prof_bench -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f gpurun_out/r05_kernel_stats_bench_final_tree.csv
rm -rf /tmp/prof_bench
head -12 gpurun_out/r05_kernel_stats_bench_final_tree.csv
grep "^{" gpurun_out/bench_r05_profiled_final_tree.json | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline'].get('launches'), d['roofline'].get('kernel_ms_total'))
"
du -sh gpurun_out
