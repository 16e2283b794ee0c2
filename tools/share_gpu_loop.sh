#!/bin/bash
# the 2-ranks-on-one-GPU bench (tests/test_bench_gpu.py) N times: how often does the batch-invariance check fail under contention?
n=$1; shift
for i in $(seq 1 $n); do
  OWC_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 python bench.py --gpus 2 --model 2b --batch 64 --steps 2 --warmup 1 --scorer-labels 4096 --no-cpu-baseline --no-pil-leg "$@" 2>&1 | grep -c "invariance check failed"
done | sort | uniq -c
