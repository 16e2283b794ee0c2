#!/bin/bash
# Per-kernel time of one decode step of a model at several batch sizes (rocprofv3 --kernel-trace, one process per batch size):
#   bash tools/profile_decode.sh r03 qwen2-vl-7b "1 8 32 128" [bf16|fp8] [knob=value ...]
# writes gpurun_out/<tag>_decode_kernels_b<B>.txt (tools/decode_kernels.py on the trace database)
set -u
TAG=${1:-r03}; MODEL=${2:-qwen2-vl-7b}; BATCHES=${3:-"1 8 32 128"}; DT=${4:-bf16}; shift 4 2>/dev/null
ROOT=$(pwd)
export TMPDIR=/tmp
for B in $BATCHES; do
  OUT=$ROOT/gpurun_out/prof_${TAG}_dec_b$B
  rm -rf $OUT
  (cd /tmp && rocprofv3 --kernel-trace -d $OUT -- python3 $ROOT/tools/bench_decode_latency.py $MODEL $B $DT "$@") > $OUT.log 2>&1
  DB=$(find $OUT -name "*.db" | head -1)
  python3 tools/decode_kernels.py $DB > gpurun_out/${TAG}_decode_kernels_b$B.txt 2>&1
  grep "ms/token-step" $OUT.log >> gpurun_out/${TAG}_decode_latency_profiled.txt
  rm -rf $OUT
done
