#!/bin/bash
# rocprofv3 PMC passes on the step's largest GEMM launch class (7B gate/up at the prefill launch-group size, SwiGLU epilogue:
# what owc_llm_prefill really launches), one counter group per pass (MI355X_MICROARCH.md "rocprofv3 PMC slots"), for the
# ping-pong kernel (default) and the lock-step kernel (--set=gemm_pingpong=0).  Run on the GPU box from the repo root:
#   bash tools/pmc_gemm.sh r02 ["pp" | "ls" | "pp ls"]
# writes gpurun_out/pmc_<tag>_<kernel>_<group>/ and profiles/<tag>_<kernel>_pmc_{raw.csv,gemm_traffic.json}
set -u
TAG=${1:-r02}
ROOT=$(pwd)
export TMPDIR=/tmp
SHAPE="7b.gateup --m=65536 --epi=swiglu --iters=4"
declare -A CGROUPS=(
  [fetch]="FETCH_SIZE"
  [write]="WRITE_SIZE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  [sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES"
  [grbm]="GRBM_GUI_ACTIVE"
)
for K in ${2:-pp ls}; do
  SET=""; KN="gemm_bf16_nt_256pp_kernel"
  if [ $K = ls ]; then SET="--set=gemm_pingpong=0"; KN="gemm_bf16_nt_256_kernel"; fi
  DIRS=""
  for G in fetch write tcc sq grbm; do
    OUT=$ROOT/gpurun_out/pmc_${TAG}_${K}_${G}
    rm -rf $OUT
    (cd /tmp && rocprofv3 --pmc ${CGROUPS[$G]} --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_gemm.py $SHAPE $SET) > $OUT.log 2>&1
    DIRS="$DIRS $OUT"
  done
  python3 tools/pmc_traffic_summary.py --kernel $KN --tag ${TAG}_gateup_swiglu_${K} \
    --shape "7B gate/up at the prefill launch group: M 65536 N 37888 K 3584, SwiGLU epilogue (C = [M, N/2] bf16)" \
    --alg-bytes A=469762048,W=271581184,C=2483027968 \
    --command "rocprofv3 --pmc <group> --kernel-trace -- python3 tools/bench_gemm.py $SHAPE $SET (one pass per counter group; tools/pmc_gemm.sh)" $DIRS | tail -30
done
