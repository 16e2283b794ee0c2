#!/bin/bash
# rocprofv3 PMC passes on ONE decode projection at a mid batch (default: 7B down projection, M = 128: the 64x64 ring kernel):
#   bash tools/pmc_decode_gemm.sh r03 down 128
# one pass per counter group; prints per-launch averages of the named kernel from the counter CSVs.
set -u
TAG=${1:-r03}; WHICH=${2:-down}; M=${3:-128}
ROOT=$(pwd)
export TMPDIR=/tmp
declare -A CGROUPS=(
  [fetch]="FETCH_SIZE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
  [sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVES"
  [sq2]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  [grbm]="GRBM_GUI_ACTIVE"
)
for G in fetch tcc sq sq2 grbm; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_dec_${WHICH}_${G}
  rm -rf $OUT
  (cd /tmp && rocprofv3 --pmc ${CGROUPS[$G]} --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_decode_gemms.py --only=$WHICH --m=$M) > $OUT.log 2>&1
done
python3 - <<PY
import csv, collections, glob
acc, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/pmc_${TAG}_dec_${WHICH}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gemm_bf16" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print(f"{k[0]:60s} {k[1]:28s} {acc[k] / cnt[k]:16.1f}  ({cnt[k]} launches)")
PY
