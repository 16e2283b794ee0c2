#!/bin/bash
# rocprofv3 PMC passes on the attention kernels of tools/bench_attn.py (one pass per counter group; per-launch averages per kernel):
#   bash tools/pmc_attn.sh r03
set -u
TAG=${1:-r03}
ROOT=$(pwd)
export TMPDIR=/tmp
declare -A CGROUPS=(
  [sq]="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES"
  [sq2]="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU"
  [grbm]="GRBM_GUI_ACTIVE"
  [tcc]="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"
)
for G in sq sq2 grbm tcc; do
  OUT=$ROOT/gpurun_out/pmc_${TAG}_attn_${G}
  rm -rf $OUT
  (cd /tmp && rocprofv3 --pmc ${CGROUPS[$G]} --kernel-trace --output-format csv -d $OUT -- python3 $ROOT/tools/bench_attn.py) > $OUT.log 2>&1
done
python3 - <<PY
import csv, collections, glob
acc, cnt = collections.defaultdict(float), collections.Counter()
for f in glob.glob("$ROOT/gpurun_out/pmc_${TAG}_attn_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd_kernel" in r["Kernel_Name"]:
            k = (r["Kernel_Name"].split("(")[0][-34:], r["Grid_Size"], r["Counter_Name"])
            acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in sorted(acc):
    print(f"{k[0]:36s} grid {k[1]:>9s} {k[2]:28s} {acc[k] / cnt[k]:16.1f}  ({cnt[k]} launches)")
PY
