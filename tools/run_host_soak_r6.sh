#!/bin/bash
# Round 6: the HOST side of N ranks at Food-101 image sizes (emulated GPU rate 4000 images/s per rank: the host is the bound, so
# `aggregate_images_per_s` is what the ranks can PREPARE; one GPU consumes 205 images/s at these sizes).
# What changed: (1) the 1-GPU boxes grant 16 CPUs (cgroup cpu.max) of the 256 they show - tools/probe_cpu_quota.py - so "8 ranks x 8
# PIL workers" of rounds 4-5 were 64+ runnable threads on 16 CPUs, and the "8-rank ceiling" of ~3100 images/s was the QUOTA
# (16 CPUs x ~200 images/s); the plug-in now sizes its pool from the quota (`usable_cpus`).  (2) the JPEG encoder writes to a memfd and
# runs outside the GIL (OWC_JPEG_BYTESIO=1: round 5's BytesIO).  (3) glibc + Pillow keep image-sized blocks mapped
# (OWC_MALLOC_KEEP=0 OWC_PILLOW_BLOCKS=0: off).  (4) one pool fan-out per unit for the staging copies.
# usage: bash tools/run_host_soak_r6.sh > profiles/r06_host_soak.txt
cd "$(dirname "$0")/.."
echo "# host: $(nproc) logical CPUs, cgroup cpu.max = $(cat /sys/fs/cgroup/cpu.max 2>/dev/null), $(ls -d /sys/devices/system/node/node[0-9]* | wc -l) NUMA nodes; $(date -u +%Y-%m-%dT%H:%MZ)"
C="--images 4096 --gpu-rate 4000 --sizes food101"
R5="OWC_JPEG_BYTESIO=1 OWC_MALLOC_KEEP=0 OWC_PILLOW_BLOCKS=0"
run() { echo "## $1 soak_host_ranks.py $2"; env $1 timeout 300 python tools/soak_host_ranks.py $2 2>/dev/null | tail -1; }
echo "# ---- one rank alone (8 workers + preparation + launch thread: within the quota)"
run "$R5" "--ranks 1 $C --threads 8"
run "OWC_JPEG_BYTESIO=0 OWC_MALLOC_KEEP=0 OWC_PILLOW_BLOCKS=0" "--ranks 1 $C --threads 8"
run "OWC_JPEG_BYTESIO=0 OWC_MALLOC_KEEP=1 OWC_PILLOW_BLOCKS=0" "--ranks 1 $C --threads 8"
run "OWC_X=0" "--ranks 1 $C"                                   # this round's defaults
run "OWC_X=0" "--ranks 1 $C --threads 12"
run "OWC_PREP_UNITS=2" "--ranks 1 $C"
echo "# ---- two ranks (8 CPUs each)"
run "$R5" "--ranks 2 $C --threads 8"
run "OWC_X=0" "--ranks 2 $C"
echo "# ---- eight ranks on 16 CPUs (two CPUs per rank: what this box can show of an 8-rank node)"
run "$R5" "--ranks 8 $C --threads 8 --pin"                     # round 5: 8 workers per rank whatever the quota
run "OWC_X=0" "--ranks 8 $C --pin"                             # defaults: the pool follows the quota (2 workers per rank)
run "OWC_X=0" "--ranks 8 $C --threads 4 --pin"
echo "# ---- production rates, defaults"
run "OWC_X=0" "--ranks 8 --gpu-rate 205 --sizes food101 --images 3156 --pin"
run "OWC_X=0" "--ranks 8 --gpu-rate 240 --images 3156 --pin"
