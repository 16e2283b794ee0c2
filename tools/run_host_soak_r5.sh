#!/bin/bash
# Round 5: what raises the 8-rank HOST CEILING at Food-101 image sizes (emulated GPU rate 1000 images/s per rank: the host is the
# bound, so `aggregate_images_per_s` is what the 8 ranks can PREPARE): NUMA pinning (every rank on its share of one socket:
# `--pin` emulates what the plug-in does from its GPU's local_cpulist), a 0.5 ms GIL switch interval, Pillow's block cache, two
# units in preparation at once, worker counts - then the production rates (205 / 240 per rank) with the plug-in's defaults.
# usage: bash tools/run_host_soak_r5.sh > profiles/r05_host_soak.txt
cd "$(dirname "$0")/.."
echo "# host: $(nproc) cores, $(ls -d /sys/devices/system/node/node[0-9]* | wc -l) NUMA nodes; $(date -u +%Y-%m-%dT%H:%MZ)"
for n in /sys/devices/system/node/node[0-9]*; do echo "#   $(basename $n): $(cat $n/cpulist)"; done
C="--ranks 8 --images 4096 --gpu-rate 1000 --sizes food101"
OFF="OWC_PREP_UNITS=1 OWC_GIL_SWITCH_MS=0 OWC_PILLOW_BLOCKS=0"
run() { echo "## $1 soak_host_ranks.py $2"; env $1 python tools/soak_host_ranks.py $2 2>/dev/null | tail -1; }
run "$OFF" "$C --threads 8"                                                   # round 4's pipeline
run "$OFF" "$C --threads 8 --pin"
run "OWC_PREP_UNITS=1 OWC_GIL_SWITCH_MS=0.5 OWC_PILLOW_BLOCKS=0" "$C --threads 8"
run "OWC_PREP_UNITS=1 OWC_GIL_SWITCH_MS=0 OWC_PILLOW_BLOCKS=256" "$C --threads 8"
run "OWC_PREP_UNITS=2 OWC_GIL_SWITCH_MS=0 OWC_PILLOW_BLOCKS=0" "$C --threads 8"
run "OWC_PREP_UNITS=2" "$C --threads 8 --pin"                                 # the plug-in's defaults + pinning
run "OWC_PREP_UNITS=3" "$C --threads 8 --pin"
run "OWC_PREP_UNITS=2" "$C --threads 12 --pin"
run "OWC_PREP_UNITS=2" "--ranks 8 --gpu-rate 205 --sizes food101 --threads 8 --images 3156 --pin"
run "OWC_PREP_UNITS=2" "--ranks 8 --gpu-rate 240 --threads 8 --images 3156 --pin"
