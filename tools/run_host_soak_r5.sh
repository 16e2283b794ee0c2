#!/bin/bash
# Round 5: what raises the 8-rank HOST CEILING at Food-101 image sizes (emulated GPU rate 1000 images/s per rank: the host is the
# bound, so `aggregate_images_per_s` is what the 8 ranks can PREPARE) - NUMA pinning (every rank on its share of one socket),
# Pillow's block cache, a shorter GIL switch interval, worker counts - and the production rate (205 per rank) with the winners.
# usage: bash tools/run_host_soak_r5.sh > profiles/r05_host_soak.txt
cd "$(dirname "$0")/.."
echo "# host: $(nproc) cores, $(ls -d /sys/devices/system/node/node[0-9]* | wc -l) NUMA nodes; $(date -u +%Y-%m-%dT%H:%MZ)"
for n in /sys/devices/system/node/node[0-9]*; do echo "#   $(basename $n): $(cat $n/cpulist)"; done
C="--ranks 8 --images 4096 --gpu-rate 1000 --sizes food101"
for cfg in "--threads 8" "--threads 8 --pin" "--threads 8 --pillow-blocks 256" "--threads 8 --switch-interval-ms 0.5" \
           "--threads 8 --pin --pillow-blocks 256 --switch-interval-ms 0.5" "--threads 12 --pin --pillow-blocks 256 --switch-interval-ms 0.5" \
           "--threads 16 --pin --pillow-blocks 256 --switch-interval-ms 0.5"; do
  echo "## soak_host_ranks.py $C $cfg"
  python tools/soak_host_ranks.py $C $cfg 2>/dev/null | tail -1
done
for cfg in "--gpu-rate 205 --sizes food101 --threads 8 --images 3156" "--gpu-rate 205 --sizes food101 --threads 8 --images 3156 --pin --pillow-blocks 256 --switch-interval-ms 0.5" \
           "--gpu-rate 240 --threads 8 --images 3156 --pin --pillow-blocks 256 --switch-interval-ms 0.5"; do
  echo "## soak_host_ranks.py --ranks 8 $cfg"
  python tools/soak_host_ranks.py --ranks 8 $cfg 2>/dev/null | tail -1
done
