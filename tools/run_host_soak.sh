#!/bin/bash
# Host-side readiness of an 8-rank node, measured on the box's own cores without 8 GPUs:
#   tools/soak_host_ranks.py  - the real generate_until host pipeline of 8 ranks against an emulated GPU rate
#   tools/time_rank_tail.py   - what rank 0 does after the GPUs are done with a task
# usage: bash tools/run_host_soak.sh > profiles/r04_host_soak.txt
cd "$(dirname "$0")/.."
echo "# host: $(nproc) cores; $(date -u +%Y-%m-%dT%H:%MZ)"
for cfg in "--gpu-rate 240 --threads 8" "--gpu-rate 240 --threads 4" "--gpu-rate 240 --threads 8 --images 3156" \
           "--gpu-rate 205 --sizes food101 --threads 8" "--gpu-rate 205 --sizes food101 --threads 16" "--gpu-rate 205 --sizes food101 --threads 8 --images 3156" \
           "--gpu-rate 1000 --threads 8" "--gpu-rate 1000 --sizes food101 --threads 8" "--gpu-rate 1000 --sizes food101 --threads 16"; do
  echo "## soak_host_ranks.py --ranks 8 --images 6144 $cfg"
  python tools/soak_host_ranks.py --ranks 8 --images 6144 $cfg 2>/dev/null | tail -1
done
for n in 2048 6250; do
  echo "## time_rank_tail.py --ranks 8 --docs-per-rank $n"
  python tools/time_rank_tail.py --ranks 8 --docs-per-rank $n 2>/dev/null | tail -1
done
