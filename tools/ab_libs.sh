#!/bin/bash
# A/B two builds of libowc_hip.so on ONE GPU box: ab/old.so and ab/new.so (the ab/ directory travels with gpurun, gpurun_out/ does not).
# usage: tools/ab_libs.sh <rounds> <command...>   e.g. tools/ab_libs.sh 2 python tools/bench_attn.py
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for v in old new; do
    cp ab/$v.so lmms_owc_amd/libowc_hip.so || exit 1
    echo "== $v ($(md5sum < lmms_owc_amd/libowc_hip.so | cut -c1-8))"
    "$@" 2>&1 | grep -v amdgpu.ids
  done
done
