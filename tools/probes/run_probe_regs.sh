#!/bin/bash
# one instance, then two and three at once (see probe_regs_contention.hip)
hipcc -O2 --offload-arch=gfx950 tools/probes/probe_regs_contention.hip -o /tmp/probe_regs 2>/dev/null || exit 1
echo "== 1 process"; /tmp/probe_regs
echo "== 2 processes"; /tmp/probe_regs & /tmp/probe_regs & wait
echo "== 3 processes"; /tmp/probe_regs & /tmp/probe_regs & /tmp/probe_regs & wait
