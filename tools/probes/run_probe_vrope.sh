#!/bin/bash
# the fused RoPE epilogue as shipped, after MFMA bursts, several waves per SIMD (see probe_vrope_epilogue.hip)
hipcc -O3 -ffp-contract=fast -Wno-unused-value --offload-arch=gfx950 -Ilmms_owc_amd/csrc -Iinclude tools/probes/probe_vrope_epilogue.hip -o /tmp/probe_vrope 2>/dev/null || exit 1
/tmp/probe_vrope 3000 40 | tail -3
