#!/bin/bash
# round 3, GPU session 29: launch shapes of the Krylov vector sweeps (tools/vec_probe.hip)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s29; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/vec_probe.hip -o /tmp/vec_probe || exit 1
timeout 600 /tmp/vec_probe 30 | tee $OUT/vec_probe_30.txt
timeout 600 /tmp/vec_probe 26 | tee $OUT/vec_probe_26.txt
