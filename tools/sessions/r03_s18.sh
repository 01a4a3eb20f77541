#!/bin/bash
# round 3, GPU session 18: Infinity-Cache blocking probe (tools/mall_probe.hip)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s18; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe || exit 1
timeout 600 /tmp/mall_probe 30 | tee $OUT/mall_probe.txt
