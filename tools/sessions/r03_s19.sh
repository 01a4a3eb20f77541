#!/bin/bash
# round 3, GPU session 19: copy probe 2 with interleaved pieces (is the tile-shape penalty an interleave-granularity effect?)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s19; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/copy_probe2.hip -o /tmp/copy_probe2 || exit 1
timeout 600 /tmp/copy_probe2 30 | tee $OUT/copy_probe2.txt
