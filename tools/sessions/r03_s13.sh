#!/bin/bash
# round 3, GPU session 13: window pass with its gathers pipelined across the LDS phase
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s13; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto || exit 1
{
timeout 120 /tmp/sc3_proto 27 13 14 10 1 1 0 1 2 1024 1024 1 1 2 19 | tail -2
for rep in 1 2; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 1024 1 1 4 3 | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 1024 1 1 4 19 | tail -1
done
} 2>&1 | tee $OUT/sc3_pipe.txt
