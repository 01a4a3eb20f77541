#!/bin/bash
# round 3, GPU session 11: window pass with 128-byte runs (31.5 KB tiles, four 512-thread workgroups per CU)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s11; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto || exit 1
{
timeout 120 /tmp/sc3_proto 27 13 14 10 1 1 0 1 2 1024 512 1 1 2 3 3 | tail -2
for rep in 1 2; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 1024 1 1 4 3 4 | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 512 1 1 4 3 3 | tail -1
done
timeout 300 /tmp/sc3_proto 36 18 14 10 1 1 0 1 3 1024 1024 1 1 4 3 4 | tail -1
} 2>&1 | tee $OUT/sc3_runs128.txt
for G in "FETCH_SIZE" "WRITE_SIZE"; do timeout 600 tools/pmc_kernels.sh sc3_win "$G" -- /tmp/sc3_proto 32 16 14 10 1 1 0 1 3 1024 512 1 1 4 3 3; done 2>&1 | tee -a $OUT/sc3_runs128.txt
