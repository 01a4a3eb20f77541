#!/bin/bash
# round 3, GPU session 2: the three-field SpinConserve prototype -- correctness at L=26, timings at L=32
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s2; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto || exit 1
{
echo "== correctness L=26 k=13"
#                      L  k  a  w oA oB tInA accA reps ntA ntB nbA nbB
timeout 120 /tmp/sc3_proto 26 13 14 10 1 1 0 1 2 512 512 2 2
timeout 120 /tmp/sc3_proto 26 13 14 10 0 0 1 0 2 1024 1024 1 1
timeout 120 /tmp/sc3_proto 26 12 14 10 2 1 0 0 2 1024 512 2 1
timeout 120 /tmp/sc3_proto 27 13 14 10 1 0 1 1 2 512 1024 1 2
echo "== timings L=32 k=16"
for cfg in "1 1 0 1 5 512 512 2 2" "1 1 0 1 5 1024 1024 1 1" "1 1 0 1 5 512 1024 2 1" "1 1 0 1 5 1024 512 1 2" \
           "1 1 1 1 5 512 512 2 2" "1 1 1 1 5 1024 1024 1 1" "0 0 0 1 5 512 512 2 2" "2 1 0 1 5 512 512 2 2" \
           "1 1 0 0 5 512 512 2 2" "1 1 0 0 5 1024 1024 1 1" "1 0 0 1 5 512 512 2 2" "0 1 0 1 5 512 512 2 2"; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 $cfg | tail -1
done
} 2>&1 | tee $OUT/sc3_proto.txt
