#!/bin/bash
# round 3, GPU session 3: SC3 prototype with the T bonds split between the passes and small dispatch groups; counters;
# workgroup-shape copy probe
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s3; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto || exit 1
hipcc --offload-arch=gfx950 -O3 tools/copy_probe2.hip -o /tmp/copy_probe2 || exit 1
{
echo "== correctness (split T bonds, small groups)"
timeout 120 /tmp/sc3_proto 27 13 14 10 3 3 2 1 2 1024 1024 1 1 2 | tail -2
timeout 120 /tmp/sc3_proto 28 14 14 10 3 3 2 0 2 1024 1024 1 1 1 | tail -2
echo "== timings L=32 k=16"
#          oA oB tInA accA reps ntA ntB nbA nbB t1
for cfg in "3 3 2 1 5 1024 1024 1 1 4" "3 3 2 1 5 1024 1024 1 1 3" "3 3 2 1 5 1024 1024 1 1 5" "3 3 2 1 5 1024 1024 1 1 2" \
           "3 3 2 0 5 1024 1024 1 1 4" "3 3 2 1 5 1024 1024 2 2 4" "3 3 2 1 5 512 512 2 2 4" "3 1 0 1 5 1024 1024 1 1 4" \
           "1 3 2 1 5 1024 1024 1 1 4" "3 3 2 1 5 1024 1024 1 1 6" "3 3 2 1 5 1024 1024 1 1 8"; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 $cfg | tail -1
done
} 2>&1 | tee $OUT/sc3_proto.txt
{
for cfg in "1 1 0 1 3 1024 1024 1 1 4" "3 3 2 1 3 1024 1024 1 1 4"; do
  echo "== PMC: $cfg"
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    timeout 600 tools/pmc_kernels.sh sc3_ "$G" -- /tmp/sc3_proto 32 16 14 10 $cfg
  done
done
} 2>&1 | tee $OUT/sc3_pmc.txt
timeout 300 /tmp/copy_probe2 30 2>&1 | tee $OUT/copy_probe2.txt
