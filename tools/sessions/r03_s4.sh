#!/bin/bash
# round 3, GPU session 4: SC3 prototype v2 -- gathers issued before the tile is waited for, diagonal on the fly,
# real symmetric bond coefficients
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s4; mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 tools/experiments/sc3_proto.hip -o /tmp/sc3_proto || exit 1
{
echo "== correctness"
timeout 120 /tmp/sc3_proto 27 13 14 10 3 3 2 1 2 1024 1024 1 1 2 7 | tail -2
timeout 120 /tmp/sc3_proto 27 14 14 10 1 1 0 1 2 1024 1024 1 1 2 1 | tail -2
timeout 120 /tmp/sc3_proto 27 13 14 10 1 1 0 0 2 512 512 2 2 2 5 | tail -2
timeout 120 /tmp/sc3_proto 26 13 14 10 1 1 1 1 2 1024 512 1 2 1 6 | tail -2
echo "== timings L=32 k=16"
#          oA oB tInA accA reps ntA ntB nbA nbB t1 variant
for v in 0 1 2 4 3 5 7; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 1024 1 1 4 $v | tail -1
done
for v in 0 7; do
  timeout 300 /tmp/sc3_proto 32 16 14 10 3 3 2 1 5 1024 1024 1 1 4 $v | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 3 2 1 5 1024 1024 1 1 2 $v | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 512 512 2 2 4 $v | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 1 5 1024 512 1 2 4 $v | tail -1
  timeout 300 /tmp/sc3_proto 32 16 14 10 1 1 0 0 5 1024 1024 1 1 4 $v | tail -1
done
} 2>&1 | tee $OUT/sc3_proto.txt
{
for cfg in "1 1 0 1 3 1024 1024 1 1 4 7"; do
  echo "== PMC: $cfg"
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES"; do
    timeout 600 tools/pmc_kernels.sh sc3_ "$G" -- /tmp/sc3_proto 32 16 14 10 $cfg
  done
done
} 2>&1 | tee $OUT/sc3_pmc.txt
