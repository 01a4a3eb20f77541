#!/bin/bash
# round 3, GPU session 37: lo pass of the SpinConserve kernels: partners computed (0), from the table (1), table + skip (2)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s37; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for m in 0 1 2; do
  export DNM_SC3_LO_MODE=$m
  echo "== lo mode $m" | tee -a $OUT/lo_modes.txt
  bash tools/prof_cmd.sh /tmp/stats_$m.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_" /tmp/stats_$m.txt | cut -c1-130 | tee -a $OUT/lo_modes.txt
done
