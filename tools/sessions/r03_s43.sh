#!/bin/bash
# round 3, GPU session 43: lo pass with 512 threads x 7 entries against 1024 x 4, alternating
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s43; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for m in 512 1024 512 1024; do
  export DNM_SC3_LO_THREADS=$m
  echo "== lo pass with $m threads" | tee -a $OUT/lo512.txt
  bash tools/prof_cmd.sh /tmp/st_$m.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_.*pass" /tmp/st_$m.txt | cut -c1-130 | tee -a $OUT/lo512.txt
  timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of|split" | tee -a $OUT/lo512.txt
done
