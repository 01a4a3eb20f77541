#!/bin/bash
# round 3, GPU session 41: window pass at 512 threads x 8 entries (DNM_SC3_WIN=2) and with two gathers in flight (=1)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s41; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for m in 0 2 1; do
  export DNM_SC3_WIN=$m
  echo "== DNM_SC3_WIN=$m" | tee -a $OUT/win_modes.txt
  bash tools/prof_cmd.sh /tmp/st_$m.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_.*pass" /tmp/st_$m.txt | cut -c1-130 | tee -a $OUT/win_modes.txt
  timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of|split" | tee -a $OUT/win_modes.txt
done
