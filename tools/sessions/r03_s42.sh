#!/bin/bash
# round 3, GPU session 42: window pass at 512 threads x 8 entries (default now) against 1024 x 4, twice each; sc3 tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s42; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for m in 512 1024 512 1024; do
  export DNM_SC3_WIN_THREADS=$m
  echo "== window pass with $m threads" | tee -a $OUT/win512.txt
  bash tools/prof_cmd.sh /tmp/st_$m.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_.*pass" /tmp/st_$m.txt | cut -c1-130 | tee -a $OUT/win512.txt
  timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of|split" | tee -a $OUT/win512.txt
done
unset DNM_SC3_WIN_THREADS
timeout 900 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -3 | tee $OUT/pytest_sc3.txt
