#!/bin/bash
# round 3, GPU session 46: window pass with 128-byte runs (32 KB tiles, four 512-thread workgroups per CU): DNM_SC3_RUN8=1
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s46; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for m in 1 0 1 0; do
  export DNM_SC3_RUN8=$m
  echo "== DNM_SC3_RUN8=$m" | tee -a $OUT/run8.txt
  bash tools/prof_cmd.sh /tmp/st_$m.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_.*pass" /tmp/st_$m.txt | cut -c1-130 | tee -a $OUT/run8.txt
  timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of|split" | tee -a $OUT/run8.txt
done
export DNM_SC3_RUN8=1
timeout 900 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -3 | tee $OUT/pytest_sc3_run8.txt
bash tools/pmc_kernels.sh sc3_win 'FETCH_SIZE' -- python3 tools/sc3_config5.py --rank 3 | tee $OUT/fetch_run8.txt
