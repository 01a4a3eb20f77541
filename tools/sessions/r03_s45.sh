#!/bin/bash
# round 3, GPU session 45: window pass with one workgroup per CU (LDS padded to 100 KB) -- do the gathers merge better?
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s45; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
for kb in 0 100; do
  export DNM_SC3_WIN_LDS_KB=$kb
  echo "== window pass LDS request padded to $kb KB" | tee -a $OUT/win_lds.txt
  bash tools/prof_cmd.sh /tmp/st_$kb.txt python3 tools/sc_bench.py 32 > /dev/null
  grep "sc3_.*pass" /tmp/st_$kb.txt | cut -c1-130 | tee -a $OUT/win_lds.txt
  bash tools/pmc_kernels.sh sc3_win 'FETCH_SIZE' -- python3 tools/sc_bench.py 32 | tee -a $OUT/win_lds.txt
  timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of" | tee -a $OUT/win_lds.txt
  bash tools/pmc_kernels.sh sc3_win 'FETCH_SIZE' -- python3 tools/sc3_config5.py --rank 3 | tee -a $OUT/win_lds.txt
done
