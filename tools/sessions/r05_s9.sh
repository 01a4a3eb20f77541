#!/bin/bash
# round 5, GPU session 9: the (Lo population, column block) window order on CHAINS (DNM_SC3G_WORDER=2), SpinConserve(32,16)
# and one rank of config 5, with counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r05_s9; mkdir -p $OUT
M=$OUT/chain_worder.txt
for ord in 1 2; do
  echo "== DNM_SC3G_WORDER=$ord (1: chains keep their order; 2: the bond-graph order)" | tee -a $M
  DNM_SC3G_WORDER=$ord timeout 600 python3 tools/sc_bench.py --model heisenberg 32 2>&1 | grep "diag_cached=0" | tee -a $M
  DNM_SC3G_WORDER=$ord timeout 600 python3 tools/sc_bench.py --model heisenberg --real 32 2>&1 | grep "REAL" | tee -a $M
  DNM_SC3G_WORDER=$ord timeout 900 python3 tools/sc3_config5.py --rank 3 2>&1 | grep -i "ms\b\|ms " | tail -3 | tee -a $M
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- (32,16): $G" | tee -a $M
    DNM_SC3G_WORDER=$ord bash tools/pmc_kernels.sh sc3_win "$G" -- python3 tools/sc_bench.py --model heisenberg 32 | tee -a $M
  done
done
DNM_SC3G_WORDER=2 timeout 600 python3 -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -3 | tee -a $M
