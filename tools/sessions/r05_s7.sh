#!/bin/bash
# round 5, GPU session 7: window-pass dispatch of the bond-graph passes by (Lo population, column block), A/B with counters
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r05_s7; mkdir -p $OUT
M=$OUT/worder.txt
for ord in 0 1; do
  echo "== DNM_SC3G_WORDER=$ord" | tee -a $M
  DNM_SC3G_WORDER=$ord timeout 600 python3 tools/models_bench.py kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep "CASE\|multiply" | cut -c1-120 | tee -a $M
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- kagome30:sc: $G" | tee -a $M
    DNM_SC3G_WORDER=$ord bash tools/pmc_kernels.sh sc3g_win "$G" -- python3 tools/models_bench.py kagome30:sc | tee -a $M
  done
done
DNM_SC3G_WORDER=1 timeout 600 python3 -m pytest tests/test_gpu_sc3_graph.py -q 2>&1 | tail -3 | tee -a $M
