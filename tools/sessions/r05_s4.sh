#!/bin/bash
# round 5, GPU session 4: lo-pass dispatch order of the bond-graph passes (window-bit orbits per XCD) A/B with counters; real arithmetic
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r05_s4; mkdir -p $OUT
M=$OUT/order.txt
for ord in 0 1; do
  echo "== DNM_SC3G_ORDER=$ord" | tee -a $M
  DNM_SC3G_ORDER=$ord timeout 600 python3 tools/models_bench.py kagome30:sc kagome30:scx kagome27b:sc 2>&1 | grep "CASE\|multiply" | tee -a $M
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- kagome30:sc: $G" | tee -a $M
    DNM_SC3G_ORDER=$ord bash tools/pmc_kernels.sh sc3g "$G" -- python3 tools/models_bench.py kagome30:sc | tee -a $M
  done
done
echo "== eigsolve(nev=2), default (real arithmetic where it applies)" | tee -a $M
timeout 900 python3 tools/models_bench.py --eigs kagome30:sc kagome27b:sc 2>&1 | grep "CASE\|multiply\|eigsolve" | tee -a $M
timeout 600 python3 -m pytest tests/test_gpu_sc3_graph.py -q 2>&1 | tail -3 | tee -a $M
