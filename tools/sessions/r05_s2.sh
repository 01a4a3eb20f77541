#!/bin/bash
# round 5, GPU session 2: the bond-graph passes in the identity labelling (no site permutation yet)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s2; mkdir -p $OUT
M=$OUT/models.txt
timeout 900 python3 tools/models_bench.py kagome27b:sc kagome30:sc bench_long_range:sc:28 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $M
echo "== chain through the graph kernels (DNM_SC3_GRAPH=1) against the chain kernels" | tee -a $M
timeout 600 python3 tools/sc_bench.py --model heisenberg 28 32 2>&1 | grep -v "amdgpu.ids\|norm" | tee -a $M
DNM_SC3_GRAPH=1 timeout 600 python3 tools/sc_bench.py --model heisenberg 28 32 2>&1 | grep -v "amdgpu.ids\|norm" | tee -a $M
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  echo "-- kagome30:sc: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3 "$G" -- python3 tools/models_bench.py kagome30:sc | grep -v "random\|copy" | tee -a $M
done
