#!/bin/bash
# round 5, GPU session 3: bond-graph passes with site relabelling and XParity: multiply + eigsolve(nev=2), then the suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s3; mkdir -p $OUT
M=$OUT/models.txt
timeout 1500 python3 tools/models_bench.py --eigs kagome27b:sc kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $M
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  echo "-- kagome30:scx: $G" | tee -a $M
  bash tools/pmc_kernels.sh sc3 "$G" -- python3 tools/models_bench.py kagome30:scx | grep -v "random\|copy" | tee -a $M
done
timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -30 > $OUT/tests.txt; tail -4 $OUT/tests.txt
