#!/bin/bash
# round 5, GPU session 10: final numbers on the final tree (models table, run_kagome, bench, suite)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s10; mkdir -p $OUT
M=$OUT/models_final.txt
timeout 1800 python3 tools/models_bench.py --eigs kagome27b:sc kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-260 | tee $M
timeout 600 python3 tools/models_bench.py bench_long_range:sc:28 heisenberg:sc:32 mbl:full:28 bench_long_range:full:28 2>&1 | grep "CASE\|multiply" | cut -c1-120 | tee -a $M
for c in kagome30:sc kagome30:scx heisenberg:sc:32; do
  for G in "FETCH_SIZE" "WRITE_SIZE"; do
    echo "-- $c: $G" | tee -a $M
    bash tools/pmc_kernels.sh sc3 "$G" -- python3 tools/models_bench.py $c | grep -v "random\|copy" | tee -a $M
  done
done
python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -4 | tee -a $M
python3 benchmarking/run_kagome.py 30 --no-z2 2>&1 | grep -v amdgpu | tail -4 | tee -a $M
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -25 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
