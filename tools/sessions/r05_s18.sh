#!/bin/bash
# round 5, GPU session 18: numbers of the tree after the partner table / native transposed exchange / deflated solver
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s18; mkdir -p $OUT
M=$OUT/models_final2.txt
timeout 1800 python3 tools/models_bench.py --eigs --real kagome27b:sc kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | cut -c1-260 | tee $M
timeout 600 python3 tools/models_bench.py --real bench_long_range:sc:28 heisenberg:sc:32 mbl:full:28 2>&1 | grep "CASE\|multiply" | cut -c1-150 | tee -a $M
python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -4 | tee -a $M
python3 benchmarking/run_kagome.py 30 --no-z2 2>&1 | grep -v amdgpu | tail -4 | tee -a $M
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 600 $OUT/bench_line.json
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -25 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
