#!/bin/bash
# round 5, GPU session 1: evidence before building (VERDICT r4 task 1a / 2 / 4):
#  1. the reference harness's Hamiltonians + kagome at size: plan, ms per multiply, counters (profiles/r05_models.txt)
#  2. bench.py with the warm Krylov numbers
#  3. pytest -m gpu --durations=25
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s1; mkdir -p $OUT
M=$OUT/models.txt
echo "== multiply, one MI355X (tools/models_bench.py)" > $M
timeout 900 python3 tools/models_bench.py kagome27b:sc kagome30:sc kagome30:scx bench_long_range:sc:28 bench_long_range:full:28 \
   bench_ising:full:28 bench_xx:full:28 mbl:full:28 2>&1 | grep -v Warning | tee -a $M
timeout 600 python3 tools/models_bench.py bench_syk:full:14 bench_syk:full:16 2>&1 | tail -12 | tee -a $M
echo "== eigsolve(nev=2), kagome" >> $M
timeout 900 python3 tools/models_bench.py --eigs kagome27b:sc kagome30:sc kagome30:scx 2>&1 | grep -v Warning | tee -a $M
echo "== counters (last dispatch of every matching kernel; FETCH_SIZE in KB, x2 per the guide)" >> $M
for c in kagome30:sc bench_long_range:sc:28; do
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- $c: $G" | tee -a $M
    bash tools/pmc_kernels.sh sc3_ "$G" -- python3 tools/models_bench.py $c | grep -v "random\|copy" | tee -a $M
  done
done
for c in bench_long_range:full:28 bench_ising:full:28; do
  for G in "FETCH_SIZE" "WRITE_SIZE"; do
    echo "-- $c: $G" | tee -a $M
    NLAST=4 bash tools/pmc_kernels.sh tile_pass "$G" -- python3 tools/models_bench.py $c | tee -a $M
  done
done
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 3000 $OUT/bench_line.json
timeout 1500 python3 -m pytest tests -m gpu -q -x --durations=40 2>&1 | tail -60 > $OUT/test_durations.txt; tail -5 $OUT/test_durations.txt
