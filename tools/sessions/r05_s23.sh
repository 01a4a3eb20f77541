#!/bin/bash
# round 5, GPU session 23: the tree as committed -- smoke, full GPU suite with durations, bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s23; mkdir -p $OUT
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -25 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt; tail -c 400 $OUT/bench_line.json
