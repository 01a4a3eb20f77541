#!/bin/bash
# round 4, GPU session 25: validation of the tree as committed -- the full GPU suite, smoke, the default bench line
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s25; mkdir -p $OUT
timeout 2700 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $OUT/pytest_gpu.txt
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 600 python3 bench.py 2>$OUT/bench.err | tail -1 > $OUT/bench_line.json; cut -c1-400 $OUT/bench_line.json
