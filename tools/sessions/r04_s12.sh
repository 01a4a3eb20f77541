#!/bin/bash
# round 4, GPU session 12: full GPU suite and the default bench line on the final code
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s12; mkdir -p $OUT
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_stderr.txt; tail -c 600 $OUT/bench_default.json
timeout 3000 python -m pytest tests -q -x -m gpu -o faulthandler_timeout=900 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $OUT/smoke.txt
