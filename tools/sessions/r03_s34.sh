#!/bin/bash
# round 3, GPU session 34: full verification -- smoke, default bench line, the whole GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s34; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee $OUT/smoke.txt
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_default_line.json; cut -c1-330 $OUT/bench_default_line.json
timeout 5000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
