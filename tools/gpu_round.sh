#!/bin/bash
# One GPU-box session: tests, a bench line, a plan sweep and rocprof summaries.
# Everything is wrapped in `timeout`; results land in gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-.}"
OUT=gpurun_out
mkdir -p $OUT
export TMPDIR=/tmp
echo "== build check" | tee $OUT/log.txt
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -5 | tee -a $OUT/log.txt
echo "== pytest -m gpu" | tee -a $OUT/log.txt
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 | tee $OUT/pytest_gpu.txt
echo "== sweep L=26" | tee -a $OUT/log.txt
timeout 300 python tools/sweep.py 26 2>&1 | tee $OUT/sweep_L26.txt
echo "== sweep L=30" | tee -a $OUT/log.txt
timeout 600 python tools/sweep.py 30 2>&1 | tee $OUT/sweep_L30.txt
echo "== bench" | tee -a $OUT/log.txt
timeout 600 python bench.py --steps 10 --warmup 2 2> $OUT/bench_stderr.txt | tee $OUT/bench.json
