#!/bin/bash
# round 3, GPU session 9: the whole GPU suite, the bench line and its rocprofv3 summaries
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s9; mkdir -p $OUT
echo "== pytest -m gpu"; timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -30 | tee $OUT/pytest_gpu.txt
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3 | tee $OUT/smoke.txt
echo "== bench"; timeout 900 python bench.py 2> $OUT/bench_stderr.txt | tee $OUT/bench.json
echo "== profile_bench"; timeout 1500 bash tools/profile_bench.sh 2>&1 | tail -5
echo "== 2-rank flow on one GPU (gloo-staged)"; timeout 900 python bench.py --gpus 2 --L 24 --steps 3 --warmup 1 2>&1 | tail -2 | tee $OUT/bench_2rank_staged.json
