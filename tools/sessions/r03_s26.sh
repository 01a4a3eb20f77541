#!/bin/bash
# round 3, GPU session 26: profiles of the default bench (kernel with the new addressing), default bench line, full GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s26; mkdir -p $OUT
bash tools/profile_bench.sh > $OUT/profile_bench.txt 2>&1
cp -r gpurun_out/profiles $OUT/profiles
timeout 900 python bench.py 2>/dev/null | tail -1 > $OUT/bench_default_line.json
timeout 5000 python -m pytest tests -q -m gpu -x 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
