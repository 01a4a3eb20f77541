#!/bin/bash
# round 5, GPU session 33: Parity known answer, then the full suite as trimmed
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s33; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "xx_models" 2>&1 | grep -v amdgpu | tail -12 | cut -c1-220 | tee $OUT/xx.txt
timeout 1500 python3 -m pytest tests -m gpu -q --durations=15 2>&1 | tail -25 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
