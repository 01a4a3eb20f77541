#!/bin/bash
# round 5, GPU session 24: the opt-in largest-problem tests on the final tree (DNM_TEST_LARGEST=1), native comm child
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s24; mkdir -p $OUT
DNM_TEST_LARGEST=1 timeout 2400 python3 -m pytest tests/test_gpu_fullsize.py tests/test_gpu_distributed.py -m gpu -q --durations=12 -k "not kagome36" 2>&1 | tail -20 | tee $OUT/largest.txt
