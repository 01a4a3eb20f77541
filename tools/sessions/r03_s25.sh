#!/bin/bash
# round 3, GPU session 25: window partitions split by rows (local rows under the exchange) -- distributed tests
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s25; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 2400 python -m pytest tests/test_gpu_distributed.py -q -x 2>&1 | tail -15 | tee $OUT/pytest_distributed.txt
