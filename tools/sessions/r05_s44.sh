#!/bin/bash
# round 5, GPU session 44: eigsolve's memory-short fallback to deflation, forced at small size
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s44; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_krylov.py -m gpu -q -k "memory_is_short or deflated" 2>&1 | tail -25 | cut -c1-220 | tee $OUT/fallback.txt
