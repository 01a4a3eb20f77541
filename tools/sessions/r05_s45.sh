#!/bin/bash
# round 5, GPU session 45: sliced launches of the one-thread-per-row kernels at small size
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s45; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_matvec.py -m gpu -q -k "slices" 2>&1 | tail -25 | cut -c1-220 | tee $OUT/slices.txt
