#!/bin/bash
# round 5, GPU session 27: XParity on several ranks (gloo-staged on one GPU) against the oracle
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s27; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_distributed.py -m gpu -q -k "xparity" 2>&1 | tail -40 | cut -c1-220 | tee $OUT/xparity_ranks.txt
