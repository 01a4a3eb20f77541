#!/bin/bash
# round 5, GPU session 28: fuzz of random operators inside XParity sectors (complex and real arithmetic)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s28; mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_matvec.py -m gpu -q -k "fuzz_xparity" 2>&1 | tail -40 | cut -c1-250 | tee $OUT/fuzz_xparity.txt
