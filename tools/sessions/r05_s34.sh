#!/bin/bash
# round 5, GPU session 34: the partitioned known answer (sc_big on three ranks) and the cases touched last
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s34; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_distributed.py -m gpu -q -k "sc_big or xparity" 2>&1 | tail -8 | cut -c1-220 | tee $OUT/sc_big.txt
