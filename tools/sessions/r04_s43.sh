#!/bin/bash
# round 4, GPU session 43: streaming kernel for reduced density matrices of 1-3 spins: parity, then against the tiled form
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r04_s43; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -m gpu -x -q -k "rdm or entrop" 2>&1 | tail -3 | tee $OUT/parity.txt
for v in 0 1 0 1; do
  echo "== DNM_RDM_SMALL=$v" | tee -a $OUT/ab.txt
  DNM_RDM_SMALL=$v timeout 300 python3 tools/rdm_bench.py 26 2>&1 | grep -E "keep=\[" | tee -a $OUT/ab.txt
done
