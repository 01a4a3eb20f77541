#!/bin/bash
# round 4, GPU session 34: the reduced density matrix on the matrix cores (rdm_mfma_kernel) -- parity, then A/B
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s34; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -m gpu -x -q -k "rdm" 2>&1 | tail -4 | tee $OUT/parity.txt
for i in 1 2; do
  echo "== vector unit (DNM_RDM_MFMA=0)" | tee -a $OUT/ab.txt
  DNM_EXPERIMENTAL=1 DNM_RDM_MFMA=0 timeout 600 python3 tools/rdm_bench.py 26 2>&1 | grep -E "spins from|keep=\[" | grep "Full" | tee -a $OUT/ab.txt
  echo "== matrix cores" | tee -a $OUT/ab.txt
  timeout 600 python3 tools/rdm_bench.py 26 2>&1 | grep -E "spins from|keep=\[" | grep "Full" | tee -a $OUT/ab.txt
done
