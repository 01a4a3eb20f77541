#!/bin/bash
# round 4, GPU session 37: rdm_mfma_kernel with loop-invariant row parts and shared traced deposits: parity + kernel times
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s37; mkdir -p $OUT
timeout 600 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_distributed.py -m gpu -x -q -k "rdm" 2>&1 | tail -3 | tee $OUT/parity.txt
for k in 10 13; do
  echo "== k=$k matrix cores" | tee -a $OUT/kernels.txt
  timeout 300 bash tools/prof_cmd.sh $OUT/m_$k.txt python3 tools/rdm_bench.py 26 $k | grep -E "rdm_|keep" | tee -a $OUT/kernels.txt
done
