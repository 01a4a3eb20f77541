#!/bin/bash
# round 4, GPU session 35: kernel times of the reduced density matrix, vector unit against matrix cores (L=26, keep 10 / 13 low spins)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s35; mkdir -p $OUT
for k in 10 13; do
  export DNM_RDM_MFMA=0
  echo "== k=$k vector unit" | tee -a $OUT/kernels.txt
  bash tools/prof_cmd.sh $OUT/v_$k.txt python3 tools/rdm_bench.py 26 $k | grep -E "rdm_|keep" | tee -a $OUT/kernels.txt
  unset DNM_RDM_MFMA
  echo "== k=$k matrix cores" | tee -a $OUT/kernels.txt
  bash tools/prof_cmd.sh $OUT/m_$k.txt python3 tools/rdm_bench.py 26 $k | grep -E "rdm_|keep" | tee -a $OUT/kernels.txt
done
