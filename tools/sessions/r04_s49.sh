#!/bin/bash
# round 4, GPU session 49: 128 x 64 MFMA tiles for the reduced density matrix (DNM_RDM_128=1) against the 64 x 64 ones
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r04_s49; mkdir -p $OUT
DNM_RDM_128=1 DNM_FUZZ_RDM_N=100 timeout 400 python3 -m pytest tests/test_gpu_krylov.py -m gpu -x -q -k "rdm or entrop" 2>&1 | tail -2 | tee $OUT/parity.txt
for v in 0 1 0 1; do
  for k in 10 13; do
    echo "== DNM_RDM_128=$v k=$k" | tee -a $OUT/kernels.txt
    DNM_RDM_128=$v timeout 200 bash tools/prof_cmd.sh $OUT/v${v}_$k.txt python3 tools/rdm_bench.py 26 $k | grep -E "rdm_mfma" | cut -c1-36,100-140 | tee -a $OUT/kernels.txt
  done
done
