#!/bin/bash
# round 4, GPU session 39: rdm_mfma_kernel -- waves per SIMD and chunk size (variants built by tools/build_variant.py)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r04_s39; mkdir -p $OUT
for v in default rdm_w4 rdm_s2w2 rdm_s2w3 default rdm_w4; do
  if [ $v = default ]; then unset DNM_LIB; else export DNM_LIB=$PWD/dynamite_amd/build/exp/lib_$v.so; fi
  echo "== $v" | tee -a $OUT/kernels.txt
  timeout 200 python3 -m pytest tests/test_gpu_krylov.py -m gpu -x -q -k "rdm" 2>&1 | tail -1 | tee -a $OUT/kernels.txt
  for k in 10 13; do
    timeout 200 bash tools/prof_cmd.sh $OUT/${v}_$k.txt python3 tools/rdm_bench.py 26 $k | grep -E "rdm_mfma" | cut -c1-30,100-140 | tee -a $OUT/kernels.txt
  done
done
