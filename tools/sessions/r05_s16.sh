#!/bin/bash
# round 5, GPU session 16: shape of the real lo pass (DNM_SC3R_SHAPE 0 / 1 / 2) now that the partner table took the
# instructions out: tiles of 64 KB x 2 workgroups per CU against 32 KB x 4
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s16; mkdir -p $OUT
M=$OUT/shape.txt
for V in 0 1 2; do
  echo "== DNM_SC3R_SHAPE=$V" | tee -a $M
  if [ $V != 0 ]; then export DNM_LIB=$PWD/dynamite_amd/exp/libdnm_shape$V.so; fi
  DNM_EXPERIMENTAL=1 timeout 600 python3 -m pytest tests/test_gpu_sc3_graph.py tests/test_gpu_sc3.py -m gpu -q -x -k "real" 2>&1 | tail -2 | tee -a $M
  python3 tools/models_bench.py --real kagome30:sc kagome30:scx heisenberg:sc:32 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "CASE\|real arith" | cut -c1-150 | tee -a $M
done
