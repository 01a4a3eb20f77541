#!/bin/bash
# round 5, GPU session 17: two gathered hops in flight in the real lo pass (A/B against one)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s17; mkdir -p $OUT
M=$OUT/twohops.txt
DNM_EXPERIMENTAL=1 timeout 600 python3 -m pytest tests/test_gpu_sc3_graph.py tests/test_gpu_sc3.py -m gpu -q -x -k "real" 2>&1 | tail -2 | tee -a $M
for V in 1 0 1 0; do
  echo "== DNM_SC3G_TWOHOPS=$V" | tee -a $M
  if [ $V = 0 ]; then export DNM_LIB=$PWD/dynamite_amd/exp/libdnm_twohops0.so; else unset DNM_LIB; fi
  python3 tools/models_bench.py --real kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | grep "CASE\|real arith" | cut -c1-150 | tee -a $M
done
