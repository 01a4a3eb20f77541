#!/bin/bash
# round 3, GPU session 14: operators that are not chains on a SpinConserve subspace -- row kernel of the internal layout
# against the reference-order row kernel
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s14; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
{
for m in long_range mbl; do
  echo "== $m, internal layout"
  DNM_SC3_TILED=0 timeout 600 python tools/sc_bench.py --model $m 28 2>&1 | grep -v amdgpu.ids
  echo "== $m, reference order"
  DNM_SC_LAYOUT=0 DNM_SC_BLOCK=0 timeout 600 python tools/sc_bench.py --model $m 28 2>&1 | grep -v amdgpu.ids
done
} | tee $OUT/sc_row_kernels.txt
