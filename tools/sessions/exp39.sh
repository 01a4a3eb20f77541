#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp39_rows4_wide.txt
echo "# 4 rows per thread (8 waves per SIMD) against 8 rows: sizes, Krylov, SpinConserve unaffected" > $O
for r in 2 3 2 3; do
  echo "## DNM_LOG_ROWS=$r" >> $O
  DNM_LOG_ROWS=$r timeout 600 python3 tools/size_scan.py 20 22 24 26 28 30 2>&1 | grep "^L=" | cut -c1-60 >> $O
done
for r in 2 3; do
  echo "## DNM_LOG_ROWS=$r" >> $O
  DNM_LOG_ROWS=$r timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu.ids | grep "L=30" >> $O
  DNM_LOG_ROWS=$r timeout 900 python3 tools/krylov_bench.py 26 xxz 2>&1 | grep -v amdgpu.ids | tail -6 >> $O
done
