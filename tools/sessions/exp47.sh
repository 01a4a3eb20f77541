#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp47_rows4_sizes.txt
echo "# 4 rows per thread + diagonal in the accumulating pass (C) against 8 rows + diagonal in the first pass (A): sizes, Krylov" > $O
for v in C A C A; do
  echo "## $v" >> $O
  if [ $v = C ]; then export DNM_LOG_ROWS=2 DNM_DIAG_PASS=last; else unset DNM_LOG_ROWS DNM_DIAG_PASS; fi
  timeout 600 python3 tools/size_scan.py 20 22 24 25 26 27 28 29 30 2>&1 | grep "^L=" | cut -c1-60 >> $O
done
for v in C A; do
  echo "## $v" >> $O
  if [ $v = C ]; then export DNM_LOG_ROWS=2 DNM_DIAG_PASS=last; else unset DNM_LOG_ROWS DNM_DIAG_PASS; fi
  timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu.ids | grep "L=30" >> $O
  timeout 900 python3 tools/krylov_bench.py 26 xxz 2>&1 | grep -v amdgpu.ids | tail -6 >> $O
  timeout 900 python3 tools/cheb_bench.py 30 2>&1 | grep -v amdgpu.ids | grep "t=1 " >> $O
done
