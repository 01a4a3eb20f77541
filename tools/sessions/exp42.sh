#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp42_window_first_wide.txt
echo "# window pass first: parity, sizes, Krylov" > $O
DNM_WINDOW_FIRST=1 timeout 1500 python3 -m pytest tests/test_gpu_matvec.py tests/test_gpu_krylov.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3 >> $O
for w in 1 0 1 0; do
  echo "## DNM_WINDOW_FIRST=$w" >> $O
  DNM_WINDOW_FIRST=$w timeout 600 python3 tools/size_scan.py 20 22 24 26 28 30 2>&1 | grep "^L=" | cut -c1-60 >> $O
done
for w in 1 0; do
  echo "## DNM_WINDOW_FIRST=$w" >> $O
  DNM_WINDOW_FIRST=$w timeout 900 python3 tools/krylov_L30.py 2>&1 | grep -v amdgpu.ids | grep "L=30" >> $O
  DNM_WINDOW_FIRST=$w timeout 900 python3 tools/krylov_bench.py 26 xxz 2>&1 | grep -v amdgpu.ids | tail -6 >> $O
done
