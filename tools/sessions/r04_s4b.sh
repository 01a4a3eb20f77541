#!/bin/bash
# round 4, GPU session 4b: real-arithmetic eigsolve (DNM_MAT_REAL_PACKED) parity tests and timings, then the degree sweep
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s4; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python -m pytest tests/test_gpu_krylov.py -q -x -k "real_packed or real_arithmetic or filtered_default" 2>&1 | tail -8 | tee $OUT/pytest_real.txt
# L=30: nev=1 (basis-free Lanczos) complex against real arithmetic
for r in 0 1; do
  echo "== DNM_EIGS_REAL=$r" | tee -a $OUT/real_L30.txt
  DNM_EIGS_REAL=$r timeout 600 python tools/krylov_L30.py 30 2>&1 | grep eigsolve | tee -a $OUT/real_L30.txt
done
for r in 0 1; do
  echo "== nev=3 tol 1e-8 DNM_EIGS_REAL=$r" | tee -a $OUT/real_L30.txt
  DNM_KRYLOV_DEBUG=1 DNM_EIGS_REAL=$r timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain 2>&1 | grep -E "filtered\)|L=30" | grep -v "restart [0-9]*," | tee -a $OUT/real_L30.txt
done
bash tools/sessions/r04_s4.sh
