#!/bin/bash
# round 4, GPU session 4: degree of the Chebyshev filter when memory keeps the basis short (VERDICT r3 task 6):
# eigsolve(nev=3, tol 1e-8) at L=30 (16 GiB vectors, about 10 fit) under DNM_EIGS_FILTER_DEGREE
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s4; mkdir -p $OUT
export DNM_EXPERIMENTAL=1 DNM_KRYLOV_DEBUG=1 DNM_EIGS_REAL=0
for d in 0 15 21 29 41; do
  if [ $d -gt 0 ]; then export DNM_EIGS_FILTER_DEGREE=$d; fi
  echo "== degree ${d} (0: the rule)" | tee -a $OUT/degree_L30.txt
  timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain 2>&1 | grep -E "filtered\)|L=30" | grep -v "restart [0-9]*," | tee -a $OUT/degree_L30.txt
done
unset DNM_EIGS_FILTER_DEGREE
for d in 0 15 21; do
  if [ $d -gt 0 ]; then export DNM_EIGS_FILTER_DEGREE=$d; fi
  echo "== L=28 nev=5 tol 1e-10 degree ${d}" | tee -a $OUT/degree_L28.txt
  timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain 2>&1 | grep -E "filtered\)|L=28" | grep -v "restart [0-9]*," | tee -a $OUT/degree_L28.txt
done
