#!/bin/bash
# round 5, GPU session 46: filter degree of eigsolve(nev=2) on the 30-site kagome torus (default rule picks 13)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s46; mkdir -p $OUT
M=$OUT/kagome_degree.txt
for D in 0 8 10 13 16 20 26 32; do
  echo "== DNM_EIGS_FILTER_DEGREE=$D (0: the rule)" | tee -a $M
  if [ $D = 0 ]; then unset DNM_EIGS_FILTER_DEGREE; else export DNM_EIGS_FILTER_DEGREE=$D; fi
  DNM_EXPERIMENTAL=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep "Solve completed\|multiplies" | cut -c1-120 | tee -a $M
done
