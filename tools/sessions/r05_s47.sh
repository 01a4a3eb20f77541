#!/bin/bash
# round 5, GPU session 47: filter degree, lower end; and the no-Z2 run
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s47; mkdir -p $OUT
M=$OUT/kagome_degree_low.txt
for D in 3 4 5 6 7 8; do
  echo "== DNM_EIGS_FILTER_DEGREE=$D" | tee -a $M
  export DNM_EIGS_FILTER_DEGREE=$D
  DNM_EXPERIMENTAL=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep "Solve completed\|multiplies" | cut -c1-100 | tee -a $M
done
for D in 6 8 13; do
  echo "== --no-z2 DNM_EIGS_FILTER_DEGREE=$D" | tee -a $M
  export DNM_EIGS_FILTER_DEGREE=$D
  DNM_EXPERIMENTAL=1 python3 benchmarking/run_kagome.py 30 --no-z2 2>&1 | grep "Solve completed\|multiplies" | cut -c1-100 | tee -a $M
done
