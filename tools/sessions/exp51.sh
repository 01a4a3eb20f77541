#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp51b_sc_nb.txt
echo "# SpinConserve block kernel: partner blocks of 2 / 3 / 4 high bonds requested together" > $O
for nb in 1 2 1 2; do
  echo "## SC_NB=$nb" >> $O
  DNM_LIB=$PWD/dynamite_amd/build/lib_nb$nb.so timeout 600 python3 tools/sc_bench.py 32 2>&1 | grep "^SpinConserve" >> $O
done
