#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp52_sc_nt.txt
echo "# SpinConserve block kernel: threads per block (rows per thread 2 / 4 / 7 at 1024 / 512 / 256 threads)" > $O
for nt in 1024 512 256 1024 512; do
  echo "## NT=$nt" >> $O
  if [ $nt = 512 ]; then unset DNM_LIB; else export DNM_LIB=$PWD/dynamite_amd/build/lib_nt$nt.so; fi
  timeout 600 python3 tools/sc_bench.py 32 2>&1 | grep "^SpinConserve" >> $O
done
