#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp53_sc_chunk.txt
echo "# SpinConserve block order inside chunks of 2^c high parts (Infinity-Cache locality of the partner blocks)" > $O
for c in 0 13 12 14 11 16 0; do
  echo "## DNM_SC_CHUNK=$c" >> $O
  DNM_SC_CHUNK=$c timeout 600 python3 tools/sc_bench.py 32 2>&1 | grep "^SpinConserve.*cached=1" >> $O
done
