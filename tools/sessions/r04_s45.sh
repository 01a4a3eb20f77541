#!/bin/bash
# round 4, GPU session 45: basis-free Lanczos at L=30 with the update folded into the next multiply against the plain driver
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
OUT=gpurun_out/r04_s45; mkdir -p $OUT
for mode in complex real; do
  for d in 0 1 0 1; do
    echo "== $mode DNM_EIGS_DEFER=$d" | tee -a $OUT/lanczos.txt
    DNM_EIGS_DEFER=$d DNM_KRYLOV_DEBUG=1 timeout 300 python3 tools/lanczos_prof.py 30 $mode 2>&1 | grep -E "eigsolve|dnm_eigsolve" | tail -3 | tee -a $OUT/lanczos.txt
  done
done
