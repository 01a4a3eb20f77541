#!/bin/bash
# round 3, GPU session 28: config 5's solver at the largest SpinConserve size one GPU holds (L=34, 2.33 G states)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s28; mkdir -p $OUT
timeout 1200 python tools/sc_eigs_big.py 32 1e-8 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_eigs_32.txt
timeout 1800 python tools/sc_eigs_big.py 34 1e-8 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_eigs_34.txt
