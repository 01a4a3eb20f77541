#!/bin/bash
# round 5, GPU session 29: transverse-field Ising chain on 33 spins in its spin-flip sectors against the free-fermion energy
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s29; mkdir -p $OUT
DNM_TEST_LARGEST=1 DNM_KRYLOV_DEBUG=1 timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k "ising_33" 2>&1 | grep -v amdgpu | tail -15 | cut -c1-220 | tee $OUT/ising33.txt
