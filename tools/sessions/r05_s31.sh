#!/bin/bash
# round 5, GPU session 31: evolve at full size against the free-fermion magnetisation profile
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s31; mkdir -p $OUT
DNM_TEST_LARGEST=1 timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "evolve_against" 2>&1 | grep -v amdgpu | tail -30 | cut -c1-250 | tee $OUT/evolve_free_fermions.txt
