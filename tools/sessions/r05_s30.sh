#!/bin/bash
# round 5, GPU session 30: XX chain / ring in SpinConserve at full size against the filled Fermi sea
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s30; mkdir -p $OUT
DNM_TEST_LARGEST=1 timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "xx_models" 2>&1 | grep -v amdgpu | tail -30 | cut -c1-250 | tee $OUT/xx_free_fermions.txt
