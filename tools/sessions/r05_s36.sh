#!/bin/bash
# round 5, GPU session 36: random-field XX chain on the Full space at L=30 against free fermions in a potential
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s36; mkdir -p $OUT
timeout 900 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -s -k "xx_models and (field30 or parity30)" 2>&1 | grep -v amdgpu | tail -8 | cut -c1-220 | tee $OUT/field30.txt
