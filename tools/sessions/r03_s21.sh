#!/bin/bash
# round 3, GPU session 21: y-traffic cache policy against the size of the vectors (Infinity Cache residency)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s21; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python tools/policy_sizes.py 18 20 21 22 23 24 25 26 28 2>&1 | grep -v amdgpu.ids | tee $OUT/policy_sizes.txt
