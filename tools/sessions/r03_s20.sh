#!/bin/bash
# round 3, GPU session 20: the tiled multiply with a rotated vector layout (timing experiment, DNM_ROT)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s20; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python tools/rot_probe.py 30 6,18,3 8,18,3 7,18,3 9,18,3 6,18,2 6,18,4 10,18,3 4,18,3 2>&1 | grep -v amdgpu.ids | tee $OUT/rot_probe.txt
