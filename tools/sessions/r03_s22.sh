#!/bin/bash
# round 3, GPU session 22: phase stamps of tile_pass_kernel workgroups (diagnostic build, -DDNM_PHASE_TIMING)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s22; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
nm -D dynamite_amd/libdynamite_amd.so | grep -q dnm_debug_phase_buffer || { echo "needs the diagnostic build"; exit 1; }
for L in 30 24 22; do timeout 600 python tools/phase_times.py $L 2>&1 | grep -v amdgpu.ids | tee $OUT/phase_times_L$L.txt; done
