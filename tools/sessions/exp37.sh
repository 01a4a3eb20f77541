#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp37_waves8.txt
echo "# 4 rows per thread compiled for 8 waves per SIMD (64 registers, 2 workgroups of 1024 threads per CU at B=12)" > $O
echo "## library with DNM_WAVES_4ROWS=8" >> $O
DNM_LIB=$PWD/dynamite_amd/build/lib_w8.so SWEEP='[{"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 11, "R": 2, "mode": 2, "amin": 4, "g": 7, "cp": 98}, {"B": 11, "R": 3, "mode": 2, "amin": 4, "g": 7, "cp": 98}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}]' timeout 900 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids >> $O
echo "## default library" >> $O
SWEEP='[{"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 11, "R": 2, "mode": 2, "amin": 4, "g": 7, "cp": 98}]' timeout 900 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids >> $O
