#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp54_early_wy.txt
echo "# the late y of the accumulating pass requested ahead of the LDS loops (cache policy bit 8 = 256)" > $O
echo "## new build" >> $O
SWEEP='[{"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 482, "env": {}}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 482, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 482, "env": {}}]' timeout 1200 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids | grep "^L=" >> $O
echo "## previous build (no such option)" >> $O
DNM_LIB=$PWD/dynamite_amd/build/lib_prev.so SWEEP='[{"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}]' timeout 1200 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids | grep "^L=" >> $O
