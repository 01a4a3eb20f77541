#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp50_per_pass_shapes.txt
echo "# different tile size / rows per thread for the window pass" > $O
SWEEP='[{"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_LOG_ROWS_WINDOW": 3}}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_LOG_ROWS_WINDOW": 2}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_TILE_BITS_WINDOW": 11}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_TILE_BITS_WINDOW": 11, "DNM_GBITS_WINDOW": 6}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_TILE_BITS_WINDOW": 13, "DNM_LOG_ROWS_WINDOW": 3}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}]' PROBE_DESCRIBE=1 timeout 1200 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids | grep "^L=\|local pass" >> $O
