#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp49_split.txt
echo "# where the group bits are split between the two passes (new order)" > $O
SWEEP='[{"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 5, "cp": 226, "env": {"DNM_GBITS_WINDOW": 6}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 4, "cp": 226, "env": {"DNM_GBITS_WINDOW": 7}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 5, "cp": 226, "env": {"DNM_GBITS_WINDOW": 6, "DNM_DIAG_PASS": "first"}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {"DNM_DIAG_PASS": "first"}}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 226, "env": {}}]' PROBE_DESCRIBE=1 timeout 1200 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids | grep "^L=\|local pass" >> $O
