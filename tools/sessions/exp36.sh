#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp36_rows_per_thread.txt
echo "# tile size x rows per thread under the round-2 defaults (4 rows per thread: 76 registers -> 6 waves per SIMD at B=11)" > $O
SWEEP='[{"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 11, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 11, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 10, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 12, "R": 2, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 10, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}, {"B": 12, "R": 3, "mode": 2, "amin": 4, "g": 6, "cp": 98}]' timeout 900 python3 tools/sweep.py 30 2>&1 | grep -v amdgpu.ids >> $O
