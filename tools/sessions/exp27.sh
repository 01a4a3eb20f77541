#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp27_b13.txt
echo "# 128 KB tiles (one workgroup per CU) under the swizzled layout" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},
{"B":13,"R":3,"mode":2,"amin":4,"g":5,"cp":98},
{"B":13,"R":4,"mode":2,"amin":4,"g":5,"cp":98},
{"B":13,"R":3,"mode":2,"amin":4,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_TILE_BITS_WINDOW":13}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_TILE_BITS_WINDOW":11}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
