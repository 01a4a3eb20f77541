#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp7_persist.txt
echo "# persistent kernel, more waves" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":2,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":2,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":11,"R":2,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_PERSIST_WGS_PER_CU":2}},
{"B":11,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_PERSIST_WGS_PER_CU":2}},
{"B":11,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":3,"mode":0,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":3,"mode":0,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
