#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp6_persist.txt
echo "# persistent kernel with a loader wave (DNM_KERNEL=2)" > $O
DNM_KERNEL=2 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -15 >> $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":3,"mode":2,"amin":4,"g":5,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_PERSIST_WGS_PER_CU":2}},
{"B":11,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
timeout 1500 bash tools/prof_multi.sh 30 '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}}' '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}}' >> $O 2>&1
