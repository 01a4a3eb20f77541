#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp25_persist2.txt
echo "# persistent kernel: register prefetch, 1024 threads x 4 rows / 512 x 8, two gather slots" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":2,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":2,"mode":0,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2}},
{"B":12,"R":2,"mode":0,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}},
{"B":12,"R":3,"mode":0,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":1}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
export SWEEP='[{"B":12,"R":2,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_KERNEL":2,"DNM_PERSIST_STAGE":"dma"}}]'
timeout 900 python3 tools/sweep.py 30 2>&1 | grep L=30 >> $O
DNM_KERNEL=2 DNM_LOG_ROWS=2 timeout 900 python3 tools/v2_check.py 20 2>&1 | tail -1 >> $O
