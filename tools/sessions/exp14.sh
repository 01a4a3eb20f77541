#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp14_dtile.txt
echo "# tabulated in-tile diagonal (DNM_DIAG_TABLE) on/off, same box, alternating" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":0}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":1}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":0}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":1}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":1,"DNM_DIAG_PASS":"last"}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_TABLE":1}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
timeout 900 python3 tools/v2_check.py 20 2>&1 | tail -2 >> $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -4 >> $O
