#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp13_diag.txt
echo "# diagonal in the first or the last pass (same box, alternating)" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"first"}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"last"}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"first"}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"last"}},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"last"}},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"last","DNM_LOG_ROWS_WINDOW":4}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
DNM_DIAG_PASS=last timeout 900 python3 tools/v2_check.py 20 2>&1 | tail -2 >> $O
timeout 1500 bash tools/prof_multi.sh 30 '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98,"env":{"DNM_DIAG_PASS":"last"}}' >> $O 2>&1
