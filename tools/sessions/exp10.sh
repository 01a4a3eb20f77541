#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp10_pairs.txt
echo "# gathered records in pairs (cache_policy bit 7)" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":226},
{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":226},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},
{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":226},
{"B":12,"R":3,"mode":2,"amin":4,"g":5,"cp":98,"env":{"DNM_GBITS_WINDOW":6}}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
timeout 1500 bash tools/prof_multi.sh 30 '{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":226}' >> $O 2>&1
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -5 >> $O
