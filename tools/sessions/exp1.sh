#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp1_swz.txt
echo "# XOR swizzle experiment (v1 kernel patched; cache_policy bits 8..13 = S)" > $O
export SWEEP='[{"B":12,"R":4,"mode":2,"amin":-1,"g":9,"cp":98},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":4450},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":3938},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":4194},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":4706},{"B":12,"R":4,"mode":2,"amin":-1,"g":9,"cp":4450},{"B":12,"R":4,"mode":2,"amin":4,"g":5,"cp":4450},{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":4450},{"B":12,"R":4,"mode":2,"amin":6,"g":6,"cp":4450}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
unset SWEEP
timeout 1500 bash tools/prof_multi.sh 30 '{"B":12,"R":4,"mode":2,"amin":-1,"g":9,"cp":98}' '{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":4450}' '{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":3938}' '{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":4706}' '{"B":12,"R":4,"mode":2,"amin":-1,"g":9,"cp":4450}' >> $O 2>&1
tail -30 $O
