#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp15_unroll.txt
echo "# record loops unrolled by two" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":4,"mode":2,"amin":4,"g":6,"cp":98}]'
timeout 900 python3 tools/sweep.py 30 >> $O 2>&1
./build_tmp/stream_probe 30 2>&1 | grep -i "tile R=8 regs nt\|persist reg 512thr grid=256" >> $O
