#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp17_scalar_prefetch.txt
echo "# next record's scalar loads behind this record's LDS reads (lib_pipe) vs base, alternating processes on one box" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98}]'
for i in 1 2 3; do
for v in base pipe; do
echo "== $v" >> $O
DNM_LIB=$PWD/build_tmp/lib_$v.so timeout 300 python3 tools/sweep.py 30 2>&1 | grep "L=30" >> $O
done; done
