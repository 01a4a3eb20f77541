#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp22_sched.txt
echo "# compiler-flag variants of the whole library, alternating processes on one box" > $O
export SWEEP='[{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98},{"B":12,"R":3,"mode":2,"amin":4,"g":6,"cp":98}]'
for i in 1; do
for v in base relaxocc maxilp memclause bias0 bias50 trackers minreg iterilp nohrp; do
echo "== $v" >> $O
DNM_LIB=$PWD/build_tmp/lib_$v.so timeout 300 python3 tools/sweep.py 30 2>&1 | grep "L=30" >> $O
done; done
DNM_LIB=$PWD/build_tmp/lib_maxilp.so timeout 600 python3 tools/v2_check.py 20 2>&1 | tail -1 >> $O
