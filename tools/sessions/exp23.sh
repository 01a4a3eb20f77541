#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp23_final_check.txt
echo "# scheduler flags adopted: SpinConserve kernels base vs new, full GPU suite, bench" > $O
for v in base new; do
echo "== sc_bench $v" >> $O
if [ $v = base ]; then export DNM_LIB=$PWD/build_tmp/lib_base.so; else unset DNM_LIB; fi
timeout 600 python3 tools/sc_bench.py 28 32 2>&1 | grep -v amdgpu >> $O
done
unset DNM_LIB
timeout 2400 python3 -m pytest tests -q -m gpu 2>&1 | tail -5 >> $O
timeout 600 python3 bench.py 2>/dev/null >> $O
