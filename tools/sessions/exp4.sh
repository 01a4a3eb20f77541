#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp4_tests.txt
echo "# first-class swizzled layout: GPU tests" > $O
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -30 >> $O
echo "== DNM_TEST_SWZ=0" >> $O
DNM_TEST_SWZ=0 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $O
echo "== DNM_TEST_SWZ=9" >> $O
DNM_TEST_SWZ=9 timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -8 >> $O
echo "== bench" >> $O
timeout 600 python3 bench.py --steps 10 --warmup 2 >> $O 2>gpurun_out/r02/exp4_bench_stderr.txt
tail -5 gpurun_out/r02/exp4_bench_stderr.txt >> $O
