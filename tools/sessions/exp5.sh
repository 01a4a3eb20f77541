#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp5_tests.txt
echo "# GPU tests under other swizzle shifts" > $O
for s in 9 5 12; do
echo "== DNM_TEST_SWZ=$s" >> $O
DNM_TEST_SWZ=$s timeout 1500 python3 -m pytest tests -q -m gpu 2>&1 | tail -12 >> $O
done
