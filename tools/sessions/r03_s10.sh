#!/bin/bash
# round 3, GPU session 10: fuzz of the internal layout, SpinConserve solvers in the layout, transposed-exchange timeline
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s10; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
echo "== pytest sc3"; timeout 1200 python -m pytest tests/test_gpu_sc3.py -q 2>&1 | tail -12 | tee $OUT/pytest_sc3.txt
{
echo "== SpinConserve(32,16) solvers, internal layout (default)"
timeout 900 python tools/sc_eigs_bench.py 32 1e-8 2
timeout 900 python tools/sc_evolve_bench.py 2>&1 | tail -6
echo "== the same in reference order (DNM_SC_LAYOUT=0)"
DNM_SC_LAYOUT=0 timeout 900 python tools/sc_eigs_bench.py 32 1e-8 2
DNM_SC_LAYOUT=0 timeout 900 python tools/sc_evolve_bench.py 2>&1 | tail -6
} 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_solvers.txt
echo "== timeline"; timeout 900 python tools/transpose_timeline.py 27 4 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee $OUT/transpose_timeline.txt
timeout 900 python tools/transpose_timeline.py 28 8 2>&1 | grep -v "amdgpu.ids\|socket.cpp" | tee -a $OUT/transpose_timeline.txt
