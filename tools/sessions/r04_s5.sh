#!/bin/bash
# round 4, GPU session 5: real-arithmetic tests (verbose), kernel profile of the basis-free Lanczos at L=30
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s5; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 900 python -X faulthandler -m pytest tests/test_gpu_krylov.py -x -v -k "real_packed or real_arithmetic or filtered_default" 2>&1 | tail -60 | tee $OUT/pytest_real.txt
for m in complex real; do
  bash tools/prof_cmd.sh $OUT/lanczos_prof_$m.txt python3 tools/lanczos_prof.py 30 $m > /dev/null
done
