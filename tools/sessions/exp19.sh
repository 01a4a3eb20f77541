#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 900 python3 tools/krylov_L30.py > gpurun_out/r02/r02_krylov_L30.txt 2>&1
timeout 900 bash tools/prof_cmd.sh gpurun_out/r02/r02_krylov_L30_prof.txt python3 tools/krylov_L30.py eigs > /dev/null 2>&1
cat gpurun_out/r02/r02_krylov_L30.txt | grep -v amdgpu; cat gpurun_out/r02/r02_krylov_L30_prof.txt
