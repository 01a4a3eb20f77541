#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp32_pass_times.txt
echo "# per-pass kernel times, new prologue against the previous library (same box)" > $O
bash tools/pass_times.sh new >> $O 2>&1
bash tools/pass_times.sh prev DNM_LIB=$PWD/dynamite_amd/build/lib_prev.so >> $O 2>&1
bash tools/pass_times.sh new2 >> $O 2>&1
bash tools/pass_times.sh prev2 DNM_LIB=$PWD/dynamite_amd/build/lib_prev.so >> $O 2>&1
