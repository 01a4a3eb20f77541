#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out/r02
O=gpurun_out/r02/exp33_pass_times.txt
echo "# per-pass kernel times: no prefetch of the first gathered record (rest of the new prologue kept)" > $O
bash tools/pass_times.sh nopf DNM_LIB=$PWD/dynamite_amd/build/lib_nopf.so >> $O 2>&1
bash tools/pass_times.sh new >> $O 2>&1
bash tools/pass_times.sh prev DNM_LIB=$PWD/dynamite_amd/build/lib_prev.so >> $O 2>&1
bash tools/pass_times.sh nopf2 DNM_LIB=$PWD/dynamite_amd/build/lib_nopf.so >> $O 2>&1
