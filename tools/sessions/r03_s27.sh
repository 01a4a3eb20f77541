#!/bin/bash
# round 3, GPU session 27: profiles of the default bench again (the summary step of session 26 failed on a key)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s27; mkdir -p $OUT
bash tools/profile_bench.sh > $OUT/profile_bench.txt 2>&1
cp -r gpurun_out/profiles $OUT/profiles
