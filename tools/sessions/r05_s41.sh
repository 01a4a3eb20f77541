#!/bin/bash
# round 5, GPU session 41: the plain-C host (examples/c_abi_demo.c)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s41; mkdir -p $OUT
gcc -O2 -Wall -Iinclude examples/c_abi_demo.c -o /tmp/c_abi_demo -Ldynamite_amd -ldynamite_amd -lm -Wl,-rpath,$PWD/dynamite_amd 2>&1 | tee $OUT/c_host.txt
for L in 20 26 30; do /tmp/c_abi_demo $L 2>&1 | tee -a $OUT/c_host.txt; done
timeout 600 python3 -m pytest tests/test_c_host.py -m gpu -q 2>&1 | tail -2 | tee -a $OUT/c_host.txt
