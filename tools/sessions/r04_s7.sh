#!/bin/bash
# round 4, GPU session 7: real-arithmetic eigsolve with lane-agnostic records, the round's profile artefacts of the default
# bench line, counters of the SpinConserve kernels, full GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s7; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
timeout 600 python -m pytest tests/test_gpu_krylov.py -x -q -o faulthandler_timeout=120 -k "real_packed or real_arithmetic" 2>&1 | tail -4 | tee $OUT/pytest_real.txt
bash tools/prof_cmd.sh $OUT/lanczos_prof_real.txt python3 tools/lanczos_prof.py 30 real > /dev/null
DNM_KRYLOV_DEBUG=1 DNM_EIGS_REAL=1 timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain 2>&1 | grep -E "filtered\)|L=30" | grep -v "restart [0-9]*," | cut -c1-300 | tee $OUT/real_nev3_L30.txt
DNM_EIGS_REAL=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain 2>&1 | grep -E "L=28" | cut -c1-300 | tee $OUT/real_nev5_L28.txt
# profile artefacts of the default bench run
bash tools/profile_bench.sh > $OUT/profile_bench.log 2>&1; tail -5 $OUT/profile_bench.log
# the default line (Krylov phases + cpu baseline)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_stderr.txt; tail -c 1500 $OUT/bench_default.json
# counters of the SpinConserve kernels
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  echo "-- $G" | tee -a $OUT/sc3_pmc.txt
  bash tools/pmc_kernels.sh sc3_ "$G" -- python3 tools/sc_bench.py 32 | grep -E "sc3_" | tee -a $OUT/sc3_pmc.txt
done
python3 tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee -a $OUT/sc3_pmc.txt
timeout 2400 python -m pytest tests -q -x -m gpu -o faulthandler_timeout=600 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
