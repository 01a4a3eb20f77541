#!/bin/bash
# round 4, GPU session 3: lo pass with 2^m rows per workgroup (VERDICT r3 task 2) against round 3's, wave priority in the
# SpinConserve passes, full GPU suite on the new default build, bench line with the Krylov phases, FETCH of pass 0
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s3; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
EXP=$PWD/dynamite_amd/build/exp
# 1. SpinConserve parity first (the kernels that changed), then everything
timeout 1500 python -m pytest tests/test_gpu_sc3.py -q -x 2>&1 | tail -5 | tee $OUT/pytest_sc3.txt
timeout 600 python -m pytest tests/test_gpu_matvec.py -q -x -k "default_spinconserve" 2>&1 | tail -3 | tee -a $OUT/pytest_sc3.txt
# 2. SpinConserve(32,16) per-kernel times: round 3's kernels / spans / spans + wave priority
for rnd in 1 2; do
  for v in sc3head sc3base sc3prio; do
    echo "== $v" | tee -a $OUT/sc3_times.txt
    DNM_LIB=$EXP/lib_$v.so bash tools/prof_cmd.sh /tmp/st_$v.txt python3 tools/sc_bench.py 32 > /dev/null
    grep -E "sc3_.*pass|SpinConserve L" /tmp/st_$v.txt | cut -c1-150 | tee -a $OUT/sc3_times.txt
  done
done
# 3. one rank of config 5
for v in sc3head sc3base sc3prio; do
  echo "== $v" | tee -a $OUT/sc3_config5.txt
  DNM_LIB=$EXP/lib_$v.so timeout 600 python tools/sc3_config5.py --rank 3 2>&1 | grep -E "rank 3 of|split" | tee -a $OUT/sc3_config5.txt
done
# 4. counters of the new SpinConserve kernels
for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  echo "-- $G" | tee -a $OUT/sc3_pmc.txt
  DNM_LIB=$EXP/lib_sc3base.so bash tools/pmc_kernels.sh sc3_ "$G" -- python3 tools/sc_bench.py 32 | grep -E "sc3_" | tee -a $OUT/sc3_pmc.txt
done
# 5. the default bench line (Krylov phases + cpu baseline)
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default_stderr.txt; tail -c 3000 $OUT/bench_default.json
# 6. FETCH of both passes, head against the new default, with the standard profile tool's command
for v in head base; do
  for G in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- $v: $G" | tee -a $OUT/fetch_head_base.txt
    NLAST=2 DNM_LIB=$EXP/lib_$v.so bash tools/pmc_kernels.sh tile_pass "$G" -- python3 bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-secondary | tee -a $OUT/fetch_head_base.txt
  done
done
# 7. everything else
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -8 | tee $OUT/pytest_gpu.txt
