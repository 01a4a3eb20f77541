#!/bin/bash
# round 4, GPU session 1: overlap experiments on tile_pass_kernel (VERDICT r3 task 1): s_setprio placements, lane-local
# LDS records before the barrier, successor-tile touch, rotated layout with counters; same-box copy probe.
# Variant libraries: tools/build_variant.py (dynamite_amd/build/exp/lib_*.so)
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r04_s1; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
EXP=$PWD/dynamite_amd/build/exp
# 1. new default-plan parity tests on the default build, then the variants that change code paths
timeout 900 python -m pytest tests/test_gpu_matvec.py -q -x -k "default_plan or default_spinconserve" 2>&1 | tail -5 | tee $OUT/pytest_default_plan.txt
for v in split2prio1 pfy512 prio3; do
  echo "== variant $v" | tee -a $OUT/pytest_variants.txt
  DNM_LIB=$EXP/lib_$v.so timeout 600 python -m pytest tests/test_gpu_matvec.py -q -x -k "default_plan_vs_oracle" 2>&1 | tail -3 | tee -a $OUT/pytest_variants.txt
done
# 2. per-pass times, two rounds interleaved
for rnd in 1 2; do
  for v in head base prio1 prio2 prio3 split1 split2 split2prio1 pf256 pf512 pf1024 pfy512 rot6; do
    bash tools/pass_times.sh $v DNM_EXPERIMENTAL=1 DNM_LIB=$EXP/lib_$v.so 2>&1 | grep "avg" | tee -a $OUT/pass_times.txt
  done
done
# 3. counters: base against the rotated layout
for v in base rot6; do
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
    echo "-- $v: $G" | tee -a $OUT/rot_pmc.txt
    DNM_LIB=$EXP/lib_$v.so bash tools/pmc_kernels.sh tile_pass "$G" -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 1 | tee -a $OUT/rot_pmc.txt
  done
done
# 4. same-box copy probe
hipcc --offload-arch=gfx950 -O3 tools/copy_probe2.hip -o /tmp/copy_probe2 && timeout 600 /tmp/copy_probe2 30 | grep -E "tile|NT=1024 R=4|NT= 256 R=1" | tee $OUT/copy_probe2.txt
