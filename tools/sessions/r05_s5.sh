#!/bin/bash
# round 5, GPU session 5: the evidence of the round on the final tree -- profiles/r05_models.txt (every Hamiltonian of the
# reference's harness + the kagome tori, before / after), counters, bench.py under rocprofv3, the suite with durations
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s5; mkdir -p $OUT
M=$OUT/models.txt
echo "== after: multiply and eigsolve(nev=2) (tools/models_bench.py --eigs)" > $M
timeout 1800 python3 tools/models_bench.py --eigs kagome27b:sc kagome30:sc kagome30:scx kagome33:sc 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $M
echo "== after: multiply (no eigsolve)" >> $M
timeout 900 python3 tools/models_bench.py bench_long_range:sc:28 bench_long_range:full:28 bench_ising:full:28 bench_xx:full:28 mbl:full:28 heisenberg:sc:32 2>&1 | grep -v "Warning\|amdgpu.ids" | tee -a $M
timeout 600 python3 tools/models_bench.py bench_syk:full:14 bench_syk:full:16 2>&1 | grep "CASE\|multiply" | tee -a $M
echo "== before (the same tree with the round-4 paths: DNM_SC_SITE_PERM=0 DNM_SC3_TILED=0 = the row kernel; XParity in reference order)" >> $M
DNM_EXPERIMENTAL=1 DNM_SC_SITE_PERM=0 DNM_SC3_TILED=0 DNM_SC_XPARITY_LAYOUT=0 timeout 900 python3 tools/models_bench.py kagome27b:sc kagome30:sc kagome30:scx bench_long_range:sc:28 2>&1 | grep "CASE\|plan\|multiply" | tee -a $M
echo "== counters (last dispatch of each kernel; FETCH_SIZE in KB: x2 x 1024 B per the guide; WRITE_SIZE in KB)" >> $M
for c in kagome30:sc kagome30:scx bench_long_range:sc:28; do
  for G in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
    echo "-- $c: $G" | tee -a $M
    bash tools/pmc_kernels.sh sc3 "$G" -- python3 tools/models_bench.py $c | grep -v "random\|copy" | tee -a $M
  done
done
echo "== run_kagome.py" >> $M
python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tee -a $M
python3 benchmarking/run_kagome.py 30 --no-z2 2>&1 | grep -v amdgpu | tee -a $M
bash tools/profile_bench.sh > $OUT/profile_bench.txt 2>&1; tail -5 $OUT/profile_bench.txt
mkdir -p $OUT/profiles; cp gpurun_out/profiles/* $OUT/profiles/ 2>/dev/null
python3 bench.py > $OUT/bench_line.json 2> $OUT/bench_err.txt
timeout 1500 python3 -m pytest tests -m gpu -q --durations=25 2>&1 | tail -45 > $OUT/test_durations.txt; tail -3 $OUT/test_durations.txt
