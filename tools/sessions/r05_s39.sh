#!/bin/bash
# round 5, GPU session 39: the reference harness's command lines over its Hamiltonians and subspaces (with and without
# --xparity) at L = 24..26 -- every phase, as a regression sweep of the tree at the end of the round
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s39; mkdir -p $OUT
M=$OUT/harness_sweep.txt
run() { echo "== benchmark.py $*" | tee -a $M; timeout 300 python3 benchmarking/benchmark.py "$@" 2>&1 | grep -v "Warning\|amdgpu.ids" | tail -14 | cut -c1-160 | tee -a $M; }
for Hn in MBL long_range ising XX heisenberg; do
  run -L 24 -H $Hn --shell --gpu --mult --norm --evolve -t 1 --eigsolve --nev 2 --rdm --check-conserves
done
for Hn in MBL XX heisenberg long_range; do
  run -L 26 -H $Hn --shell --gpu --subspace spinconserve --mult --norm --evolve -t 1 --eigsolve --nev 2 --rdm
done
run -L 26 -H heisenberg --shell --gpu --subspace spinconserve --xparity plus --mult --norm --evolve -t 1 --eigsolve --nev 2
run -L 24 -H ising --shell --gpu --xparity plus --mult --norm --evolve -t 1 --eigsolve --nev 2
run -L 24 -H heisenberg --shell --gpu --subspace parity --xparity minus --mult --norm --evolve -t 1 --eigsolve --nev 2
run -L 22 -H heisenberg --shell --gpu --subspace auto --mult --norm --evolve -t 1 --eigsolve --nev 2
run -L 16 -H SYK --shell --gpu --mult --norm --evolve -t 1 --eigsolve --nev 2
grep -c "Traceback\|Error" $M | tee -a $M
