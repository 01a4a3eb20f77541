#!/bin/bash
# round 5, GPU session 11: eigsolve(nev > 1) without a stored basis (deflation) -- tests, kagome-30 timing against the
# filtered restart, the 36-site torus with nev = 2
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r05_s11; mkdir -p $OUT
python3 -c "import torch; f,t=torch.cuda.mem_get_info(); print('device memory: free %.1f GiB of %.1f GiB' % (f/2**30, t/2**30))" 2>&1 | grep -v amdgpu | tee $OUT/deflation.txt
timeout 900 python3 -m pytest tests/test_gpu_krylov.py tests/test_gpu_sc3_graph.py -m gpu -q -x -k "deflated or fuzz or basis_free or eigsolve" 2>&1 | tail -5 | tee -a $OUT/deflation.txt
timeout 900 python3 -m pytest tests/test_gpu_distributed.py -m gpu -q -x 2>&1 | tail -5 | tee -a $OUT/deflation.txt
echo "== run_kagome 30 (filtered thick restart, the default)" | tee -a $OUT/deflation.txt
DNM_KRYLOV_DEBUG=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -6 | tee -a $OUT/deflation.txt
echo "== run_kagome 30, DNM_EIGS_BASISFREE=1 (deflation)" | tee -a $OUT/deflation.txt
DNM_EIGS_BASISFREE=1 DNM_KRYLOV_DEBUG=1 python3 benchmarking/run_kagome.py 30 2>&1 | grep -v amdgpu | tail -7 | tee -a $OUT/deflation.txt
echo "== kagome 36a, nev = 1 then nev = 2" | tee -a $OUT/deflation.txt
DNM_TEST_LARGEST=1 DNM_KRYLOV_DEBUG=1 timeout 1500 python3 -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -s -k kagome36 2>&1 | grep -v amdgpu | tail -12 | tee -a $OUT/deflation.txt
