#!/bin/bash
# round 3, GPU session 7: sc3 tests after fixes, filter degree / orthogonalisation experiments, serialized-load probe
set -u
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/r03_s7; mkdir -p $OUT
export DNM_EXPERIMENTAL=1
echo "== pytest sc3"; timeout 1200 python -m pytest tests/test_gpu_sc3.py -q 2>&1 | tail -30 | tee $OUT/pytest_sc3.txt
echo "== sc_bench"; timeout 600 python tools/sc_bench.py 32 2>&1 | grep -v amdgpu.ids | tee $OUT/sc_bench.txt
echo "== pytest distributed"; timeout 1200 python -m pytest tests/test_gpu_distributed.py -x -q 2>&1 | tail -8 | tee $OUT/pytest_dist.txt
{
echo "== filter degree, L=28 nev=5 tol 1e-10"
for d in 5 9 13 17; do
  DNM_EIGS_FILTER_DEGREE=$d DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain 2>&1 | grep "^L=\|restarts," | tail -2
done
echo "== with partial re-orthogonalisation"
for d in 9 21; do
  DNM_EIGS_FILTER_PRO=1 DNM_EIGS_FILTER_DEGREE=$d DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 28 mbl 5 1e-10 lowest --no-plain 2>&1 | grep "^L=\|restarts," | tail -2
done
echo "== L=30 nev=3 tol 1e-8"
DNM_EIGS_FILTER=0 DNM_KRYLOV_DEBUG=1 timeout 900 python - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from dynamite_amd import models
from dynamite_amd.config import config
from dynamite_amd.computations import eigsolve
config._initialize()
H = models.mbl(30); H.establish_L()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    vals = H.eigsolve(nev=3, tol=1e-8)
    torch.cuda.synchronize()
    print("L=30 mbl nev=3 plain: %.3f s %s %s" % (time.perf_counter() - t0, eigsolve.last_stats, vals[:3]), flush=True)
PY
for d in 9 15; do
  DNM_EIGS_FILTER_DEGREE=$d DNM_KRYLOV_DEBUG=1 timeout 900 python tools/eigs_filter_bench.py 30 mbl 3 1e-8 lowest --no-plain 2>&1 | grep "^L=\|restarts," | tail -2
done
} 2>&1 | grep -v amdgpu.ids | tee $OUT/eigs_filter.txt
hipcc --offload-arch=gfx950 -O3 tools/copy_probe2.hip -o /tmp/copy_probe2 && timeout 300 /tmp/copy_probe2 30 2>&1 | tail -12 | tee $OUT/copy_probe2.txt
