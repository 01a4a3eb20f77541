#!/bin/bash
# GPU box: the two forms of a basis-free Lanczos step at L=30 side by side, with counters.
#   plain:    fused multiply (window pass + contiguous pass with the beta vector) + the update sweep        144 B/amp
#   deferred: the sweep folded into the next multiply's first pass (tools/experiments/r04_deferred_lanczos.patch, its
#             kernel / handle parts applied to the round-6 tree: dynamite_amd/build/exp/lib_deferred.so)    112 B/amp
# per kernel: ms (kernel trace), FETCH_SIZE / WRITE_SIZE, L2 hits, SQ busy / wave cycles, VALU share, occupancy
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp DNM_EXPERIMENTAL=1 DNM_LIB=$PWD/dynamite_amd/build/exp/lib_deferred.so
OUT=gpurun_out/deferred; mkdir -p $OUT
for form in 0 1; do
  export DNM_EIGS_DEFER=$form
  echo "######## DNM_EIGS_DEFER=$form ($([ $form = 0 ] && echo plain || echo deferred))"
  python3 tools/lanczos_prof.py 30 complex 2>&1 | grep "eigsolve"
  bash tools/prof_cmd.sh $OUT/trace_$form.txt python3 tools/lanczos_prof.py 30 complex | grep -i "kernel \|tile_pass\|axpby\|lanczos\|sweep\|update\|scale" | head -12
  NLAST=2 bash tools/pmc_kernels.sh tile_pass 'FETCH_SIZE' -- python3 tools/lanczos_prof.py 30 complex
  NLAST=2 bash tools/pmc_kernels.sh tile_pass 'WRITE_SIZE TCC_HIT_sum TCC_MISS_sum' -- python3 tools/lanczos_prof.py 30 complex
  NLAST=2 bash tools/pmc_kernels.sh tile_pass 'SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VALU' -- python3 tools/lanczos_prof.py 30 complex
done
