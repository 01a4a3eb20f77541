#!/bin/bash
# GPU box: the harness's long_range model on the Full space at L=28 (the planner's worst case: 27 bonds + 28 single flips)
# under plans of two and three passes -- what tiles of more segments could buy at best.
#   default:       2 passes, 18 index bits each (12 in LDS + 6 in the XCD group's L2), 17 masks gathered
#   DNM_GBITS=0:   no XCD groups: 3 passes of 12 LDS bits, the few masks no tile holds gathered from HBM -- the shape a plan
#                  of multi-segment tiles would have (every mask from LDS, three sweeps over the vectors)
#   DNM_GBITS=3:   in between
cd "${GRAFT_REPO_ROOT:-.}"; export DNM_EXPERIMENTAL=1
for g in default 0 3; do
  echo "== DNM_GBITS=$g"
  if [ $g = default ]; then env -u DNM_GBITS python3 tools/models_bench.py bench_long_range:full:28 2>&1 | grep -i "plan\|multiply" | cut -c1-700
  else DNM_GBITS=$g python3 tools/models_bench.py bench_long_range:full:28 2>&1 | grep -i "plan\|multiply" | cut -c1-700; fi
done
echo "== for scale: the random-field Heisenberg chain (2 passes, 80 B/amp)"
python3 tools/models_bench.py mbl:full:28 2>&1 | grep -i "multiply" | cut -c1-200
