#!/bin/bash
# GPU box: what the bond-graph passes of kagome-30 (real arithmetic, what eigsolve runs) would take WITHOUT some of their
# hops (results are wrong: timing only) -- bounds what reworking the gathered / LDS hops of the lo pass can gain.
cd "${GRAFT_REPO_ROOT:-.}"; export DNM_EXPERIMENTAL=1
run() { echo "== $*"; env "$@" python3 tools/models_bench.py kagome30:sc --real 2>&1 | grep -i "ms\|plan" | tail -3; }
run DNM_NOP=1
run DNM_SC3G_KEEP_GATA=6
run DNM_SC3G_KEEP_GATA=0
run DNM_SC3G_KEEP_LDSA=11
run DNM_SC3G_KEEP_LDSA=0
run DNM_SC3G_KEEP_GATA=0 DNM_SC3G_KEEP_LDSA=0
run DNM_SC3G_KEEP_GATB=0
run DNM_SC3G_KEEP_GATB=0 DNM_SC3G_KEEP_LDSB=0
