#!/bin/bash
# usage: tools/pmc_cmd.sh KERNEL_SUBSTR 'CTR ...' -- cmd...   prints the last dispatch of the matching kernel
set -u
K=$1; CTRS=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
rm -rf /tmp/pmc_cmd; rocprofv3 --pmc $CTRS -d /tmp/pmc_cmd -o p -- "$@" > /tmp/pmc_cmd.txt 2>&1
python3 - "$K" <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob("/tmp/pmc_cmd/*.db")[0])
rows = list(c.execute("select dispatch_id, counter_name, value, duration, grid_size from counters_collection where kernel_name like ? order by dispatch_id", ("%" + sys.argv[1] + "%",)))
if rows:
    last = max(r[0] for r in rows)
    for d, n, v, dur, g in rows:
        if d == last: print("   %-28s %18.0f   (%.3f ms, grid %d)" % (n, v, dur / 1e6, g))
PY
