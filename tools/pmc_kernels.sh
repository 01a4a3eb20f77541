#!/bin/bash
# usage: tools/pmc_kernels.sh KERNEL_SUBSTR 'CTR ...' -- cmd...   last dispatch of EVERY kernel whose name matches
set -u
K=$1; CTRS=$2; shift 3
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
rm -rf /tmp/pmc_k; rocprofv3 --pmc $CTRS -d /tmp/pmc_k -o p -- "$@" > /tmp/pmc_k.txt 2>&1
python3 - "$K" <<'PY'
import sqlite3, sys, glob
c = sqlite3.connect(glob.glob("/tmp/pmc_k/*.db")[0])
rows = list(c.execute("select dispatch_id, kernel_name, counter_name, value, duration, grid_size from counters_collection where kernel_name like ? order by dispatch_id", ("%" + sys.argv[1] + "%",)))
import os
nlast = int(os.environ.get("NLAST", "1"))        # NLAST=2: the last two dispatches (both passes of a two-launch plan)
ids = {}
for d, k, n, v, dur, g in rows:
    ids.setdefault(k, set()).add(d)
keep = {k: sorted(v)[-nlast:] for k, v in ids.items()}
for d, k, n, v, dur, g in rows:
    if d in keep[k]:
        short = k.replace("(anonymous namespace)::", "").split("(")[0][-60:]
        tag = ("  [dispatch -%d]" % (len(keep[k]) - 1 - keep[k].index(d))) if nlast > 1 else ""
        print("   %-60s %-14s %18.0f   (%.3f ms, grid %d)%s" % (short, n, v, dur / 1e6, g, tag))
PY
