#!/bin/bash
# usage: tools/prof_cmd.sh OUTFILE cmd...   kernel-trace stats of an arbitrary command (top kernels by total time)
set -u
OUTF=$1; shift
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; export DNM_EXPERIMENTAL=1
rm -rf /tmp/rp_cmd
rocprofv3 --kernel-trace --stats -f csv -d /tmp/rp_cmd -o t -- "$@" > /tmp/rp_cmd.log 2>&1
grep -v "rocprofv3\|amdgpu.ids" /tmp/rp_cmd.log | tail -3 > $OUTF
python3 - >> $OUTF <<'PY'
import csv, glob
f = glob.glob("/tmp/rp_cmd/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print("%-72s %8s %12s %12s %7s" % ("kernel", "calls", "total ms", "avg ms", "%"))
for r in rows[:14]:
    print("%-72s %8s %12.2f %12.3f %7s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                            float(r["AverageNs"]) / 1e6, r["Percentage"]))
PY
cat $OUTF
