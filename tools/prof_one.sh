#!/bin/bash
# usage: tools/prof_one.sh TAG 'SWEEP-JSON' L  -- kernel trace + two PMC passes for one plan configuration
set -u
TAG=$1; export SWEEP="$2"; L=${3:-30}
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- python3 tools/sweep.py $L > $OUT/run.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc1 -o p -- python3 tools/sweep.py $L > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -o p -- python3 tools/sweep.py $L > /dev/null 2>&1
find $OUT -name "*.csv" | head -20
python3 - <<PY
import csv, glob, collections
for f in glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    agg = collections.OrderedDict()
    for r in rows:
        n = r["Kernel_Name"][:60]
        if "tile_pass" not in n and "gather" not in n: continue
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        agg.setdefault((n, r.get("Grid_Size_X", r.get("Grid_Size",""))), []).append(d)
    seq = [(int(r["Start_Timestamp"]), r["Kernel_Name"][:50], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6) for r in rows if "tile_pass" in r["Kernel_Name"]]
    seq.sort()
    print("last launches (ms):", [round(s[2], 3) for s in seq[-8:]])
for tag in ("pmc1", "pmc2"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % tag, recursive=True):
        rows = list(csv.DictReader(open(f)))
        per = collections.OrderedDict()
        for r in rows:
            if "tile_pass" not in r["Kernel_Name"]: continue
            per.setdefault((r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
        last = list(per.items())[-8:]
        for k, v in last:
            print(tag, "dispatch", k, {a: round(b / 1e6, 3) for a, b in v.items()}, "(x1e6)")
PY
